#!/usr/bin/env python3
"""Development probe: isac_covariance_dev at the bench shapes (A = 64: 733 824 samples; --ants 256) -- GPU time (HIP events, kernel + the two
reducers), max |Ra - G'G/N| against NumPy on a sub-sampled column set, run-to-run bit identity.   python tools/cov_probe.py [--ants 64]"""
import argparse, ctypes as C, hashlib, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--ants", type=int, default=64); ap.add_argument("--reps", type=int, default=30)
args = ap.parse_args()
pkg = importlib.import_module(bench.PKG)
ctx = pkg.Context(0)
K, L, A = 3276, 224, args.ants
N = K * L
g = ctx.empty((K, L, A))
ctx.check(ctx.lib.isac_synth_qpsk_grid_dev(ctx.handle, C.c_void_p(g.ptr), K, L, A, C.c_uint64(5), 0))
ra = ctx.empty((A, A))
def cov(): ctx.check(ctx.lib.isac_covariance_dev(ctx.handle, C.c_void_p(g.ptr), C.c_int64(N), C.c_int32(A), C.c_void_p(ra.ptr)))
for _ in range(3): cov()
ts, hs = [], set()
for _ in range(args.reps):
    ctx.sync(); ctx.timer_start(); cov(); ts.append(ctx.timer_stop_ms()); hs.add(hashlib.sha256(ra.numpy().tobytes()).hexdigest()[:12])
ts = np.array(ts)
G = g.numpy().reshape(N, A, order="F")
ref = (G.conj().T @ G) / N
err = np.abs(ra.numpy() - ref).max() / np.abs(ref).max()
print(f"A {A}: covariance min {ts.min():.4f} ms median {np.median(ts):.4f} ms; rel err vs NumPy {err:.2e}; {len(hs)} distinct result(s) over {args.reps} runs; env {os.environ.get('ISAC_COV_REG_OPERANDS')}")
