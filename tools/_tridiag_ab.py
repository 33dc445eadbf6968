#!/usr/bin/env python3
"""Development probe: digest + host-call time of isac_eigh (tridiagonal-QL pipeline) for a few orders above 64 -- run once with and once without
ISAC_EIG_TRIDIAG_UNFUSED to compare the one-pass Householder reduction with the two-pass one bit for bit."""
import ctypes as C, hashlib, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module(bench.PKG)
ctx = pkg._lib.Context(0)
for a in (65, 100, 128, 129, 200, 256, 320):
    rng = np.random.default_rng(a)
    m = rng.standard_normal((a, a)) + 1j * rng.standard_normal((a, a))
    h = np.asfortranarray(m @ m.conj().T / a + np.diag(rng.uniform(0, 3, a)))
    w = np.zeros(a); v = np.zeros((a, a), dtype=np.complex128, order="F")
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        ctx.check(ctx.lib.isac_eigh(ctx.handle, h.ctypes.data_as(C.c_void_p), C.c_int32(a), w.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)))
        ts.append(time.perf_counter() - t0)
    wr = np.linalg.eigvalsh(h)
    ok = np.abs(w - wr).max() < 1e-12 * np.abs(wr).max() and np.abs(h @ v - v * w).max() < 1e-11 * np.abs(wr).max()
    print(f"n={a:4d} digest {hashlib.sha256(w.tobytes() + v.tobytes()).hexdigest()[:16]} ok={ok} call {1e3 * min(ts):.3f} ms")
