#!/usr/bin/env python3
"""Development probe: host-call time of isac_eigh for several orders (run twice: default dispatch and ISAC_EIG_QL=1)."""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
pkg = importlib.import_module(bench.PKG)
ctx = pkg.default_context()
for a in (4, 8, 16, 24, 32, 48, 64):
    rng = np.random.default_rng(a)
    m = rng.standard_normal((a, a)) + 1j * rng.standard_normal((a, a))
    h = np.asfortranarray(m @ m.conj().T / a + np.diag(rng.uniform(0, 3, a)))
    w = np.zeros(a); v = np.zeros((a, a), dtype=np.complex128, order="F")
    def run():
        ctx.check(ctx.lib.isac_eigh(ctx.handle, h.ctypes.data_as(C.c_void_p), C.c_int32(a), w.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)))
    run(); run()
    t0 = time.perf_counter()
    for _ in range(20): run()
    ms = 1e3 * (time.perf_counter() - t0) / 20
    err = np.abs(np.sort(w) - np.linalg.eigvalsh(h)).max() / np.abs(w).max()
    print(f"A={a:3d}  {ms:7.3f} ms  eig err {err:.1e}  orth {np.abs(v.conj().T @ v - np.eye(a)).max():.1e}")
