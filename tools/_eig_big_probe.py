import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module(bench.PKG)
ctx = pkg.Context(0); lib = ctx.lib
for A in (96, 128, 256):
    rng = np.random.default_rng(A); m = rng.standard_normal((A, A)) + 1j * rng.standard_normal((A, A)); h = np.asfortranarray(m @ m.conj().T)
    w = np.zeros(A); v = np.zeros((A, A), dtype=np.complex128, order="F")
    for _ in range(2):
        t0 = time.perf_counter()
        ctx.check(lib.isac_eigh(ctx.handle, h.ctypes.data_as(C.c_void_p), C.c_int32(A), w.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)))
        dt = time.perf_counter() - t0
    wr = np.linalg.eigvalsh(h)
    print(A, "ms", round(dt * 1e3, 2), "err", np.abs(w - wr).max() / np.abs(wr).max(), "orth", np.abs(v.conj().T @ v - np.eye(A)).max())
