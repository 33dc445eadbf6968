import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
pkg = importlib.import_module(bench.PKG)
cell = bench.Cell(pkg, 0, 0, 64, 16, 1)
ctx = cell.ctx
lib = ctx.lib
cell.step()
A = 64
ra = np.zeros((A, A), dtype=np.complex128, order="F")
ctx.check(lib.isac_fft2d_get_covariance(ctx.handle, ra.ctypes.data_as(C.c_void_p), C.c_int32(A)))
w = np.zeros(A); v = np.zeros((A, A), dtype=np.complex128, order="F")
for trial, h in [("Ra", ra), ("random", None)]:
    if h is None:
        rng = np.random.default_rng(0); m = rng.standard_normal((A, A)) + 1j * rng.standard_normal((A, A)); h = np.asfortranarray(m @ m.conj().T)
    t0 = time.perf_counter()
    ctx.check(lib.isac_eigh(ctx.handle, h.ctypes.data_as(C.c_void_p), C.c_int32(A), w.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)))
    dt = time.perf_counter() - t0
    wr = np.linalg.eigvalsh(h)
    print(trial, "host ms", dt * 1e3, "eig rel err", np.abs(w - wr).max() / np.abs(wr).max(), "small-eig rel err", np.abs((w - wr) / wr).max(),
          "cond", wr.max() / wr.min(), "resid", np.abs(h @ v - v * w).max() / np.abs(wr).max(), "orth", np.abs(v.conj().T @ v - np.eye(A)).max())
