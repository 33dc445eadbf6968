#!/usr/bin/env python3
"""Per-stage GPU times of one CPI with HIP events on the context stream (development aid).
   python tools/stage_times.py [--ants 64]"""
import argparse, ctypes as C, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--ants", type=int, default=64); ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
pkg = importlib.import_module(bench.PKG)
cell = bench.Cell(pkg, 0, 0, args.ants, 16, 1)

ctx = cell.ctx; lib = ctx.lib; echo = cell.echo[0]
for _ in range(3):
    cell.step()
def timed(fn, reps=args.reps):
    ts = []
    for _ in range(reps):
        ctx.sync(); ctx.timer_start(); t0 = time.perf_counter(); fn(); ms = ctx.timer_stop_ms(); ts.append((ms, 1e3 * (time.perf_counter() - t0)))
    a = np.array(ts); return a[:, 0].min(), np.median(a[:, 0]), np.median(a[:, 1])
def mono(): pkg.sensing.monoStaticSensing(cell.tx_wave, (cell.K, cell.Lsym, cell.A), cell.carrier, cell.rp, cell.los, seed=cell.seed, nfft=4096, out=echo, ctx=ctx)
def mono_spec(): pkg.sensing.monoStaticSensing(cell.tx_wave, (cell.K, cell.Lsym, cell.A), cell.carrier, cell.rp, cell.los, seed=cell.seed, noise_domain="spectral", nfft=4096, out=echo, ctx=ctx)
def mono_spec_fused(): pkg.sensing.monoStaticSensing(cell.tx_wave, (cell.K, cell.Lsym, cell.A), cell.carrier, cell.rp, cell.los, seed=cell.seed, noise_domain="spectral", nfft=4096, out=echo, ctx=ctx,
                                                     fuse_fft2d=(cell.rp, cell.cfar, cell.tx_grid))
def mono_time_fused(): pkg.sensing.monoStaticSensing(cell.tx_wave, (cell.K, cell.Lsym, cell.A), cell.carrier, cell.rp, cell.los, seed=cell.seed, nfft=4096, out=echo, ctx=ctx,
                                                     fuse_fft2d=(cell.rp, cell.cfar, cell.tx_grid))
def fused_cpi():
    mono_spec_fused()
    try: pkg.sensing.estimation.fft2D(cell.rp, cell.cfar, echo, cell.tx_grid, ctx=ctx, reuse_range=True)
    except pkg.IsacError: pass
def mono_nonoise(): pkg.sensing.monoStaticSensing(cell.tx_wave, (cell.K, cell.Lsym, cell.A), cell.carrier, cell.rp, cell.los, nfft=4096, out=echo, ctx=ctx)
def fft2d():
    try: pkg.sensing.estimation.fft2D(cell.rp, cell.cfar, echo, cell.tx_grid, ctx=ctx)
    except pkg.IsacError: pass
ra = ctx.empty((cell.A, cell.A))
def cov(): ctx.check(lib.isac_covariance_dev(ctx.handle, C.c_void_p(echo.ptr), C.c_int64(cell.K * cell.Lsym), C.c_int32(cell.A), C.c_void_p(ra.ptr)))
h = np.asfortranarray(np.eye(cell.A) + 0j); w = np.zeros(cell.A)
_m = importlib.import_module(bench.PKG + ".sensing.estimation.fft2D"); _mm = importlib.import_module(bench.PKG + ".sensing._marshal")
_cf = _m._cfar_block(cell.cfar); _ep = _mm.est_block(cell.rp)
def range_stage(): ctx.check(lib.isac_fft2d_range_stage_dev(ctx.handle, C.byref(_ep), C.byref(_cf), C.c_void_p(echo.ptr), C.c_void_p(cell.tx_grid.ptr), cell.K, cell.Lsym, cell.A))
def eig():
    hh = ra.numpy(); ctx.check(lib.isac_eigh(ctx.handle, hh.ctypes.data_as(C.c_void_p), C.c_int32(cell.A), w.ctypes.data_as(C.c_void_p), None))
for name, fn in [("mono spectral philox (unfused)", mono_spec), ("mono spectral philox + range (fused)", mono_spec_fused), ("mono time philox + range (fused)", mono_time_fused),
                 ("fused CPI: mono+range, cached fft2D", fused_cpi), ("monoStaticSensing (time philox)", mono), ("monoStaticSensing (no noise)", mono_nonoise), ("fft2D (all)", fft2d), ("range stage alone (range_kernel)", range_stage), ("covariance", cov), ("eigh (incl. H2D/D2H)", eig), ("whole step", cell.step)]:
    mn, med, wall = timed(fn)
    print(f"{name:32s} gpu min {mn:8.3f} ms  median {med:8.3f} ms   host wall median {wall:8.3f} ms")
