#!/usr/bin/env python3
"""Development probe: do two different stages co-execute when looped on two contexts at the same time?
Prints each stage's solo time per call and the per-pair wall time when both loops run concurrently."""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import numpy as np

pkg = importlib.import_module(bench.PKG)
cell = bench.Cell(pkg, 0, 0, 64, 16, 1, inflight=2)
c1, c2 = cell.ctxs
lib = c1.lib
for _ in range(3):
    cell.step()
from importlib import import_module
m = import_module(pkg.__name__ + ".sensing.estimation.fft2D")
mm = import_module(pkg.__name__ + ".sensing._marshal")
r0, r1, q0, q1 = m._cut_rectangle(cell.cfar.CUTIdx)
det = cell.cfar.cfarDetector2D
cf = cell.L.CfarConfig(det.ProbabilityFalseAlarm, (C.c_int32 * 2)(*det.GuardBandSize), (C.c_int32 * 2)(*det.TrainingBandSize), r0, r1, q0, q1)
ep = mm.est_block(cell.rp)
ras = {id(c): c.empty((cell.A, cell.A)) for c in (c1, c2)}


def mono(c, slot):
    pkg.sensing.monoStaticSensing(cell.tx_waves[slot], (cell.K, cell.Lsym, cell.A), cell.carrier, cell.rp, cell.los, seed=cell.seed, nfft=4096, out=cell.echo[slot], ctx=c)
def rng(c, slot):
    c.check(lib.isac_fft2d_range_stage_dev(c.handle, C.byref(ep), C.byref(cf), C.c_void_p(cell.echo[slot].ptr), C.c_void_p(cell.tx_grids[slot].ptr), cell.K, cell.Lsym, cell.A))
def cov(c, slot):
    c.check(lib.isac_covariance_dev(c.handle, C.c_void_p(cell.echo[slot].ptr), C.c_int64(cell.K * cell.Lsym), C.c_int32(cell.A), C.c_void_p(ras[id(c)].ptr)))

stages = {"mono": mono, "range": rng, "cov": cov}
N = 20
solo = {}
for name, fn in stages.items():
    fn(c1, 0); c1.sync()
    t0 = time.perf_counter()
    for _ in range(N):
        fn(c1, 0)
    c1.sync()
    solo[name] = 1e3 * (time.perf_counter() - t0) / N
    print(f"solo {name:6s} {solo[name]:.3f} ms")
names = list(stages)
for i, a in enumerate(names):
    for b in names[i:]:
        c1.sync(); c2.sync()
        t0 = time.perf_counter()
        for _ in range(N):
            stages[a](c1, 0)
            stages[b](c2, 1)
        c1.sync(); c2.sync()
        pair = 1e3 * (time.perf_counter() - t0) / N
        print(f"pair {a:6s}+ {b:6s} {pair:.3f} ms   (sum of solos {solo[a] + solo[b]:.3f}, max {max(solo[a], solo[b]):.3f})")
