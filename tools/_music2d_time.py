#!/usr/bin/env python3
"""Development probe: sensing.estimation.music2D (music2D.m) on the bench shape, device-resident inputs."""
import importlib, os, sys, time
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG)
for ants in (16, 64):
    cell = bench.Cell(pkg, 0, 0, ants, 16, 1, inflight=1)
    cell.step()
    echo, txg = cell.echo[0], cell.tx_grids[0]
    f = lambda: pkg.sensing.estimation.music2D(cell.rp, SimpleNamespace(scs=30), echo, txg, ctx=cell.ctx)
    est = f(); cell.ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5): est = f()
    cell.ctx.sync()
    print(f"A={ants}: music2D {1e3 * (time.perf_counter() - t0) / 5:.2f} ms per call  L={est.L} azi={est.aziEst} rng={est.rngEst[:3]} vel={est.velEst[:3]}")
