#!/usr/bin/env python3
"""Development probe: CDL apply (one slot, 64 -> 2 antennas) and batched SINR -> CQI (273 PRB x 14 symbols of REs) kernel times."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
pkg = importlib.import_module(bench.PKG)
ctx = pkg.default_context()
CM, PL = pkg.communication.channelModels, pkg.communication.phyLayer
rng = np.random.default_rng(0)
T, nt = 61440, 64
x = ctx.to_device(np.asfortranarray(rng.standard_normal((T, nt)) + 1j * rng.standard_normal((T, nt))))
for prof in ("CDL-D", "CDL-A"):
    ch = CM.CDLChannel(DelayProfile=prof, TransmitAntennaArraySize=(4, 8, 2, 1, 1))
    CM.applyCDL(ch, x, ctx=ctx); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(10): CM.applyCDL(ch, x, ctx=ctx)
    ctx.sync(); wall = 1e3 * (time.perf_counter() - t0) / 10
    n_paths = ch.path_delays().size
    print(f"{prof}: applyCDL one slot [61440 x 64] -> [61440 x 2], {n_paths} paths: {wall:.2f} ms per call incl. host parameter prep")
n_re = 3276 * 14
for nr, p, nl in ((2, 4, 1), (2, 4, 2), (4, 32, 4)):
    h = ctx.to_device(np.asfortranarray(rng.standard_normal((n_re, nr, p)) + 1j * rng.standard_normal((n_re, nr, p))))
    w, _ = np.linalg.qr(rng.standard_normal((p, nl)) + 1j * rng.standard_normal((p, nl)))
    PL.cqiFromChannel(h, 0.5, w, ctx=ctx); ctx.sync()
    ctx.timer_start()
    for _ in range(10): PL.cqiFromChannel(h, 0.5, w, ctx=ctx)
    print(f"SINR->CQI over {n_re} REs, Nr={nr} P={p} layers={nl}: {ctx.timer_stop_ms() / 10:.3f} ms per call")
