#!/usr/bin/env python3
"""Development probe: CDL apply (one slot; DL 64 -> 2 and UL 2 -> 64 antennas), batched SINR -> CQI (273 PRB x 14 symbols of REs) and the
CSI report (Type-I PMI search + subband CQI on the reference's 4-port CSI-RS, 546 REs) times."""
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

# UL direction (cdl.m:78-85, stepped at gNBPhy.m:838-840): Nt = 2 UE antennas -> Nr = 64 gNB antennas
xu = ctx.to_device(np.asfortranarray(rng.standard_normal((T, 2)) + 1j * rng.standard_normal((T, 2))))
for prof in ("CDL-D", "CDL-A"):
    ch = CM.CDLChannel(DelayProfile=prof, TransmitAntennaArraySize=(1, 1, 2, 1, 1), ReceiveAntennaArraySize=(4, 8, 2, 1, 1))
    CM.applyCDL(ch, xu, ctx=ctx); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(10): CM.applyCDL(ch, xu, ctx=ctx)
    ctx.sync(); wall = 1e3 * (time.perf_counter() - t0) / 10
    print(f"{prof} UL: applyCDL one slot [61440 x 2] -> [61440 x 64], {ch.path_delays().size} paths: {wall:.2f} ms per call incl. host parameter prep")
# CSI report: uePhy.m:901-908 on the reference's CSI-RS configuration (4 ports, 2 REs per RB, 273 RBs), ranks 1 and 2
from types import SimpleNamespace
k = np.concatenate([[12 * r + 1, 12 * r + 2] for r in range(273)]); l = np.ones_like(k)
rep = SimpleNamespace(NSizeBWP=273, NStartBWP=0, PanelDimensions=(2, 1), CodebookMode=1, PMIMode="Subband", CQIMode="Subband", SubbandSize=16)
car = SimpleNamespace(NSizeGrid=273, NStartGrid=0, SymbolsPerSlot=14)
table = np.array([-3.46, 1.54, 6.54, 11.05, 13.54, 16.04, 17.54, 20.04, 22.04, 24.43, 26.93, 27.43, 29.43, 32.43, 35.43])
hd = ctx.to_device(np.asfortranarray(rng.standard_normal((k.size, 2, 4)) + 1j * rng.standard_normal((k.size, 2, 4))))
for nl in (1, 2):
    PL.cqiSelect(car, SimpleNamespace(k=k, l=l), rep, nl, hd, 0.05, table, ctx=ctx)
    t0 = time.perf_counter()
    for _ in range(20): PL.cqiSelect(car, SimpleNamespace(k=k, l=l), rep, nl, hd, 0.05, table, ctx=ctx)
    print(f"CSI report (PMI search over 32 entries x {k.size} REs + 18 subband CQIs), {nl} layer(s): {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per call (host wall)")
