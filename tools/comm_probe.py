#!/usr/bin/env python3
"""Development probe of the communication seams at the bench numerology (61 440-sample slot at 122.88 MHz):
   * applyCDL, one job (host path gains) and applyCDLBatch, n UEs x one DL waveform (device path gains), DL 64 -> 2 and UL 2 -> 64:
     GPU time per call from HIP events (everything the call enqueues), host wall per call;
   * roofline terms of the DL apply: 3M flops on MFMA and the bytes of x + Z (written, read) + y.
   python tools/comm_probe.py [--ues 10]"""
import argparse, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--ues", type=int, default=10); ap.add_argument("--reps", type=int, default=10); ap.add_argument("--batch-only", action="store_true", help="downlink batches only (kernel traces: nothing but the batched launches)")
args = ap.parse_args()
pkg = importlib.import_module(bench.PKG)
CM = pkg.communication.channelModels
ctx = pkg.Context(0)
T = 61440
rng = np.random.default_rng(1)
def timed(fn, reps=args.reps):
    fn(); ctx.sync()
    g, w = [], []
    for _ in range(reps):
        ctx.sync(); ctx.timer_start(); t0 = time.perf_counter(); fn(); w.append(1e3 * (time.perf_counter() - t0)); g.append(ctx.timer_stop_ms())
    return float(np.min(g)), float(np.median(g)), float(np.median(w))
for name, txs, rxs in (("DL 64 -> 2", (4, 8, 2, 1, 1), (1, 1, 2, 1, 1)), ("UL 2 -> 64", (1, 1, 2, 1, 1), (4, 8, 2, 1, 1)))[:1 if args.batch_only else 2]:
    nt, nr = int(np.prod(txs)), int(np.prod(rxs))
    x = ctx.to_device(np.asfortranarray(rng.standard_normal((T, nt)) + 1j * rng.standard_normal((T, nt))))
    for prof in ("CDL-D", "CDL-A"):
        ch = CM.CDLChannel(DelayProfile=prof, TransmitAntennaArraySize=txs, ReceiveAntennaArraySize=rxs)
        n_paths = ch.path_delays().size
        mn, med, wall = (0.0, 0.0, 0.0) if args.batch_only else timed(lambda: CM.applyCDL(ch, x, ctx=ctx))
        chans = [CM.CDLChannel(DelayProfile=prof, TransmitAntennaArraySize=txs, ReceiveAntennaArraySize=rxs) for _ in range(args.ues)]
        outs = [ctx.empty((T, nr)) for _ in chans]
        bmn, bmed, bwall = timed(lambda: CM.applyCDLBatch(chans, [x] * args.ues, ctx=ctx, outs=outs))
        nc = n_paths * min(nt, nr)
        flops = 6.0 * T * nc * max(nt, nr) * (1 if nt >= nr else 1)          # 3M form: 3 real products x 2 flops per complex MAC
        byts = 16.0 * T * (nt + nr + 2 * nc)
        print(f"{name} {prof} ({n_paths} paths): one job gpu {mn:.3f} ms (median {med:.3f}), host wall {wall:.3f} ms | batch of {args.ues}: gpu {bmn:.3f} ms = {bmn / args.ues:.4f} ms per job, "
              f"host wall {bwall:.3f} ms | per job: {flops / 1e9:.2f} GF issued (3M) -> {flops / 1e9 / (bmn / args.ues):.1f} TF/s = {flops / 1e9 / (bmn / args.ues) / 78.6e3 * 1e3:.3f} of the fp64 MFMA peak; "
              f"{byts / 1e6:.0f} MB (x + y + Z written and read) -> {byts / 1e9 / (bmn / args.ues) :.2f} TB/s")
