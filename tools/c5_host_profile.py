#!/usr/bin/env python3
"""Development probe: where does the HOST spend a config-5 frame?  Builds bench.py's config-5 objects for --cells cells, runs frames under cProfile and prints the
wall per frame, the time until every call of the frame has been issued, and the top host functions.   python tools/c5_host_profile.py [--cells 21]"""
import argparse, cProfile, importlib, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--cells", type=int, default=21); ap.add_argument("--frames", type=int, default=3)
a = ap.parse_args()
pkg = importlib.import_module(bench.PKG)
pool = bench.SlotPool(pkg, 0, 8)
ctxs = [pkg.Context(0) for _ in range(2)]; ctx_csi = pkg.Context(0)
sense = [bench.Cell(pkg, 0, c, 64, 16, 1, pool=pool, n_buf=1) for c in range(a.cells)]
comm = [bench.CommCell(pkg, ctxs, ctx_csi, c, 64, 10) for c in range(a.cells)]
def sync():
    pool.sync(); ctx_csi.sync(); [c.sync() for c in ctxs]
def frame(parts):
    t0 = time.perf_counter()
    for sc, cc in zip(sense, comm):
        pool.submit(sc); cc.enqueue_frame()
    t1 = time.perf_counter()
    for cc in comm:
        cc.csi_reports()
    t2 = time.perf_counter()
    parts.append((t1 - t0, t2 - t1))
parts = []
frame(parts); pool.drain(); sync()
parts = []
pr = cProfile.Profile()
t0 = time.perf_counter(); pr.enable()
for _ in range(a.frames):
    frame(parts)
pr.disable(); t_issue = time.perf_counter() - t0
pool.drain(); sync()
wall = time.perf_counter() - t0
print(f"{a.cells} cells, {a.frames} frames: wall {1e3 * wall / a.frames:.1f} ms per frame; host done issuing after {1e3 * t_issue / a.frames:.1f} ms per frame "
      f"(sensing submit + CDL calls {1e3 * sum(p[0] for p in parts) / a.frames:.1f} ms, CSI estimates + reports incl. their synchronisations {1e3 * sum(p[1] for p in parts) / a.frames:.1f} ms)")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:4000])
