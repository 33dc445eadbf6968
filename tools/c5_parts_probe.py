#!/usr/bin/env python3
"""Development probe: config-5 frame time part by part (which parts add up, which overlap) -- wall per frame over 3 frames for subsets of the frame's work."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG)
pool = bench.SlotPool(pkg, 0, 8)
ctxs = [pkg.Context(0) for _ in range(int(os.environ.get("ISAC_C5_CDL_CONTEXTS", "2")))]; ctx_csi = pkg.Context(0)
if bench.CommCell.SHARE:
    for c_ in ctxs:
        c_.set_cdl_share_spectra(True)
N = 21
sense = [bench.Cell(pkg, 0, c, 64, 16, 1, pool=pool, n_buf=1) for c in range(N)]
comm = [bench.CommCell(pkg, ctxs, ctx_csi, c, 64, 10) for c in range(N)]
def sync():
    pool.drain(); pool.sync(); ctx_csi.sync(); [c.sync() for c in ctxs]
def run(name, fn, frames=3):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(frames):
        fn()
    t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    print(f"{name:58s} wall {1e3 * (t2 - t0) / frames:7.1f} ms per frame (host done issuing after {1e3 * (t1 - t0) / frames:7.1f} ms)")
def dl_only():
    for cc in comm:
        ul, cc.WITH_UL = cc.WITH_UL, False
        cc.enqueue_frame(); cc.WITH_UL = ul
def ul_only():
    for cc in comm:
        for gi, g in enumerate(cc.groups):
            cc.CM.applyCDLBatch([cc.ul_chans[u] for _ in range(cc.UL_SLOTS) for u in g], [cc.ul_waves[u] for _ in range(cc.UL_SLOTS) for u in g], ctx=cc.ctxs[gi % len(cc.ctxs)], outs=cc.ul_rx[gi], gains=cc.ul_gains[gi])
def sensing_only():
    for sc in sense:
        pool.submit(sc)
def csi_only():
    for cc in comm:
        cc.csi_reports()
def cdl_all():
    for cc in comm:
        cc.enqueue_frame()
def everything():
    for sc, cc in zip(sense, comm):
        pool.submit(sc); cc.enqueue_frame()
    for cc in comm:
        cc.csi_reports(); cc.srs_reports()
def srs_only():
    for cc in comm:
        cc.srs_reports()
run("DL applies only (3 360 jobs, 84 calls)", dl_only)
run("UL applies only (840 jobs, 42 calls)", ul_only)
run("DL + UL applies", cdl_all)
run("sensing CPIs only (21)", sensing_only)
run("CSI estimates + reports only (840)", csi_only)
run("SRS estimates + reports only (210)", srs_only)
def no_reports():
    for sc, cc in zip(sense, comm):
        pool.submit(sc); cc.enqueue_frame()
def interleaved():
    for sc, cc in zip(sense, comm):
        pool.submit(sc); cc.enqueue_frame(); cc.csi_reports(); cc.srs_reports()
def reports_only():
    for cc in comm:
        cc.csi_reports(); cc.srs_reports()
run("sensing + DL + UL (no reports)", no_reports)
run("CSI + SRS reports only", reports_only)
run("whole frame, reports after all applies", everything)
run("whole frame, reports interleaved per cell", interleaved)
