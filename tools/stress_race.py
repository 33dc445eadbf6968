#!/usr/bin/env python3
"""Stress harness for the two single-case mismatches of the round-5 fuzz campaign (profiles/r05_fuzz_campaigns.txt):

  (i)  a host-array fft2D call whose |rdm|^2 window came back off by 1e-2 once in ~4 700 cases   -> attributed to hipMemcpy on the NULL stream racing the context's
       non-blocking streams, fixed by upload_now (csrc/isac_common.hpp);
  (ii) a cqiFromChannel call whose MEAN came back 60.88 instead of 69.06 once                      -> unexplained.

N worker processes on ONE GPU, each with its own context, beside B CPU burner processes (host oversubscription, as in the 16-process xdist campaign), repeat
  * cqiFromChannel / isac_precoded_sinr_cqi_dev on a FIXED channel estimate -- alternating the library-scratch form (per-RE values in ctx scratch, as the fuzz test called it)
    with the caller-buffer form (per-RE values in a caller DeviceArray), and
  * the host-array fft2D call on a FIXED scene (the scene of fuzz seed 1045: 8 antennas, 24 PRB, 4 slots)
and compare EVERY result with the first call's, bit for bit.  On a mismatch the worker classifies it: for the SINR mean it re-reads the per-RE device buffer and recomputes the mean
on the host (kernel output wrong vs reduction / copy-back wrong) and repeats the call; for fft2D it repeats the call (transient vs persistent) and reports which fields moved.

  python tools/stress_race.py --procs 16 --burners 16 --sinr-calls 20000 --fft-calls 3000 [--lib tools/_ab/libisac_hip_prefix.so] [--out gpurun_out/r06_stress.txt]

--lib swaps another build of the library in for the duration of the run (A/B against the pre-fix library: tools/build_prefix_variant.sh)."""
from __future__ import annotations

import argparse
import ctypes as C
import importlib
import json
import multiprocessing as mp
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "5g_based_system_level_integrated_sensing_and_communication_simulator_amd"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def burner(stop, n_threads=1):
    """One burner process: n_threads threads of NumPy work that releases the GIL (element-wise passes over a few MB) -- `--burners B --burner-threads t` puts B x t runnable
    threads beside the workers (the campaign that saw the mismatches ran 16 xdist workers with 256 FFT threads each on this 256-CPU host)."""
    import threading

    def spin(seed):
        x = np.random.default_rng(seed).standard_normal(1 << 18)
        while not stop.is_set():
            x = np.tanh(x * 1.0001 + 1e-3)

    th = [threading.Thread(target=spin, args=(os.getpid() * 131 + i,), daemon=True) for i in range(max(1, n_threads))]
    for t in th:
        t.start()
    for t in th:
        t.join()


def worker(rank, args, q):
    try:
        q.put(_worker(rank, args))
    except Exception as e:                                     # a crashed worker must not hang the parent
        q.put({"rank": rank, "error": repr(e)})


def _worker(rank, args):
    pkg = importlib.import_module(PKG)
    L = pkg._lib
    ctx = L.default_context()
    PL = pkg.communication.phyLayer
    rng = np.random.default_rng(9500 + 1036)                   # the configuration of the fuzz case that returned the wrong mean
    nl, p, nr, n_re = 2, 8, 4, 2000 + 37 * (rank % 5)
    h = np.asfortranarray((rng.standard_normal((n_re, nr, p)) + 1j * rng.standard_normal((n_re, nr, p))) * 3.0)
    w, _ = np.linalg.qr(rng.standard_normal((p, nl)) + 1j * rng.standard_normal((p, nl)))
    w = np.asfortranarray(w / np.sqrt(nl))
    sigma = 0.7
    table = np.ascontiguousarray(PL.precodedSINR.__globals__["DOWNLINK_SINR90PC"], dtype=np.float64)
    out = {"rank": rank, "sinr_calls": 0, "sinr_mismatch": [], "fft_calls": 0, "fft_mismatch": []}

    def sinr_call(host_h, per):
        """One isac_precoded_sinr_cqi_dev call; host_h: upload the channel estimate through a fresh DeviceArray (alloc + pageable copy + free, as the fuzz did)."""
        d_h = ctx.to_device(h) if host_h else d_h_keep
        mean, cqi = C.c_double(0), C.c_int32(0)
        ctx.check(ctx.lib.isac_precoded_sinr_cqi_dev(ctx.handle, C.c_void_p(d_h.ptr), C.c_int64(n_re), C.c_int32(nr), C.c_int32(p), w.ctypes.data_as(C.c_void_p),
                                                     C.c_int32(nl), C.c_double(sigma), table.ctypes.data_as(C.c_void_p), C.c_int32(table.size),
                                                     C.c_void_p(per.ptr if per is not None else 0), C.byref(mean), C.byref(cqi)))
        return mean.value, cqi.value

    crng = np.random.default_rng(77 + rank)

    def churn():
        if not args.churn:
            return
        n = int(crng.integers(1, 1 << 16))
        junk = crng.standard_normal(n) + 1j * crng.standard_normal(n)
        d = ctx.to_device(junk)
        np.fft.fft(junk)                                        # a little host work between the library calls
        d.free()

    d_h_keep = ctx.to_device(h)
    per = ctx.empty((n_re,), np.float64)
    mean0, cqi0 = sinr_call(True, per)
    per0 = per.numpy()
    host_mean0 = float(per0.sum() / n_re)
    t0 = time.perf_counter()
    for i in range(args.sinr_calls):
        form = i % 3                                           # 0: library scratch + fresh upload (the fuzz's form), 1: caller buffer + fresh upload, 2: caller buffer, resident H
        churn()
        m, c = sinr_call(form != 2, None if form == 0 else per)
        out["sinr_calls"] += 1
        if m != mean0 or c != cqi0:
            rec = {"i": i, "form": form, "mean": m, "want": mean0, "cqi": c}
            if form != 0:
                now = per.numpy()
                rec["per_re_equal_first"] = bool(np.array_equal(now, per0))
                rec["per_re_bad"] = int((now != per0).sum())
                rec["host_mean_of_device_per_re"] = float(now.sum() / n_re)
            m2, c2 = sinr_call(form != 2, None if form == 0 else per)
            rec["repeat_mean"] = m2
            out["sinr_mismatch"].append(rec)
    out["sinr_s"] = time.perf_counter() - t0

    # ---- host-array fft2D on the scene of fuzz seed 1045
    from test_gpu_fuzz import _scene
    import oracle as O
    sc, los = _scene(1045)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    echo = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, los, sc.noise, nfft=sc.wave.Nfft)
    est0, dbg0 = pkg.sensing.estimation.fft2D(rp, cf, echo, sc.tx_grid, return_debug=True)

    def same(est, dbg):
        bad = []
        if not np.array_equal(dbg.power_window, dbg0.power_window):
            bad.append("power_window(%d entries, max rel %.2e)" % (int((dbg.power_window != dbg0.power_window).sum()),
                                                                  float(np.abs(dbg.power_window - dbg0.power_window).max() / np.abs(dbg0.power_window).max())))
        if not np.array_equal(dbg.Ra, dbg0.Ra):
            bad.append("Ra")
        if not all(np.array_equal(a, b) for a, b in zip(dbg.detections, dbg0.detections)):
            bad.append("detections")
        for f in ("rngEst", "velEst", "aziEst"):
            if not np.array_equal(getattr(est, f), getattr(est0, f)):
                bad.append(f)
        return bad

    t0 = time.perf_counter()
    for i in range(args.fft_calls):
        churn()
        est, dbg = pkg.sensing.estimation.fft2D(rp, cf, echo, sc.tx_grid, return_debug=True)
        out["fft_calls"] += 1
        bad = same(est, dbg)
        if bad:
            est2, dbg2 = pkg.sensing.estimation.fft2D(rp, cf, echo, sc.tx_grid, return_debug=True)
            out["fft_mismatch"].append({"i": i, "fields": bad, "repeat_fields": same(est2, dbg2)})
    out["fft_s"] = time.perf_counter() - t0
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=16)
    ap.add_argument("--burners", type=int, default=16)
    ap.add_argument("--burner-threads", type=int, default=1)
    ap.add_argument("--churn", type=int, default=0, help="1: a random-size pageable upload + free (and a little host FFT work) between the calls, as a test process does")
    ap.add_argument("--sinr-calls", type=int, default=20000)
    ap.add_argument("--fft-calls", type=int, default=3000)
    ap.add_argument("--lib", default=None, help="another libisac_hip.so to run with (swapped in for the run, restored afterwards)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--tag", default="tree")
    args = ap.parse_args()
    lib = os.path.join(ROOT, PKG, "libisac_hip.so")
    keep = None
    if args.lib:
        keep = lib + ".stress_keep"
        shutil.copy2(lib, keep)
        shutil.copy2(args.lib, lib)
    try:
        mp.set_start_method("spawn")
        stop = mp.Event()
        burners = [mp.Process(target=burner, args=(stop, args.burner_threads), daemon=True) for _ in range(args.burners)]
        for b in burners:
            b.start()
        q = mp.Queue()
        t0 = time.perf_counter()
        procs = [mp.Process(target=worker, args=(r, args, q)) for r in range(args.procs)]
        for pr in procs:
            pr.start()
        res = [q.get() for _ in procs]
        for pr in procs:
            pr.join()
        wall = time.perf_counter() - t0
        stop.set()
        for b in burners:
            b.join(timeout=5)
            if b.is_alive():
                b.terminate()
    finally:
        if keep:
            shutil.move(keep, lib)
    errs = [r for r in res if "error" in r]
    ok = [r for r in res if "error" not in r]
    n_s, n_f = sum(r["sinr_calls"] for r in ok), sum(r["fft_calls"] for r in ok)
    mm_s = [dict(m, rank=r["rank"]) for r in ok for m in r["sinr_mismatch"]]
    mm_f = [dict(m, rank=r["rank"]) for r in ok for m in r["fft_mismatch"]]
    # one-sided 95 % bound on the per-call rate when nothing was seen: 3 / n
    summ = {"tag": args.tag, "lib": args.lib or "tree", "procs": args.procs, "burners": args.burners, "burner_threads": args.burner_threads, "churn": args.churn, "host_cpus": os.cpu_count(), "wall_s": round(wall, 1),
            "sinr_calls": n_s, "sinr_mismatches": len(mm_s), "sinr_rate_bound_95": (3.0 / n_s if n_s and not mm_s else None),
            "fft2d_host_calls": n_f, "fft2d_mismatches": len(mm_f), "fft2d_rate_bound_95": (3.0 / n_f if n_f and not mm_f else None),
            "worker_errors": errs, "sinr_detail": mm_s[:20], "fft2d_detail": mm_f[:20]}
    line = json.dumps(summ)
    print(line)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "a") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
