#!/usr/bin/env python3
"""Config 5 of BASELINE.json ("full ISAC: SINR->CQI + mono-static sensing, 21 cells x 10 UE, 8 GPUs") assembled from the
seams this repository accelerates.  One process per GPU; cells are sharded round-robin (cell c -> rank c mod world); the
only collective is the final gather of the per-cell records (RCCL on GPUs).

    python examples/config5.py --cells 21 --ues 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/config5.py --cells 21 --ues 10

Per cell (simulation/cellSimulation.m order):
  1. layout: a seeded block of buildings around the gNB, UEs and targets dropped in the cell
  2. line of sight for every UE / target link        networkTopology.blockages.city.checkLoS      (los.hip)
  3. sensing CPI                                      sensing.monoStaticSensing -> estimation.fft2D (echo/rdm/music.hip)
  4. per UE: CDL-D (LoS) or CDL-A (NLoS) downlink channel over one slot   communication.channelModels (cdl.hip)
  5. per UE: CSI report of a 4-port CSI-RS channel estimate -- Type-I single-panel PMI search + wideband / subband CQI
     (uePhy.m:901-908 -> cqiSelect -> dlPMISelect, setupCSIRS.m:5-23 configuration)              communication.phyLayer (cqi.hip)
What is NOT here (out of scope): scheduler, HARQ, LDPC/PDSCH chain, rank selection (riSelect: one layer is reported); the
CSI-RS channel estimate is the channel's own frequency response at the CSI-RS resource elements (perfect estimation).

`PROBE = {"cell": c, "ue": u}` (set by tests) makes main() keep that UE's seam inputs / outputs in PROBE["capture"] for oracle checks.
"""
from __future__ import annotations

import argparse, importlib, json, os, sys, time
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (Cell = config-2 sensing set-up of one cell)


def cell_layout(cell_id, n_ues, n_targets):
    rng = np.random.default_rng(0xC0FFEE + cell_id)
    plans, heights = [], []
    for _ in range(12):                                     # a dozen blocks within 400 m of the gNB
        cx, cy = rng.uniform(-400, 400, 2)
        if np.hypot(cx, cy) < 40:
            continue
        w, d = rng.uniform(15, 40, 2)
        fp = np.array([[cx - w, cx + w, cx + w, cx - w, cx - w], [cy - d, cy - d, cy + d, cy + d, cy - d]])
        plans.append(fp); heights.append(float(rng.uniform(8, 35)))
    r, az = rng.uniform(30, 350, n_ues), rng.uniform(-np.pi, np.pi, n_ues)
    ue = np.stack([r * np.cos(az), r * np.sin(az), np.full(n_ues, 1.5)], axis=1)
    return plans, heights, ue


PROBE = None
DOWNLINK_SINR90PC = np.array([-3.46, 1.54, 6.54, 11.05, 13.54, 16.04, 17.54, 20.04, 22.04, 24.43, 26.93, 27.43, 29.43, 32.43, 35.43])   # setupSINRtoCQIMappingTable.m:7-11


csirs_positions, freq_response = bench.csirs_positions, bench.freq_response       # (shared with bench.py --workload config5)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=21)
    ap.add_argument("--ues", type=int, default=10)
    ap.add_argument("--ants", type=int, default=64)
    ap.add_argument("--slots", type=int, default=16)
    ap.add_argument("--targets", type=int, default=1)
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("ISAC_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank %= max(torch.cuda.device_count(), 1)
            dist.init_process_group(backend)
    pkg = importlib.import_module(bench.PKG)
    d = importlib.import_module(bench.PKG + "._dist")
    B = pkg.networkTopology.blockages
    CM, PL = pkg.communication.channelModels, pkg.communication.phyLayer
    mine = d.shard_cells(args.cells, rank, world)
    gnb = np.array([0.0, 0.0, 30.0])
    t = {"setup": 0.0, "los": 0.0, "sensing": 0.0, "cdl": 0.0, "cqi": 0.0}
    recs, extra = [], []
    t_all = time.perf_counter()
    pool = bench.SlotPool(pkg, local_rank, 1)               # one context for all cells of this rank (its scratch, tables and kernel attributes are set up once)
    for c in mine:
        t0 = time.perf_counter()
        cell = bench.Cell(pkg, local_rank, c, args.ants, args.slots, args.targets, pool=pool, n_buf=1)
        ctx = cell.ctx
        plans, heights, ue = cell_layout(c, args.ues, args.targets)
        town = B.city.from_floor_plans(plans, heights, ctx=ctx)
        t1 = time.perf_counter(); t["setup"] += t1 - t0
        # 2. LoS (networkSimulation.m:134-160)
        ue_los = town.checkLoS(ue, gnb)
        tgt_los = town.checkLoS(np.atleast_2d(cell.cellp.targetPosition), gnb) if hasattr(cell.cellp, "targetPosition") else np.ones(args.targets, bool)
        cell.los = np.asarray(tgt_los, dtype=np.uint8).reshape(-1)[: args.targets]
        t2 = time.perf_counter(); t["los"] += t2 - t1
        # 3. sensing CPI (cellSimulation.m:189-202); all targets blocked -> NaN like the reference's try/catch
        est = None
        if cell.los.any():
            try:
                est = cell.step()
            except pkg.IsacError:
                est = None
        ctx.sync()
        t3 = time.perf_counter(); t["sensing"] += t3 - t2
        # 4./5. per UE: CDL channel over one slot of the cell's downlink waveform, then wideband CQI
        profiles = CM.updateCDLModels(SimpleNamespace(ueLoSConditions=ue_los.astype(int), numUEs=args.ues))
        nt_shape = (args.ants // 16, 8, 2, 1, 1) if args.ants >= 16 else (1, args.ants // 2, 2, 1, 1)
        slot_T = 61440
        wave = ctx.empty((slot_T, args.ants))
        car = pkg._lib.Carrier(cell.K, 4096, 30, 0)
        grid = ctx.empty((cell.K, 14, args.ants))
        ctx.check(ctx.lib.isac_synth_qpsk_grid_dev(ctx.handle, bench.C.c_void_p(grid.ptr), cell.K, 14, args.ants, bench.C.c_uint64(0xD1 + c), 0))
        ctx.check(ctx.lib.isac_ofdm_modulate_dev(ctx.handle, bench.C.c_void_p(grid.ptr), 14, args.ants, bench.C.byref(car), bench.C.c_double(1.0),
                                                 bench.C.c_void_p(wave.ptr), bench.C.c_int64(slot_T)))
        cqis, pmis = [], []
        csi_k, csi_l = csirs_positions(273)
        report = SimpleNamespace(NSizeBWP=273, NStartBWP=0, PanelDimensions=(2, 1), CodebookMode=1, PMIMode="Subband", CQIMode="Subband",
                                 SubbandSize=16)                                            # setupCSIRS.m:17-23 (subbandSize.m: 16 or 32 at 273 PRBs)
        carrier = SimpleNamespace(NSizeGrid=273, NStartGrid=0, SymbolsPerSlot=14)
        for u in range(args.ues):
            t4 = time.perf_counter()
            ch = CM.CDLChannel(DelayProfile=profiles[u], TransmitAntennaArraySize=nt_shape, Seed=73)     # cdl.m:57-64 (same seed for every UE)
            t_ch = ch.time
            rx = CM.applyCDL(ch, wave, ctx=ctx)                  # rxWaveform [T x 2] stays on the device (the UE PHY is out of scope)
            ctx.sync()
            t5 = time.perf_counter(); t["cdl"] += t5 - t4
            hf = freq_response(ch, csi_k, cell.K, 30e3, 4)      # [nRE x nRx x 4 ports]
            dist_m = float(np.linalg.norm(ue[u] - gnb))
            pl_db = 32.4 + 20.0 * np.log10(3.5) + 30.0 * np.log10(max(dist_m, 10.0))       # distance-only path loss, 3.5 GHz
            noise_dbm = -174.0 + 10.0 * np.log10(100e6) + 7.0                               # 100 MHz, 7 dB noise figure
            nvar = 10.0 ** (-(46.0 - pl_db - noise_dbm) / 10.0)                             # unit-power channel at 46 dBm
            cqi, pmi, cinfo, _ = PL.cqiSelect(carrier, SimpleNamespace(k=csi_k, l=csi_l), report, 1, ctx.to_device(hf), nvar, DOWNLINK_SINR90PC, ctx=ctx)
            t["cqi"] += time.perf_counter() - t5
            cqis.append(None if np.isnan(cqi[0]) else int(cqi[0]))
            pmis.append([None if np.isnan(v) else int(v) for v in pmi.i1])
            if PROBE is not None and PROBE.get("cell") == c and PROBE.get("ue") == u:
                PROBE["capture"] = dict(profile=profiles[u], tx_size=nt_shape, t0=t_ch, wave=wave.numpy(), rx=rx.numpy() if hasattr(rx, "numpy") else np.asarray(rx),
                                        hf=hf, nvar=nvar, report=report, csi_k=csi_k, csi_l=csi_l, cqi=cqi, pmi=pmi, subband_cqi=cinfo.SubbandCQI,
                                        ue_los=bool(ue_los[u]), est=est)
        recs.append(d.make_record(c, est, time.perf_counter() - t0))
        extra.append({"cell": c, "ue_los": int(ue_los.sum()), "cqi": cqis, "pmi_i1": pmis, "n_walls": town._tab().n_walls})
    recs = np.array(recs).reshape(-1, d.RECORD_LEN)
    on_gpu = dist is not None and dist.get_backend() == "nccl"
    allr = d.gather_records(recs, dist, torch.device("cuda", local_rank) if on_gpu else None)
    wall = time.perf_counter() - t_all
    if rank == 0:
        out = {"config": f"{args.cells} cells x {args.ues} UE, {args.ants}-antenna gNB, {args.slots} sensing slots per CPI, {world} GPU(s)",
               "wall_s": round(wall, 3), "rank0_stage_s": {k: round(v, 3) for k, v in t.items()},
               "cells": [{"cell": int(r[0]), "nRng": None if np.isnan(r[1]) else int(r[1]), "rngEst0": None if np.isnan(r[2]) else round(float(r[2]), 3),
                          "velEst0": None if np.isnan(r[3]) else round(float(r[3]), 3), "aziEst0": None if np.isnan(r[4]) else float(r[4])} for r in allr],
               "rank0_ues": extra}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
