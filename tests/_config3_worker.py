"""Worker of tests/test_gpu_dist.py::test_config3_seven_cells_at_size_against_oracle: one rank of BASELINE configs[2] (7 cells, each
= config 2: A = 64, 16 slots, 273 PRB; cell c -> rank c mod world, networkSimulation.m:44-60).  Every rank builds the seeded scenes of
ITS cells (tests/golden/make_golden.py: cell7_kwargs / cell7_spectral_noise -- the inputs the committed oracle fixture was generated
from), runs the bench's own device path on them (per-target demodulation -> fused synthesis + range kernel with the injected AWGN on
the demodulated grid -> cached-range fft2D), and the ranks exchange the per-cell records through the product's one collective
(_dist.gather_records; gloo here, RCCL on a multi-GPU node).  Rank 0 prints one JSON line.  TEST INFRASTRUCTURE."""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    import torch
    import torch.distributed as dist
    from conftest import load_pkg, make_scene
    from make_golden import CELLS7, cell7_kwargs, cell7_spectral_noise, detection_digest
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("ISAC_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if world > 1:
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    pkg = load_pkg()
    d = __import__("importlib").import_module(pkg.__name__ + "._dist")
    ctx = pkg.Context(local_rank % max(n_dev, 1))
    cells = [int(c) for c in os.environ.get("ISAC_CONFIG3_CELLS", ",".join(str(c) for c in range(len(CELLS7)))).split(",")]
    mine = [cells[i] for i in d.shard_cells(len(cells), rank, world)]
    recs, full = [], {}
    for c in mine:
        sc = make_scene(**cell7_kwargs(c))
        w = cell7_spectral_noise(c, sc.K, sc.L, sc.A)
        rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
        cf = pkg.sensing.detection.cfar2D(rp)
        d_wave, d_txg, d_w = ctx.to_device(sc.tx_wave), ctx.to_device(sc.tx_grid), ctx.to_device(w)
        echo = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, spectral_noise=d_w, fuse_fft2d=(rp, cf, d_txg))
        est, dbg = pkg.sensing.estimation.fft2D(rp, cf, echo, d_txg, return_debug=True, reuse_range=True)
        est.total_detections = sum(x.shape[1] for x in dbg.detections)
        recs.append(d.make_record(c, est))
        full[c] = dict(rngEst=est.rngEst.tolist(), velEst=est.velEst.tolist(), aziEst=est.aziEst.tolist(), det_sha256=detection_digest(dbg.detections),
                       n_det=int(est.total_detections), ra_trace=float(np.trace(dbg.Ra).real), rank=rank)
        del d_wave, d_txg, d_w, echo
    on_gpu = world > 1 and backend == "nccl"
    allr = d.gather_records(np.array(recs).reshape(-1, d.RECORD_LEN), dist if world > 1 else None, torch.device("cuda", local_rank) if on_gpu else None)
    fulls = [full]
    if world > 1:
        fulls = [None] * world
        dist.all_gather_object(fulls, full)              # test-only: the complete estimate lists (the product gathers fixed-size records)
    if rank == 0:
        merged = {}
        for f in fulls:
            merged.update({str(k): v for k, v in f.items()})
        print(json.dumps({"world": world, "backend": backend if world > 1 else None, "records": allr.tolist(), "cells": merged}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
