"""Multi-GPU plumbing for the embarrassingly parallel cell loop (networkSimulation.m:44-60: one worker
per cell).  One process per GPU; cells are sharded round-robin; the only collective is one all-gather
of fixed-size per-cell result records at the end (RCCL over xGMI on GPUs, gloo in the CPU tests)."""
from __future__ import annotations

import numpy as np

RECORD_LEN = 8   # [cellID, nRng, rngEst0, velEst0, aziEst0, nDetTotal, elapsed_s, valid]


def shard_cells(n_cells: int, rank: int, world: int) -> list[int]:
    """cell c -> rank (c mod world)   (SURVEY.md 8e)."""
    return [c for c in range(n_cells) if c % world == rank]


def make_record(cell_id: int, est, elapsed_s: float = 0.0) -> np.ndarray:
    """Fixed-size record of one cell's estResults (NaN where the reference would return NaN)."""
    r = np.full(RECORD_LEN, np.nan)
    r[0] = cell_id
    r[6] = elapsed_s
    r[7] = 0.0
    if est is not None:
        r[1] = est.rngEst.size
        r[2] = est.rngEst[0] if est.rngEst.size else np.nan
        r[3] = est.velEst[0] if est.velEst.size else np.nan
        r[4] = est.aziEst[0] if est.aziEst.size else np.nan
        r[5] = getattr(est, "total_detections", np.nan)
        r[7] = 1.0
    return r


def gather_records(records: np.ndarray, dist=None, device=None) -> np.ndarray:
    """All ranks' records, ordered by cell id.  `records` is [n_local x RECORD_LEN] (n_local may differ
    between ranks: 7 cells on 2/4/8 GPUs); padded to the maximum count for one all_gather."""
    records = np.asarray(records, dtype=np.float64).reshape(-1, RECORD_LEN)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        out = records
    else:
        import torch
        world = dist.get_world_size()
        n = torch.tensor([records.shape[0]], dtype=torch.int64, device=device)
        counts = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(counts, n)
        n_max = max(int(c.item()) for c in counts)
        buf = torch.full((max(n_max, 1), RECORD_LEN), float("nan"), dtype=torch.float64, device=device)
        if records.shape[0]:
            buf[: records.shape[0]] = torch.as_tensor(records, dtype=torch.float64, device=device)
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
        out = np.concatenate([p[: int(c.item())].cpu().numpy() for p, c in zip(parts, counts)], axis=0) if n_max else records
    if out.shape[0]:
        out = out[np.argsort(out[:, 0], kind="stable")]
    return out
