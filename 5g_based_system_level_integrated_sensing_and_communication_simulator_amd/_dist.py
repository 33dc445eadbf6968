"""Multi-GPU plumbing for the embarrassingly parallel cell loop (networkSimulation.m:44-60: one worker
per cell).  One process per GPU; cells are sharded round-robin; the only collective is one all-gather
of fixed-size per-cell result records at the end (RCCL over xGMI on GPUs, gloo in the CPU tests).

The record carries what the reference returns per cell -- the whole estResults struct of fft2D.m:102,114-115 (every range / velocity /
azimuth estimate, not the first one) -- and, for the full-ISAC workload, what networkSimulation.m:173-232 collects per UE from the
communication side (wideband CQI, subband CQIs, PMI i1 / i2): fixed capacities with the TRUE counts in front, so that one all_gather
of equal-sized buffers moves it and a reader can tell a truncated list from a complete one."""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

EST_CAP = 64     # estimates kept per list (counts in the header are the true ones)
UE_CAP = 16      # UEs kept per cell
SB_CAP = 32      # subbands kept per UE (273 PRB / 16-PRB subbands = 18)
UE_LEN = 8 + 4 * SB_CAP   # [ue, wideband CQI, i1 (3), nSB, subband CQI [SB_CAP], subband i2 [SB_CAP], RI, nSB of the SRS report, UL TPMI [SB_CAP], UL CQI [SB_CAP]]
_UE_RI = 6 + 2 * SB_CAP   # round 6: the rank riSelect picked (uePhy.m:900) and the gNB's SRS measurement of the UE (gNBPhy.m:1033-1058: TPMI and CQI per SRS subband)
HDR_LEN = 12     # [cellID, nRng, rngEst0, velEst0, aziEst0, nDetTotal, elapsed_s, valid, nVel, nAzi, nUE, reserved]
RECORD_LEN = HDR_LEN + 3 * EST_CAP + UE_CAP * UE_LEN
_OFF_RNG, _OFF_VEL, _OFF_AZI, _OFF_UE = HDR_LEN, HDR_LEN + EST_CAP, HDR_LEN + 2 * EST_CAP, HDR_LEN + 3 * EST_CAP


def shard_cells(n_cells: int, rank: int, world: int) -> list[int]:
    """cell c -> rank (c mod world)   (SURVEY.md 8e)."""
    return [c for c in range(n_cells) if c % world == rank]


def _ue_block(u: int, report, rank=None, srs=None, srs_band: int = 0) -> np.ndarray:
    """report = (cqi, pmi, info) as communication.phyLayer.cqiSelect returns it: cqi [1 + nSB x nCW] (row 0 wideband), pmi.i1 (3), pmi.i2 [1 + nSB];
    rank = riSelect's RI; srs = (pmi [nSB], sinrSubbandPMI, cqiRBs) as communication.phyLayer.srsReportBatch returns it (cqiRBs is constant over a subband of
    srs_band PRBs: the value at each subband's first PRB is kept)."""
    b = np.full(UE_LEN, np.nan)
    b[0] = u
    if rank is not None:
        b[_UE_RI] = rank
    if srs is not None:
        tp, cq = np.asarray(srs[0], dtype=np.float64).reshape(-1), np.asarray(srs[2], dtype=np.float64).reshape(-1)
        b[_UE_RI + 1] = tp.size
        b[_UE_RI + 2:_UE_RI + 2 + min(tp.size, SB_CAP)] = tp[:SB_CAP]
        per_sb = cq[::max(int(srs_band), 1)][:tp.size]
        b[_UE_RI + 2 + SB_CAP:_UE_RI + 2 + SB_CAP + min(per_sb.size, SB_CAP)] = per_sb[:SB_CAP]
    if report is None:
        return b
    cqi, pmi = report[0], report[1]
    c = np.asarray(cqi, dtype=np.float64)
    c = c.reshape(c.shape[0], -1)[:, 0] if c.ndim > 1 else c.reshape(-1)
    b[1] = c[0] if c.size else np.nan
    i1 = np.asarray(getattr(pmi, "i1", []), dtype=np.float64).reshape(-1)[:3]
    b[2:2 + i1.size] = i1
    sb = c[1:]
    b[5] = sb.size
    b[6:6 + min(sb.size, SB_CAP)] = sb[:SB_CAP]
    i2 = np.asarray(getattr(pmi, "i2", []), dtype=np.float64).reshape(-1)
    i2 = i2[1:] if i2.size == sb.size + 1 else i2                # (row 0 = wideband entry where present)
    b[6 + SB_CAP:6 + SB_CAP + min(i2.size, SB_CAP)] = i2[:SB_CAP]
    return b


def make_record(cell_id: int, est, elapsed_s: float = 0.0, ue_reports=None, ue_ranks=None, ue_srs=None, srs_band: int = 0) -> np.ndarray:
    """Fixed-size record of one cell: its estResults (NaN where the reference would return NaN: cellSimulation.m:196-202) and, when given,
    the last CSI report of each of its UEs (list of (cqi, pmi, info) or None per UE), the rank of each report (ue_ranks) and the gNB's last SRS
    measurement of each UE (ue_srs: list of (pmi, sinrSubbandPMI, cqiRBs), srs_band = SRS subband size in PRBs)."""
    r = np.full(RECORD_LEN, np.nan)
    r[0] = cell_id
    r[6] = elapsed_s
    r[7] = 0.0
    r[8] = r[9] = r[10] = 0.0
    if est is not None:
        rng, vel, azi = (np.asarray(getattr(est, k), dtype=np.float64).reshape(-1) for k in ("rngEst", "velEst", "aziEst"))
        r[1] = rng.size
        r[2] = rng[0] if rng.size else np.nan
        r[3] = vel[0] if vel.size else np.nan
        r[4] = azi[0] if azi.size else np.nan
        r[5] = getattr(est, "total_detections", np.nan)
        r[7] = 1.0
        r[8], r[9] = vel.size, azi.size
        r[_OFF_RNG:_OFF_RNG + min(rng.size, EST_CAP)] = rng[:EST_CAP]
        r[_OFF_VEL:_OFF_VEL + min(vel.size, EST_CAP)] = vel[:EST_CAP]
        r[_OFF_AZI:_OFF_AZI + min(azi.size, EST_CAP)] = azi[:EST_CAP]
    if ue_reports is not None:
        r[10] = len(ue_reports)
        for u, rep in enumerate(ue_reports[:UE_CAP]):
            r[_OFF_UE + u * UE_LEN:_OFF_UE + (u + 1) * UE_LEN] = _ue_block(u, rep, None if ue_ranks is None else ue_ranks[u], None if ue_srs is None else ue_srs[u], srs_band)
    return r


def unpack_record(r: np.ndarray) -> SimpleNamespace:
    """The record as a namespace: cell, valid, rngEst / velEst / aziEst (as far as kept) with their true counts, elapsed_s, ues = [namespace(ue, cqi, i1, sbCQI, sbI2, ri, ulTPMI, ulCQI)]."""
    r = np.asarray(r, dtype=np.float64).reshape(-1)
    n = lambda v: 0 if np.isnan(v) else int(v)                    # noqa: E731
    n_rng, n_vel, n_azi, n_ue = n(r[1]), n(r[8]), n(r[9]), n(r[10])
    ues = []
    for u in range(min(n_ue, UE_CAP)):
        b = r[_OFF_UE + u * UE_LEN:_OFF_UE + (u + 1) * UE_LEN]
        n_sb = min(n(b[5]), SB_CAP)
        n_ul = min(n(b[_UE_RI + 1]), SB_CAP)
        ints = lambda a: [None if np.isnan(v) else int(v) for v in a]     # noqa: E731
        ues.append(SimpleNamespace(ue=n(b[0]), cqi=None if np.isnan(b[1]) else int(b[1]), i1=ints(b[2:5]), sbCQI=ints(b[6:6 + n_sb]), sbI2=ints(b[6 + SB_CAP:6 + SB_CAP + n_sb]),
                                   ri=None if np.isnan(b[_UE_RI]) else int(b[_UE_RI]), ulTPMI=ints(b[_UE_RI + 2:_UE_RI + 2 + n_ul]),
                                   ulCQI=ints(b[_UE_RI + 2 + SB_CAP:_UE_RI + 2 + SB_CAP + n_ul])))
    return SimpleNamespace(cell=int(r[0]), valid=bool(r[7] == 1.0), nRng=n_rng, nVel=n_vel, nAzi=n_azi, nDetTotal=None if np.isnan(r[5]) else int(r[5]), elapsed_s=float(r[6]),
                           rngEst=r[_OFF_RNG:_OFF_RNG + min(n_rng, EST_CAP)].copy(), velEst=r[_OFF_VEL:_OFF_VEL + min(n_vel, EST_CAP)].copy(),
                           aziEst=r[_OFF_AZI:_OFF_AZI + min(n_azi, EST_CAP)].copy(), nUE=n_ue, ues=ues)


def record_json(r: np.ndarray) -> dict:
    """JSON-able form of a record (bench.py's `cells` list): every estimate of the cell and every UE's CQI / PMI."""
    u = unpack_record(r)
    d = {"cell": u.cell, "valid": u.valid, "nRng": u.nRng if u.valid else None, "nVel": u.nVel if u.valid else None, "nAzi": u.nAzi if u.valid else None, "nDetTotal": u.nDetTotal,
         "rngEst": [round(float(v), 6) for v in u.rngEst], "velEst": [round(float(v), 6) for v in u.velEst], "aziEst": [float(v) for v in u.aziEst]}
    if u.nUE:
        d["ues"] = [{"ue": x.ue, "cqi": x.cqi, "i1": x.i1, "sbCQI": x.sbCQI, "sbI2": x.sbI2, **({"ri": x.ri} if x.ri is not None else {}),
                     **({"ulTPMI": x.ulTPMI, "ulCQI": x.ulCQI} if x.ulTPMI else {})} for x in u.ues]
    return d


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
