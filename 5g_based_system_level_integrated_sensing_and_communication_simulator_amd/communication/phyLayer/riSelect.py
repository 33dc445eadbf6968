"""``communication.phyLayer.riSelect`` (+communication/+phyLayer/riSelect.m:207-292) for the Type-I single-panel codebook as uePhy.m:900 calls it:
one CSI report per valid rank (isac_csi_report_batch_dev -- every rank's PMI search on the GPU), the rank whose totalSINR (isac_csi_report.ri_total_sinr,
riSelect.m:253-276) beats the best so far by more than 0.1."""
from __future__ import annotations

import numpy as np

from .csiReport import cqiSelectBatch


def riSelectBatch(carrier, csirs, reportConfig, H_list, nVar_list, SINRTable, *, ctx=None):
    """riSelect + the cqiSelect of uePhy.m:900-908 for many UEs that share the CSI-RS / report configuration.
    Returns per UE (RI or NaN, CQI, PMISet, CQIInfo) -- the report at the selected rank (at the last valid rank when every totalSINR is NaN)."""
    H_list = list(H_list)
    if not H_list:
        return []
    _, nr, p = H_list[0].shape
    max_rank = min(nr, p)                                                 # riSelect.m:219-220
    restr = np.asarray(getattr(reportConfig, "RIRestriction", np.ones(8))).reshape(-1)
    valid = [r for r in range(1, max_rank + 1) if r <= restr.size and restr[r - 1]]
    if not valid:
        raise ValueError("riSelectBatch: RIRestriction leaves no valid rank")
    per_rank = {r: cqiSelectBatch(carrier, csirs, reportConfig, r, H_list, nVar_list, SINRTable, ctx=ctx, with_ri_total=True) for r in valid}
    out = []
    for u in range(len(H_list)):
        best, ri = -np.inf, np.nan
        for r in valid:
            tot = per_rank[r][u][3]
            if tot > best + 0.1:                                          # riSelect.m:278-282 (NaN compares false)
                best, ri = tot, r
        rep = per_rank[ri if not np.isnan(ri) else valid[-1]][u]
        out.append((ri, rep[0], rep[1], rep[2]))
    return out


def riSelect(carrier, csirs, reportConfig, H, nVar=1e-10, *, ctx=None):
    """[RI, PMISet] = riSelect(carrier, csirs, reportConfig, H, nVar).  H: a DeviceArray [nRE x nRx x P] gathered at the CSI-RS REs or numpy [K x L x nRx x P]."""
    from ... import _lib as L
    if not isinstance(H, L.DeviceArray):
        ctx = ctx or L.default_context()
        h = np.asarray(H, dtype=np.complex128)
        k = np.asarray(csirs.k, dtype=np.int64).reshape(-1) - 1
        l = np.asarray(csirs.l, dtype=np.int64).reshape(-1) - 1
        if k.size == 0:
            return float("nan"), None
        H = ctx.to_device(np.asfortranarray(h[k, l, :, :]))
    ri, _, pmi, _ = riSelectBatch(carrier, csirs, reportConfig, [H], [nVar], np.zeros(1), ctx=ctx)[0]
    return ri, pmi
