"""Oracle restatement of the Type-I single-panel PMI search and the subband SINR -> CQI report of config 5
(SURVEY.md 8a row a11 / 8f rank 3).  TEST INFRASTRUCTURE ONLY.

Follows, line by line (loops kept as loops):
  * getPMIType1SinglePanelCodebook   +communication/+phyLayer/dlPMISelect.m:853-1083   (1 and 2 layers, codebook modes 1 / 2,
                                     2 ports and > 2 ports; no subset / i2 restriction -- the reference sets none, setupCSIRS.m:17-23)
  * getVlm                           dlPMISelect.m:1774-1782
  * getPrecodedSINR                  dlPMISelect.m:1825-1834
  * the search itself                dlPMISelect.m:385-500  (SINR of every CSI-RS RE for every codebook entry, total rounded to four
                                     decimals :449, first maximiser in MATLAB's column-major index order :453, per-subband i2 :465-498)
  * getSubbandInfo                   cqiSelect.m:1209-1245 / dlPMISelect.m:1836-1883
  * the CQI report                   cqiSelect.m:500-687 (CSI-RS-object syntax, no PRGSize), getCQI :697-722
CSI-RS resource positions are an explicit input (1-based (k, l) subscripts of the first port's REs): nrCSIRSIndices is toolbox code.
Only ranks 1-2 are restated: the reference's UEs have two receive antennas (riSelect caps the rank at min(Nr, P)).
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from .cqi import get_cqi

# TS 38.214 Table 5.2.2.2.1-2 as dlPMISelect.m:622-625 holds it: N1; N2; O1; O2
_PANEL_CONFIGS = np.array([[2, 2, 4, 3, 6, 4, 8, 4, 6, 12, 4, 8, 16],
                           [1, 2, 1, 2, 1, 2, 1, 3, 2, 1, 4, 2, 1],
                           [4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4],
                           [1, 4, 1, 4, 1, 4, 1, 4, 4, 1, 4, 4, 1]])


def oversampling_factors(n1: int, n2: int):
    """dlPMISelect.m:622-637."""
    idx = np.flatnonzero((_PANEL_CONFIGS[0] == n1) & (_PANEL_CONFIGS[1] == n2))
    if idx.size == 0:
        raise ValueError("panel configuration not in TS 38.214 Table 5.2.2.2.1-2")
    return int(_PANEL_CONFIGS[2, idx[0]]), int(_PANEL_CONFIGS[3, idx[0]])


def get_vlm(n1, n2, o1, o2, l, m):
    """dlPMISelect.m:1774-1782."""
    um = np.exp(2j * np.pi * m * np.arange(n2) / (o2 * n2))
    ul = np.exp(2j * np.pi * l * np.arange(n1) / (o1 * n1))
    return (ul[:, None] * um[None, :]).reshape(-1)          # reshape((ul.*um).',[],1): n2 fastest


def type1_single_panel_codebook(panel_dimensions, codebook_mode: int, n_layers: int, n_ports: int):
    """W [P x nLayers x i2 x i11 x i12 x i13] (dlPMISelect.m:853-1083; trailing singleton dimensions kept explicit)."""
    phi = lambda x: np.exp(1j * np.pi * x / 2)                                           # :888
    if n_ports == 2:                                                                     # :892-916
        if n_layers == 1:
            w = np.zeros((2, 1, 4, 1, 1, 1), dtype=np.complex128)
            for i, v in enumerate(([1, 1], [1, 1j], [1, -1], [1, -1j])):
                w[:, 0, i, 0, 0, 0] = np.array(v) / np.sqrt(2)
        elif n_layers == 2:
            w = np.zeros((2, 2, 2, 1, 1, 1), dtype=np.complex128)
            w[:, :, 0, 0, 0, 0] = 0.5 * np.array([[1, 1], [1, -1]])
            w[:, :, 1, 0, 0, 0] = 0.5 * np.array([[1, 1], [1j, -1j]])
        else:
            raise NotImplementedError
        return w
    n1, n2 = panel_dimensions
    o1, o2 = oversampling_factors(n1, n2)
    p = 2 * n1 * n2
    if n_layers == 1:                                                                    # :926-985
        if codebook_mode == 1:
            i11l, i12l, i2l = n1 * o1, n2 * o2, 4
            w = np.zeros((p, 1, i2l, i11l, i12l, 1), dtype=np.complex128)
            for i11 in range(i11l):
                for i12 in range(i12l):
                    for i2 in range(i2l):
                        vlm = get_vlm(n1, n2, o1, o2, i11, i12)
                        w[:, 0, i2, i11, i12, 0] = (1 / np.sqrt(p)) * np.concatenate([vlm, phi(i2) * vlm])
        else:
            i11l = n1 * o1 // 2
            i12l = 1 if n2 == 1 else n2 * o2 // 2
            i2l = 16
            w = np.zeros((p, 1, i2l, i11l, i12l, 1), dtype=np.complex128)
            add = [(0, 0), (1, 0), (0, 1), (1, 1)]
            for i11 in range(i11l):
                for i12 in range(i12l):
                    for i2 in range(i2l):
                        f = i2 // 4
                        if n2 == 1:
                            l, m = 2 * i11 + f, 0
                        else:
                            l, m = 2 * i11 + add[f][0], 2 * i12 + add[f][1]
                        vlm = get_vlm(n1, n2, o1, o2, l, m)
                        w[:, 0, i2, i11, i12, 0] = (1 / np.sqrt(p)) * np.concatenate([vlm, phi(i2 % 4) * vlm])
        return w
    if n_layers == 2:                                                                    # :987-1083
        if n1 > n2 and n2 > 1:
            i13l, k1, k2 = 4, [0, o1, 0, 2 * o1], [0, 0, o2, 0]
        elif n1 == n2:
            i13l, k1, k2 = 4, [0, o1, 0, o1], [0, 0, o2, o2]
        elif n1 == 2 and n2 == 1:
            i13l, k1, k2 = 2, [0, o1], [0, 0]
        else:
            i13l, k1, k2 = 4, [0, o1, 2 * o1, 3 * o1], [0, 0, 0, 0]
        if codebook_mode == 1:
            i11l, i12l, i2l = n1 * o1, n2 * o2, 2
            w = np.zeros((p, 2, i2l, i11l, i12l, i13l), dtype=np.complex128)
            for i11 in range(i11l):
                for i12 in range(i12l):
                    for i13 in range(i13l):
                        for i2 in range(i2l):
                            vlm = get_vlm(n1, n2, o1, o2, i11, i12)
                            vlp = get_vlm(n1, n2, o1, o2, i11 + k1[i13], i12 + k2[i13])
                            ph = phi(i2)
                            w[:, :, i2, i11, i12, i13] = (1 / np.sqrt(2 * p)) * np.block([[vlm[:, None], vlp[:, None]],
                                                                                          [ph * vlm[:, None], -ph * vlp[:, None]]])
        else:
            i11l = n1 * o1 // 2
            i12l = 1 if n2 == 1 else n2 * o2 // 2
            i2l = 8
            w = np.zeros((p, 2, i2l, i11l, i12l, i13l), dtype=np.complex128)
            add = [(0, 0), (1, 0), (0, 1), (1, 1)]
            for i11 in range(i11l):
                for i12 in range(i12l):
                    for i13 in range(i13l):
                        for i2 in range(i2l):
                            f = i2 // 2
                            if n2 == 1:
                                l, lp, m, mp = 2 * i11 + f, 2 * i11 + f + k1[i13], 0, 0
                            else:
                                l, lp = 2 * i11 + add[f][0], 2 * i11 + k1[i13] + add[f][0]
                                m, mp = 2 * i12 + add[f][1], 2 * i12 + k2[i13] + add[f][1]
                            vlm = get_vlm(n1, n2, o1, o2, l, m)
                            vlp = get_vlm(n1, n2, o1, o2, lp, mp)
                            ph = phi(i2 % 2)
                            w[:, :, i2, i11, i12, i13] = (1 / np.sqrt(2 * p)) * np.block([[vlm[:, None], vlp[:, None]],
                                                                                          [ph * vlm[:, None], -ph * vlp[:, None]]])
        return w
    raise NotImplementedError("ranks 3-8 are not restated (two-antenna UEs)")


def get_precoded_sinr(h, n_var, w):
    """dlPMISelect.m:1825-1834: noise = nVar I; den = noise / ((W' H') H W + noise); sinr = real(1 ./ diag(den) - 1)."""
    n_l = w.shape[1]
    noise = n_var * np.eye(n_l)
    den = noise @ np.linalg.inv((w.conj().T @ h.conj().T) @ h @ w + noise)
    return np.real(1.0 / np.diag(den) - 1.0)


def subband_info(mode: str, n_start_bwp: int, n_size_bwp: int, nsbprb: int):
    """getSubbandInfo (cqiSelect.m:1209-1245)."""
    if mode.lower() == "wideband" or n_size_bwp < 24:
        return SimpleNamespace(NumSubbands=1, SubbandSizes=[n_size_bwp])
    first = nsbprb - n_start_bwp % nsbprb
    last = (n_start_bwp + n_size_bwp) % nsbprb or nsbprb
    n = (n_size_bwp - (first + last)) // nsbprb + 2
    sizes = [nsbprb] * n
    sizes[0], sizes[-1] = first, last
    return SimpleNamespace(NumSubbands=n, SubbandSizes=sizes)


def matlab_round4(x):
    """round(x, 4, 'decimal'): half away from zero on x * 1e4."""
    x = np.asarray(x, dtype=np.float64)
    return np.sign(x) * np.floor(np.abs(x) * 1e4 + 0.5) / 1e4


def _nanmean_matlab(a, axis):
    with np.errstate(invalid="ignore", divide="ignore"):
        cnt = np.sum(~np.isnan(a), axis=axis)
        s = np.nansum(a, axis=axis)
        return np.where(cnt > 0, s / np.maximum(cnt, 1), np.nan)


def dl_pmi_select(report, n_layers: int, h, csirs_k, csirs_l, n_var: float, symbols_per_slot: int = 14):
    """dlPMISelect.m:240-500, Type1SinglePanel.  ``h`` [K x L x nRx x P]; ``csirs_k/l`` 1-based subscripts of the first port's CSI-RS REs
    (relative to the BWP).  ``report``: NSizeBWP, NStartBWP, PanelDimensions, CodebookMode, PMIMode, SubbandSize.
    Returns (PMISet{i1 [3], i2 [numSubbands]} 1-based / NaN, info{SINRPerRE, SINRPerSubband, W})."""
    n_ports = h.shape[3]
    sb = subband_info(report.PMIMode, report.NStartBWP, report.NSizeBWP, report.SubbandSize)
    w = type1_single_panel_codebook(report.PanelDimensions, report.CodebookMode, n_layers, n_ports)
    idx_sizes = w.shape[2:]
    k_all, l_all = np.asarray(csirs_k), np.asarray(csirs_l)
    sinr_re = np.full((report.NSizeBWP * 12, symbols_per_slot, n_layers) + idx_sizes, np.nan)
    for k, l in zip(k_all, l_all):                                                       # :385-427
        ht = h[k - 1, l - 1]                                                             # [nRx x P]
        for i11 in range(idx_sizes[1]):
            for i12 in range(idx_sizes[2]):
                for i13 in range(idx_sizes[3]):
                    for i2 in range(idx_sizes[0]):
                        cw = w[:, :, i2, i11, i12, i13]
                        if np.any(cw):
                            sinr_re[k - 1, l - 1, :, i2, i11, i12, i13] = get_precoded_sinr(ht, n_var, cw)
    pmi = SimpleNamespace(i1=np.full(3, np.nan), i2=np.full(sb.NumSubbands, np.nan))
    sb_sinr = np.full((sb.NumSubbands, n_layers) + idx_sizes, np.nan)
    if k_all.size == 0 or np.all(np.isnan(sinr_re)):                                     # :364-376, :436-444
        return pmi, SimpleNamespace(SINRPerRE=sinr_re, SINRPerSubband=sb_sinr, W=w, SubbandInfo=sb)
    total = np.nansum(sinr_re, axis=(0, 1, 2))                                           # :446
    total = matlab_round4(total)                                                         # :449
    flat = total.reshape(-1, order="F")                                                  # find(..., 1) walks column-major
    first = int(np.flatnonzero(flat == flat.max())[0])
    i2w, i11, i12, i13 = np.unravel_index(first, idx_sizes, order="F")                   # :453
    pmi.i1 = np.array([i11 + 1, i12 + 1, i13 + 1], dtype=np.float64)
    start = 0
    for s in range(sb.NumSubbands):                                                      # :465-498
        rows = slice(start * 12, (start + sb.SubbandSizes[s]) * 12)
        vals = sinr_re[rows]
        if np.all(np.isnan(vals)):
            pmi.i2[s] = np.nan
        else:
            m = _nanmean_matlab(_nanmean_matlab(vals, 0), 0)                              # mean(mean(., 'omitnan'), 'omitnan'): dims 1 then 2
            sb_sinr[s] = m
            tmp = matlab_round4(np.nansum(m[:, :, i11, i12, i13], axis=0))               # sum over layers :490
            pmi.i2[s] = int(np.argmax(tmp)) + 1                                          # [~, i2] = max(.)
        start += sb.SubbandSizes[s]
    return pmi, SimpleNamespace(SINRPerRE=sinr_re, SINRPerSubband=sb_sinr, W=w, SubbandInfo=sb)


def cqi_select(report, n_layers: int, h, csirs_k, csirs_l, n_var: float, sinr_table):
    """cqiSelect.m:500-687 for the CSI-RS-object syntax without PRGSize.  Returns (CQI [(numSubbands+1) or 1], PMISet, CQIInfo, PMIInfo)."""
    cqi_sb = subband_info(report.CQIMode, report.NStartBWP, report.NSizeBWP, report.SubbandSize)
    pmi, info = dl_pmi_select(report, n_layers, h, csirs_k, csirs_l, n_var)
    all_nan = np.all(np.isnan(pmi.i1)) and np.all(np.isnan(pmi.i2))
    # NaN(CQISubbandInfo.NumSubbands, nLayers) (:519); the PMI-'Subband' branch indexes rows up to size(PMISet.i2, 2) and MATLAB grows the array
    n_rows = cqi_sb.NumSubbands if report.PMIMode.lower() == "wideband" else max(cqi_sb.NumSubbands, pmi.i2.size)
    sinr_sb = np.full((n_rows, n_layers), np.nan)
    if not all_nan:
        i11, i12, i13 = (int(v) - 1 for v in pmi.i1)
        if report.PMIMode.lower() == "wideband":                                         # :578-588 getSubbandSINR with i2 replicated
            i2 = int(pmi.i2[0]) - 1
            start = 0
            for s in range(cqi_sb.NumSubbands):
                rows = slice(start * 12, (start + cqi_sb.SubbandSizes[s]) * 12)
                sinr_sb[s] = _nanmean_matlab(_nanmean_matlab(info.SINRPerRE[rows, :, :, i2, i11, i12, i13], 0), 0)
                start += cqi_sb.SubbandSizes[s]
        else:                                                                            # :589-606
            for s in range(pmi.i2.size):
                if not np.isnan(pmi.i2[s]):
                    sinr_sb[s] = info.SINRPerSubband[s, :, int(pmi.i2[s]) - 1, i11, i12, i13]
    sinr_cw = np.array([[np.sum(r) if not np.any(np.isnan(r)) else np.nan] for r in sinr_sb])   # one codeword up to 4 layers :609-624
    if sinr_cw.shape[0] > 1:                                                             # :628-630
        sinr_cw = np.concatenate([[[_nanmean_matlab(sinr_cw[:, 0], 0)]], sinr_cw])
    if all_nan:                                                                          # :633-647
        n = 0 if cqi_sb.NumSubbands == 1 else cqi_sb.NumSubbands
        return np.full(n + 1, np.nan), pmi, SimpleNamespace(SINRPerSubbandPerCW=np.full(n + 1, np.nan), SubbandCQI=np.full(n + 1, np.nan)), info
    cqi_all = np.array([get_cqi(v, sinr_table) for v in sinr_cw[:, 0]], dtype=np.float64)  # :650
    if report.CQIMode.lower() == "subband":                                              # :654-676
        diff = cqi_all[1:] - cqi_all[0]
        off = np.full(diff.shape, np.nan)
        off[diff == 0] = 0
        off[diff == 1] = 1
        off[diff >= 2] = 2
        off[diff <= -1] = 3
        cqi = np.concatenate([[cqi_all[0]], off])
        sub_cqi = cqi_all
    else:
        cqi = cqi_all[:1]
        sub_cqi = cqi_all[:1]
        sinr_cw = sinr_cw[:1]
    return cqi, pmi, SimpleNamespace(SINRPerSubbandPerCW=sinr_cw[:, 0], SubbandCQI=sub_cqi), info


def ri_select(report, h, csirs_k, csirs_l, n_var: float, ri_restriction=None):
    """riSelect.m:207-280, Type1SinglePanel.  Returns (RI or NaN, PMISet of that rank, totalSINR [maxRank])."""
    n_rx, n_ports = h.shape[2], h.shape[3]
    max_rank = min(n_rx, n_ports)                                                        # :219-220
    restr = np.ones(8) if ri_restriction is None else np.asarray(ri_restriction)
    valid = [r for r in range(1, max_rank + 1) if r <= restr.size and restr[r - 1]]      # :226-231
    sb = subband_info(report.PMIMode, report.NStartBWP, report.NSizeBWP, report.SubbandSize)
    if not valid or np.asarray(csirs_k).size == 0:                                       # :232-242
        return np.nan, SimpleNamespace(i1=np.full(3, np.nan), i2=np.full(sb.NumSubbands, np.nan)), np.full(max_rank, np.nan)
    best, ri, pmi_set = -np.inf, np.nan, None
    total = np.full(max_rank, np.nan)                                                    # :246-247
    pmi = None
    for rank in valid:
        pmi, info = dl_pmi_select(report, rank, h, csirs_k, csirs_l, n_var)              # :254
        sub = np.full((sb.NumSubbands, rank), np.nan)
        if not np.any(np.isnan(pmi.i1)):
            i11, i12, i13 = (int(v) - 1 for v in pmi.i1)
            for s in range(sb.NumSubbands):                                              # :261-268
                if not np.isnan(pmi.i2[s]):
                    sub[s, :] = info.SINRPerSubband[s, :, int(pmi.i2[s]) - 1, i11, i12, i13] * rank
            layer = _nanmean_matlab(sub, 0)                                              # :272 mean(., 1, 'omitnan')
            total[rank - 1] = np.sum(layer[layer >= 1])                                  # :276
        if total[rank - 1] > best + 0.1:                                                 # :278-282
            best, ri, pmi_set = total[rank - 1], rank, pmi
    if np.all(np.isnan(total)):                                                          # :287-290
        ri, pmi_set = np.nan, pmi
    return ri, pmi_set, total
