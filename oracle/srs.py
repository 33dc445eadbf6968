"""Oracle restatement of the uplink channel-quality measurement from SRS (SURVEY.md section 2 row 12's touch points).
TEST INFRASTRUCTURE ONLY -- nothing in the product imports this.

Follows, loops kept as loops:
  * maxPUSCHPrecodingMatrixIndicator   +communication/+phyLayer/maxPUSCHPrecodingMatrixIndicator.m:29-70
  * precodedSINR                       +communication/+phyLayer/precodedSINR.m:10-16
  * sinrPerSubband                     +communication/+phyLayer/sinrPerSubband.m:12-36
  * pmiSelect                          +communication/+phyLayer/pmiSelect.m:28-65
  * the report the gNB makes of it     +communication/+phyLayer/gNBPhy.m:1033-1058 (NaN subbands, per-RB CQI)
nrPUSCHCodebook is 5G Toolbox code (absent from /root/reference): what is restated is the published content of TS 38.211 Tables
6.3.1.5-1 (two ports, one layer) and 6.3.1.5-4 (two ports, two layers) and the single-port case; four ports are not restated
(the reference's UE has two transmit antennas).  **parity unpinned** like the rest of oracle/ (no MATLAB in the image)."""
from __future__ import annotations

import numpy as np


def max_pusch_tpmi(n_layers: int, n_ports: int) -> int:
    """maxPUSCHPrecodingMatrixIndicator.m:29-70."""
    if n_ports not in (1, 2, 4):
        raise ValueError("Invalid number of ports")
    if n_layers > n_ports:
        raise ValueError("The number of layers must be lower than or equal to the number of ports")
    if n_layers == 1:
        return {1: 0, 2: 5, 4: 27}[n_ports]
    if n_layers == 2:
        return 2 if n_ports == 2 else 21
    return 6 if n_layers == 3 else 4


def pusch_codebook(n_layers: int, n_ports: int, tpmi: int):
    """nrPUSCHCodebook(nlayers, nports, tpmi).' : [nports x nlayers] (TS 38.211 6.3.1.5, tables -1 and -4)."""
    if n_ports == 1:
        return np.ones((1, 1), dtype=np.complex128)
    if n_ports != 2:
        raise NotImplementedError("four-port PUSCH codebooks are not restated")
    if n_layers == 1:
        cols = ([1, 0], [0, 1], [1, 1], [1, -1], [1, 1j], [1, -1j])
        return (np.array(cols[tpmi], dtype=np.complex128) / np.sqrt(2)).reshape(2, 1)
    mats = (np.array([[1, 0], [0, 1]]) / np.sqrt(2), np.array([[1, 1], [1, -1]]) / 2, np.array([[1, 1], [1j, -1j]]) / 2)
    return mats[tpmi].astype(np.complex128)


def precoded_sinr(h, sigma, w):
    """precodedSINR.m:10-16: noise = sigma^2 I; den = noise / ((W' H') H W + noise); real(sum(1 ./ diag(den) - 1))."""
    noise = sigma ** 2 * np.eye(w.shape[1])
    den = noise @ np.linalg.inv((w.conj().T @ h.conj().T) @ h @ w + noise)
    return float(np.real(np.sum(1.0 / np.diag(den) - 1.0)))


def sinr_per_subband(sinr, band_size: int):
    """sinrPerSubband.m:12-36.  sinr [K x L x nTPMI]; returns (sinrSubband [nSB x nTPMI], subbandIndices [nSB x 2], 1-based)."""
    nrb = sinr.shape[0] / 12
    r = nrb / band_size
    fl = int(np.floor(r))
    starts = [12 * band_size * v + 1 for v in range(fl)]                 # 0 : r - 1
    ends = [12 * band_size * v for v in range(1, fl + 1)]                # 1 : r
    if fl != r:                                                           # extraBand
        starts.append(12 * band_size * fl + 1)
        ends.append(int(round(12 * band_size * r)))
    n_tpmi = sinr.shape[2]
    n_sb = int(np.ceil(nrb / band_size))
    out = np.zeros((n_sb, n_tpmi))
    for s in range(n_sb):
        blk = sinr[starts[s] - 1:ends[s], :, :]
        cnt = np.sum(np.sum(blk, axis=2) != 0)
        with np.errstate(invalid="ignore", divide="ignore"):
            tot = np.zeros(n_tpmi)
            for e in range(n_tpmi):                                       # sum(., [1 2]): column-major order, subcarriers fastest
                acc = 0.0
                for l in range(blk.shape[1]):
                    for k in range(blk.shape[0]):
                        acc += blk[k, l, e]
                tot[e] = acc
            out[s, :] = tot / cnt if cnt else np.full(n_tpmi, np.nan)
    return out, np.stack([np.array(starts), np.array(ends)], axis=1)


def pmi_select(n_layers: int, hest, noiseest: float, band_size: int):
    """pmiSelect.m:28-65.  hest [K x L x R x P].  Returns (pmi [nSB] 0-based / NaN, sinr [nSB x nTPMI], subbandIndices) or (NaN, NaN, NaN)."""
    n_ports = hest.shape[3]
    max_tpmi = max_pusch_tpmi(n_layers, n_ports)
    n_sc, n_sym = hest.shape[:2]
    sinr = np.zeros((n_sc, n_sym, max_tpmi + 1))
    have = np.sum(hest, axis=(2, 3)) != 0
    if not np.any(have) or noiseest == 0:
        return np.nan, np.nan, np.nan
    sigma = np.sqrt(noiseest)
    for tpmi in range(max_tpmi + 1):
        w = pusch_codebook(n_layers, n_ports, tpmi)
        for l in range(n_sym):
            for k in range(n_sc):
                if have[k, l]:
                    sinr[k, l, tpmi] = precoded_sinr(hest[k, l], sigma, w)
    bands, idx = sinr_per_subband(sinr, band_size)
    pmi = np.argmax(bands, axis=1).astype(np.float64) + 1                 # [~, pmi] = max(sinrBands, [], 2): the first maximiser
    pmi[np.isnan(bands[:, 0])] = np.nan
    return pmi - 1, bands, idx


def srs_report(n_layers: int, hest, noiseest: float, band_size: int, n_rbs_ul: int, sinr_table_db):
    """gNBPhy.m:1033-1058 on pmiSelect's output.  Returns (pmi [nSB], sinrSubbandPMI [nSB], cqiRBs [NRBsUL])."""
    pmi, sinr_sb, _ = pmi_select(n_layers, hest, noiseest, band_size)
    pmi = np.array(pmi, dtype=np.float64).reshape(-1)
    sinr_sb = np.array(sinr_sb, dtype=np.float64)
    nan_idx = np.isnan(pmi)
    if np.any(nan_idx) and not np.all(nan_idx):
        pmi[nan_idx] = np.floor(np.mean(pmi[~nan_idx]))
        sinr_sb[nan_idx, :] = np.mean(sinr_sb[~nan_idx, :], axis=0)
    sel = np.array([sinr_sb[i, int(pmi[i])] for i in range(pmi.size)])
    table = np.asarray(sinr_table_db, dtype=np.float64)
    cqi_rbs = np.zeros(n_rbs_ul)
    n_sb = sel.size
    for i in range(1, n_sb):                                              # i = 1 : numSubbands - 1
        with np.errstate(divide="ignore", invalid="ignore"):
            filt = table[table <= 10 * np.log10(sel[i - 1])]
        nz = np.flatnonzero(filt)                                         # find(values, 1, 'last'): the last NONZERO value of the filtered table
        if nz.size:
            cqi_rbs[(i - 1) * band_size:i * band_size] = (nz[-1] + 1) - 1
    if n_sb >= 2:
        cqi_rbs[(n_sb - 1) * band_size:] = cqi_rbs[(n_sb - 1) * band_size - 1]
    cqi_rbs[cqi_rbs <= 1] = 1
    return pmi, sel, cqi_rbs
