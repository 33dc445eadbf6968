"""Uplink channel quality from SRS: ``communication.phyLayer.pmiSelect`` (+communication/+phyLayer/pmiSelect.m:28-65) with ``sinrPerSubband``
(sinrPerSubband.m:12-36) and ``maxPUSCHPrecodingMatrixIndicator`` (maxPUSCHPrecodingMatrixIndicator.m:29-70), and the report the gNB makes of the result
(gNBPhy.m:1033-1058: subbands without SRS, per-RB CQI).  The per-RE x per-TPMI LMMSE SINR and the subband sums run on the GPU (isac_srs_pmi_select_batch_dev);
the PUSCH codebook (TS 38.211 Tables 6.3.1.5-1 / -4: one or two SRS ports) is scalar host prep (isac_pusch_codebook)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ... import _lib as L
from .precodedSINR import UPLINK_SINR90PC


def maxPUSCHPrecodingMatrixIndicator(nlayers: int, nports: int) -> int:
    if nports not in (1, 2, 4):
        raise ValueError(f"Invalid number of ports ({nports}). The number of ports must be 1, 2 or 4.")
    if nlayers > nports:
        raise ValueError(f"The number of layers ({nlayers}) must be lower than or equal to the number of ports ({nports}).")
    if nlayers == 1:
        return {1: 0, 2: 5, 4: 27}[nports]
    if nlayers == 2:
        return 2 if nports == 2 else 21
    return 6 if nlayers == 3 else 4


def puschCodebook(nlayers: int, nports: int):
    """nrPUSCHCodebook(nlayers, nports, tpmi).' for every TPMI: W [nports x nlayers x (maxTPMI + 1)]."""
    lib = L.load()
    n = C.c_int32(0)
    st = lib.isac_pusch_codebook(C.c_int32(int(nlayers)), C.c_int32(int(nports)), None, C.c_int64(0), C.byref(n))
    if st != 0:
        raise L.IsacError(st, "isac_pusch_codebook: one or two antenna ports, layers <= ports")
    w = np.zeros((int(nports), int(nlayers), n.value), dtype=np.complex128, order="F")
    st = lib.isac_pusch_codebook(C.c_int32(int(nlayers)), C.c_int32(int(nports)), w.ctypes.data_as(C.c_void_p), C.c_int64(w.size), C.byref(n))
    if st != 0:
        raise L.IsacError(st, "isac_pusch_codebook failed")
    return w


def srsReportBatch(nlayers, H_list, re_k, nVar_list, bandSize, NRBsUL, SINRTable=UPLINK_SINR90PC, *, ctx=None):
    """The SRS measurement of gNBPhy.m:1023-1058 for many UEs that share the SRS positions (one call: one launch per stage, one copy back).
    H_list: DeviceArrays [nRE x R x P], the channel estimates at the SRS resource elements; re_k: 0-based subcarrier of each RE.
    Returns a list of (pmi [nSB] 0-based, sinrSubbandPMI [nSB], cqiRBs [NRBsUL]) per UE."""
    H_list = list(H_list)
    if not H_list:
        return []
    ctx = ctx or H_list[0].ctx
    n_re, r, p = H_list[0].shape
    k = np.ascontiguousarray(np.asarray(re_k, dtype=np.int32).reshape(-1))
    if k.size != n_re or any(tuple(h.shape) != (n_re, r, p) for h in H_list):
        raise ValueError("srsReportBatch: every H must be [nRE x R x P] at the same SRS resource elements")
    n_ue = len(H_list)
    ptrs = (C.c_void_p * n_ue)(*[h.ptr for h in H_list])
    nvar = np.ascontiguousarray(np.asarray(nVar_list, dtype=np.float64).reshape(-1))
    if nvar.size != n_ue:
        raise ValueError("srsReportBatch: one noise variance per UE")
    table = np.ascontiguousarray(np.asarray(SINRTable, dtype=np.float64))
    reps = (L.SrsReport * n_ue)()
    ctx.check(ctx.lib.isac_srs_pmi_select_batch_dev(ctx.handle, C.c_int32(n_ue), ptrs, C.c_int64(n_re), C.c_int32(r), C.c_int32(p), k.ctypes.data_as(C.c_void_p),
                                                    C.c_int32(int(NRBsUL)), C.c_int32(int(bandSize)), C.c_int32(int(nlayers)), nvar.ctypes.data_as(C.c_void_p),
                                                    table.ctypes.data_as(C.c_void_p), C.c_int32(table.size), reps))
    return [(np.array(rep.pmi[: rep.n_subbands]), np.array(rep.sinr_subband_pmi[: rep.n_subbands]), np.array(rep.cqi_rb[: rep.n_rb])) for rep in reps]


def pmiSelect(nlayers, hest, noiseest, bandSize, *, ctx=None):
    """[pmi, sinr] = pmiSelect(nlayers, hest, noiseest, bandSize): hest numpy [K x L x R x P].  Returns (pmi [nSB] 0-based / NaN, the SINR of each subband's PMI);
    (NaN, NaN) when there is no estimate or noiseest == 0 (pmiSelect.m:60-64).  The NaN replacement of gNBPhy.m:1035-1040 is NOT applied here: subbands
    without an estimate stay NaN as in the reference's function (srsReportBatch is the gNB's whole step)."""
    h = np.asarray(hest, dtype=np.complex128)
    n_sc, n_sym, r, p = h.shape
    have = np.sum(h, axis=(2, 3)) != 0                                    # pmiSelect.m:36
    if not np.any(have) or noiseest == 0:
        return np.nan, np.nan
    ll, kk = np.nonzero(have.T)                                           # column-major walk: subcarriers fastest
    ctx = ctx or L.default_context()
    d_h = ctx.to_device(np.asfortranarray(h[kk, ll, :, :]))
    n_rb = -(-n_sc // 12)
    # subbands without an estimate: run the gNB step and undo nothing -- their pmi comes back filled; mask them again from the RE list
    pmi, sel, _ = srsReportBatch(nlayers, [d_h], kk, [noiseest], bandSize, n_rb, ctx=ctx)[0]
    n_sb = pmi.size
    present = np.zeros(n_sb, dtype=bool)
    present[np.minimum(kk // (12 * int(bandSize)), n_sb - 1)] = True
    pmi = np.where(present, pmi, np.nan)
    sel = np.where(present, sel, np.nan)
    return pmi, sel
