"""CSI report seam of config 5: ``cqiSelect`` / ``dlPMISelect`` (+communication/+phyLayer/cqiSelect.m:500-687, dlPMISelect.m:240-500) for the
Type-I single-panel codebook, as uePhy.m:901-908 calls them.  The exhaustive per-RE x per-codebook-entry LMMSE SINR evaluation, the
subband means and the totals run on the GPU (isac_csi_report_dev); the codebook itself is scalar host prep (isac_type1sp_codebook).

CSI-RS positions are explicit: ``csirs`` carries ``k`` / ``l`` -- the 1-based subcarrier / symbol subscripts of the first CSI-RS port's
resource elements relative to the BWP (what nrCSIRSIndices + ind2sub give dlPMISelect.m:354-362; the toolbox call itself is not ours)."""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np

from ... import _lib as L


def type1SinglePanelCodebook(reportConfig, nLayers: int, nPorts: int):
    """getPMIType1SinglePanelCodebook (dlPMISelect.m:853-1083): W [P x nLayers x i2 x i11 x i12 x i13]."""
    lib = L.load()
    n1, n2 = (int(v) for v in getattr(reportConfig, "PanelDimensions", (1, 1))) if nPorts > 2 else (1, 1)
    dims = (C.c_int32 * 4)()
    st = lib.isac_type1sp_codebook(C.c_int32(nPorts), C.c_int32(n1), C.c_int32(n2), C.c_int32(int(getattr(reportConfig, "CodebookMode", 1))),
                                   C.c_int32(int(nLayers)), None, C.c_int64(0), dims)
    if st != 0:
        raise L.IsacError(st, "isac_type1sp_codebook: unsupported rank / panel configuration (ranks 1-2 of TS 38.214 Table 5.2.2.2.1-2)")
    shape = (nPorts, int(nLayers), dims[0], dims[1], dims[2], dims[3])
    w = np.zeros(shape, dtype=np.complex128, order="F")
    st = lib.isac_type1sp_codebook(C.c_int32(nPorts), C.c_int32(n1), C.c_int32(n2), C.c_int32(int(getattr(reportConfig, "CodebookMode", 1))),
                                   C.c_int32(int(nLayers)), w.ctypes.data_as(C.c_void_p), C.c_int64(w.size), dims)
    if st != 0:
        raise L.IsacError(st, "isac_type1sp_codebook failed")
    return w


def cqiSelect(carrier, csirs, reportConfig, nLayers, H, nVar, SINRTable, *, ctx=None, return_total=False):
    """[CQI, PMISet, CQIInfo, PMIInfo] = cqiSelect(carrier, csirs, reportConfig, nLayers, H, nVar, SINRTable).

    H: numpy [K x L x nRx x P] (the reference's layout) or a DeviceArray [nRE x nRx x P] already gathered at the CSI-RS REs.
    reportConfig: NSizeBWP, NStartBWP, PanelDimensions, CodebookMode, PMIMode, CQIMode, SubbandSize.  NaN where the reference reports NaN."""
    ctx = ctx or (H.ctx if isinstance(H, L.DeviceArray) else L.default_context())
    k = np.ascontiguousarray(np.asarray(csirs.k, dtype=np.int32).reshape(-1) - 1)
    l = np.ascontiguousarray(np.asarray(csirs.l, dtype=np.int32).reshape(-1) - 1)
    if isinstance(H, L.DeviceArray):
        d_h = H
        n_re, nr, p = H.shape
    else:
        h = np.asarray(H, dtype=np.complex128)
        nr, p = h.shape[2], h.shape[3]
        d_h = ctx.to_device(np.asfortranarray(h[k, l, :, :])) if k.size else None
        n_re = k.size
    if n_re != k.size:
        raise ValueError("H and csirs disagree on the number of CSI-RS resource elements")
    w = type1SinglePanelCodebook(reportConfig, nLayers, p)
    dims = (C.c_int32 * 4)(*w.shape[2:])
    table = np.ascontiguousarray(np.asarray(SINRTable, dtype=np.float64))
    rep = L.CsiReport()
    tot = np.zeros(int(np.prod(w.shape[2:])), dtype=np.float64)
    n_size = int(getattr(reportConfig, "NSizeBWP", None) or carrier.NSizeGrid)
    n_start = int(getattr(reportConfig, "NStartBWP", 0) or 0)
    ctx.check(ctx.lib.isac_csi_report_dev(ctx.handle, C.c_void_p(d_h.ptr if d_h is not None else 0), C.c_int64(n_re), C.c_int32(nr), C.c_int32(p),
                                          k.ctypes.data_as(C.c_void_p), l.ctypes.data_as(C.c_void_p), C.c_int32(n_size), C.c_int32(n_start),
                                          C.c_int32(int(reportConfig.SubbandSize)), C.c_int32(1 if str(reportConfig.PMIMode).lower() == "subband" else 0),
                                          C.c_int32(1 if str(reportConfig.CQIMode).lower() == "subband" else 0), w.ctypes.data_as(C.c_void_p),
                                          C.c_int32(int(nLayers)), dims, C.c_double(float(nVar)), table.ctypes.data_as(C.c_void_p), C.c_int32(table.size),
                                          C.byref(rep), tot.ctypes.data_as(C.c_void_p), None))
    pmi = SimpleNamespace(i1=np.array(rep.i1[:3]), i2=np.array(rep.i2[: rep.n_subbands_pmi]))
    cqi = np.array(rep.cqi[: rep.n_cqi])
    info = SimpleNamespace(SINRPerSubbandPerCW=np.array(rep.sinr_per_subband_cw[: rep.n_cqi]), SubbandCQI=np.array(rep.subband_cqi[: rep.n_cqi]))
    pinfo = SimpleNamespace(W=w, TotalSINR=tot.reshape(w.shape[2:], order="F"), RITotalSINR=rep.ri_total_sinr)
    return cqi, pmi, info, pinfo


def cqiSelectBatch(carrier, csirs, reportConfig, nLayers, H_list, nVar_list, SINRTable, *, ctx=None, codebook=None, with_ri_total=False):
    """cqiSelect for many UEs that share the CSI-RS / report configuration (uePhy.m:901-908 runs it once per UE; isac_csi_report_batch_dev runs the
    cell's UEs in one call: one upload, one launch per stage, one synchronisation).  H_list: DeviceArrays [nRE x nRx x P] gathered at the CSI-RS REs;
    nVar_list: one noise variance per UE.  Returns a list of (CQI, PMISet, CQIInfo) per UE -- with_ri_total: (CQI, PMISet, CQIInfo, totalSINR of riSelect.m:253-276 at this rank).
    `codebook`: W from type1SinglePanelCodebook (built if None)."""
    H_list = list(H_list)
    if not H_list:
        return []
    ctx = ctx or H_list[0].ctx
    k = np.ascontiguousarray(np.asarray(csirs.k, dtype=np.int32).reshape(-1) - 1)
    l = np.ascontiguousarray(np.asarray(csirs.l, dtype=np.int32).reshape(-1) - 1)
    n_re, nr, p = H_list[0].shape
    if n_re != k.size or any(tuple(h.shape) != (n_re, nr, p) for h in H_list):
        raise ValueError("cqiSelectBatch: every H must be [nRE x nRx x P] at the same CSI-RS resource elements")
    w = codebook if codebook is not None else type1SinglePanelCodebook(reportConfig, nLayers, p)
    dims = (C.c_int32 * 4)(*w.shape[2:])
    table = np.ascontiguousarray(np.asarray(SINRTable, dtype=np.float64))
    n_ue = len(H_list)
    ptrs = (C.c_void_p * n_ue)(*[h.ptr for h in H_list])
    nvar = np.ascontiguousarray(np.asarray(nVar_list, dtype=np.float64).reshape(-1))
    if nvar.size != n_ue:
        raise ValueError("cqiSelectBatch: one noise variance per UE")
    reps = (L.CsiReport * n_ue)()
    n_size = int(getattr(reportConfig, "NSizeBWP", None) or carrier.NSizeGrid)
    n_start = int(getattr(reportConfig, "NStartBWP", 0) or 0)
    ctx.check(ctx.lib.isac_csi_report_batch_dev(ctx.handle, C.c_int32(n_ue), ptrs, C.c_int64(n_re), C.c_int32(nr), C.c_int32(p), k.ctypes.data_as(C.c_void_p),
                                                l.ctypes.data_as(C.c_void_p), C.c_int32(n_size), C.c_int32(n_start), C.c_int32(int(reportConfig.SubbandSize)),
                                                C.c_int32(1 if str(reportConfig.PMIMode).lower() == "subband" else 0),
                                                C.c_int32(1 if str(reportConfig.CQIMode).lower() == "subband" else 0), w.ctypes.data_as(C.c_void_p),
                                                C.c_int32(int(nLayers)), dims, nvar.ctypes.data_as(C.c_void_p), table.ctypes.data_as(C.c_void_p),
                                                C.c_int32(table.size), reps, None))
    out = []
    for rep in reps:
        pmi = SimpleNamespace(i1=np.array(rep.i1[:3]), i2=np.array(rep.i2[: rep.n_subbands_pmi]))
        info = SimpleNamespace(SINRPerSubbandPerCW=np.array(rep.sinr_per_subband_cw[: rep.n_cqi]), SubbandCQI=np.array(rep.subband_cqi[: rep.n_cqi]))
        out.append((np.array(rep.cqi[: rep.n_cqi]), pmi, info) + ((rep.ri_total_sinr,) if with_ri_total else ()))
    return out


def dlPMISelect(carrier, csirs, reportConfig, nLayers, H, nVar=1e-10, *, ctx=None):
    """[PMISet, info] = dlPMISelect(carrier, csirs, reportConfig, nLayers, H, nVar) (dlPMISelect.m:1)."""
    rc = SimpleNamespace(**vars(reportConfig))
    rc.CQIMode = getattr(reportConfig, "CQIMode", "Wideband")
    _, pmi, _, pinfo = cqiSelect(carrier, csirs, rc, nLayers, H, nVar, np.zeros(1), ctx=ctx)
    return pmi, pinfo
