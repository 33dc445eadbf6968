"""communication.phyLayer.precodedSINR (+communication/+phyLayer/precodedSINR.m:11-17), batched over resource elements,
and getCQI (+communication/+phyLayer/cqiSelect.m:697-722) with the tables of setupSINRtoCQIMappingTable.m:7-11."""
from __future__ import annotations

import ctypes as C

import os

import numpy as np

from ... import _lib as L

_DEBUG_UPLOAD = os.environ.get("ISAC_DEBUG_UPLOAD") is not None
DOWNLINK_SINR90PC = np.array([-3.46, 1.54, 6.54, 11.05, 13.54, 16.04, 17.54, 20.04, 22.04, 24.43, 26.93, 27.43, 29.43, 32.43, 35.43])
UPLINK_SINR90PC = np.array([-5.46, -0.46, 4.54, 9.05, 11.54, 14.04, 15.54, 18.04, 20.04, 22.43, 24.93, 25.43, 27.43, 30.43, 33.43])


def _run(H, sigma, W, table, want_per_re, ctx):
    ctx = ctx or (H.ctx if isinstance(H, L.DeviceArray) else L.default_context())
    d_h = H if isinstance(H, L.DeviceArray) else ctx.to_device(L.as_c128_f(H))
    if len(d_h.shape) == 2:                                    # a single RE: [Nr x P]
        n_re, nr, p = 1, d_h.shape[0], d_h.shape[1]
    else:
        n_re, nr, p = d_h.shape
    w = L.as_c128_f(W)
    if w.shape[0] != p:
        raise ValueError("W must be [P x nLayers]")
    per = ctx.empty((n_re,), np.float64) if want_per_re else None
    mean, cqi = C.c_double(0), C.c_int32(0)
    tab = None if table is None else np.ascontiguousarray(table, dtype=np.float64)
    ctx.check(ctx.lib.isac_precoded_sinr_cqi_dev(ctx.handle, C.c_void_p(d_h.ptr), C.c_int64(n_re), C.c_int32(nr), C.c_int32(p),
                                                 w.ctypes.data_as(C.c_void_p), C.c_int32(w.shape[1]), C.c_double(float(sigma)),
                                                 tab.ctypes.data_as(C.c_void_p) if tab is not None else None,
                                                 C.c_int32(0 if tab is None else tab.size), C.c_void_p(per.ptr if per is not None else 0),
                                                 C.byref(mean), C.byref(cqi)))
    if _DEBUG_UPLOAD and not isinstance(H, L.DeviceArray):      # development switch (ISAC_DEBUG_UPLOAD=1): read the channel estimate back and compare it with what was uploaded
        import sys
        back, src = d_h.numpy().reshape(-1, order="F"), L.as_c128_f(H).reshape(-1, order="F")
        bad = np.flatnonzero(back != src)
        if bad.size:
            z = int(np.count_nonzero(back[bad] == 0))
            sys.stderr.write(f"ISAC_DEBUG_UPLOAD: {bad.size} of {src.size} elements of the uploaded H differ on the device ({z} of them are zero there); first {bad[0]} last {bad[-1]} "
                             f"(byte offsets {16 * int(bad[0])} .. {16 * int(bad[-1]) + 15}); device pointer {d_h.ptr:#x}\n")
    return (per.numpy() if per is not None else None), mean.value, cqi.value


def precodedSINR(H, sigma, W, *, ctx=None):
    """sinr = precodedSINR(H, sigma, W).  H [Nr x P] -> scalar, or a batch H [nRE x Nr x P] -> [nRE] (one GPU thread per RE)."""
    per, _, _ = _run(H, sigma, W, None, True, ctx)
    single = (len(np.shape(H)) == 2) if not isinstance(H, L.DeviceArray) else (len(H.shape) == 2)
    return float(per[0]) if single else per


def getCQI(linearSINR, SINRTable):
    """CQI = getCQI(linearSINR, SINRTable) (cqiSelect.m:697-722); host scalar lookup."""
    if np.all(np.isnan(linearSINR)):
        return float("nan")
    with np.errstate(divide="ignore", invalid="ignore"):
        s_db = 10.0 * np.log10(linearSINR)
    idx = np.flatnonzero(np.asarray(SINRTable) <= s_db)
    return 0 if idx.size == 0 else int(idx[-1]) + 1


def cqiFromChannel(H, sigma, W, SINRTable=DOWNLINK_SINR90PC, *, ctx=None):
    """Wideband CQI of a channel estimate: mean over REs of precodedSINR, then getCQI -- the per-UE
    'SINR -> CQI' output of config 5.  Returns (cqi, mean linear SINR)."""
    _, mean, cqi = _run(H, sigma, W, SINRTable, False, ctx)
    return (float("nan") if cqi < 0 else int(cqi)), mean
