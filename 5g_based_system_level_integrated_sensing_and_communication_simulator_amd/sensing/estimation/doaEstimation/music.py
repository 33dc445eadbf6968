"""sensing.estimation.doaEstimation.music (+sensing/+estimation/+doaEstimation/music.m:1-125), ULA branch.
The UPA branch of the reference ends in the undefined ``tools.find2DPeaks`` (music.m:69) and errors."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .... import _lib as L
from ..._marshal import est_block


def music(numDets, radarEstParams, Ra, *, ctx=None):
    """[L, aziEst, eleEst] = music(numDets, radarEstParams, Ra).  ``numDets`` None/[] -> model order
    from determineNumTargets (music.m:21-22,109-125)."""
    ctx = ctx or L.default_context()
    ra = L.as_c128_f(Ra)
    A = ra.shape[0]
    if ra.shape != (A, A):
        raise ValueError("Ra must be square")
    nd = -1 if (numDets is None or (hasattr(numDets, "__len__") and len(numDets) == 0)) else int(numDets)
    ep = est_block(radarEstParams)
    cap = 4096
    azi = np.zeros(cap)
    ele = np.zeros(cap)
    l_out, n_est = C.c_int32(0), C.c_int32(0)
    ctx.check(ctx.lib.isac_music_doa(ctx.handle, C.c_int32(nd), C.byref(ep), ra.ctypes.data_as(C.c_void_p), C.c_int32(A),
                                     C.byref(l_out), azi.ctypes.data_as(C.c_void_p), ele.ctypes.data_as(C.c_void_p),
                                     C.c_int32(cap), C.byref(n_est)))
    n = n_est.value
    return int(l_out.value), azi[:n].copy(), ele[:n].copy()
