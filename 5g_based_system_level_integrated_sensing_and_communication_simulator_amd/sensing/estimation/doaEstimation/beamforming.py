"""sensing.estimation.doaEstimation.digitalBF / mvdrBF (+sensing/+estimation/+doaEstimation/digitalBF.m:55-86,
mvdrBF.m:55-86), ULA branch.  No caller in the reference; they reuse the MUSIC eigendecomposition + scan kernel."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .... import _lib as L
from ..._marshal import est_block


def _scan(method, numDets, radarEstParams, Ra, ctx):
    ctx = ctx or L.default_context()
    ra = L.as_c128_f(Ra)
    A = ra.shape[0]
    ep = est_block(radarEstParams)
    cap = 4096
    azi, ele, n = np.zeros(cap), np.zeros(cap), C.c_int32(0)
    ctx.check(ctx.lib.isac_beamscan_doa(ctx.handle, C.c_int32(method), C.c_int32(int(numDets)), C.byref(ep), ra.ctypes.data_as(C.c_void_p),
                                        C.c_int32(A), azi.ctypes.data_as(C.c_void_p), ele.ctypes.data_as(C.c_void_p), C.c_int32(cap),
                                        C.byref(n)))
    return azi[: n.value].copy(), ele[: n.value].copy()


def digitalBF(numDets, radarEstParams, Ra, *, ctx=None):
    """[aziEst, eleEst] = digitalBF(numDets, radarEstParams, Ra)   (digitalBF.m:1)."""
    return _scan(1, numDets, radarEstParams, Ra, ctx)


def mvdrBF(numDets, radarEstParams, Ra, *, ctx=None):
    """[aziEst, eleEst] = mvdrBF(numDets, radarEstParams, Ra)   (mvdrBF.m:1)."""
    return _scan(2, numDets, radarEstParams, Ra, ctx)
