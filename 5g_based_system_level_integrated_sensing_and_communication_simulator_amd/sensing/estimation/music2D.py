"""sensing.estimation.music2D (+sensing/+estimation/music2D.m:1-123).  Dead code in the reference (no caller) but
named by the north star: MUSIC DoA + MUSIC range / velocity spectra."""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np

from ... import _lib as L
from .._marshal import est_block


def music2D(rdrEstParams, bsParams, rxGrid, txGrid, *, ctx=None):
    """estResults = music2D(rdrEstParams, bsParams, rxGrid, txGrid) -> .aziEst .eleEst .rngEst .velEst (+ .L).
    ``bsParams.scs`` in kHz (music2D.m:34)."""
    dev = isinstance(rxGrid, L.DeviceArray)
    ctx = ctx or (rxGrid.ctx if dev else L.default_context())
    K, Ls, A = (rxGrid.shape if dev else np.shape(rxGrid))
    d_rx = rxGrid if dev else ctx.to_device(L.as_c128_f(rxGrid))
    d_tx = txGrid if isinstance(txGrid, L.DeviceArray) else ctx.to_device(L.as_c128_f(txGrid))
    zone = np.asarray(rdrEstParams.cfarEstZone, dtype=np.float64)
    mp = L.Music2dParams(float(rdrEstParams.fc), float(rdrEstParams.Tsri), float(bsParams.scs) * 1e3, float(zone[0, 1]), float(zone[1, 1]) * 2.0)
    ep = est_block(rdrEstParams)
    res = L.EstResult()
    ctx.check(ctx.lib.isac_music2d_dev(ctx.handle, C.byref(ep), C.byref(mp), C.c_void_p(d_rx.ptr), C.c_void_p(d_tx.ptr),
                                       C.c_int32(K), C.c_int32(Ls), C.c_int32(A), C.byref(res)))
    return SimpleNamespace(aziEst=np.array(res.azi_est[: res.n_azi]), eleEst=np.array(res.ele_est[: res.n_azi]),
                           rngEst=np.array(res.rng_est[: res.n_rng]), velEst=np.array(res.vel_est[: res.n_vel]), L=int(res.num_dets))
