"""sensing.estimation.fft2D (+sensing/+estimation/fft2D.m:1-204)."""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np

from ... import _lib as L
from .._marshal import est_block


def _cut_rectangle(cut: np.ndarray):
    """cfar2D.m:23-24 always builds a full rectangle enumerated rows-fastest; recover its corners."""
    cut = np.asarray(cut)
    if cut.ndim != 2 or cut.shape[0] != 2 or cut.shape[1] == 0:
        raise ValueError("cfar.CUTIdx must be a non-empty [2 x nCUT] index list")
    r0, r1 = int(cut[0].min()), int(cut[0].max())
    c0, c1 = int(cut[1].min()), int(cut[1].max())
    cc, rr = np.meshgrid(np.arange(c0, c1 + 1), np.arange(r0, r1 + 1))
    if cut.shape[1] != rr.size or not (np.array_equal(cut[0], rr.ravel(order="F")) and np.array_equal(cut[1], cc.ravel(order="F"))):
        raise L.IsacError(7, "fft2D: CUTIdx is not the rows-fastest rectangle sensing.detection.cfar2D builds")
    return r0, r1, c0, c1


def _cfar_block(cfar) -> L.CfarConfig:
    """cfarConfig (CUTIdx + detector) -> isac_cfar_config; the rectangle recovery is cached on the object."""
    det = cfar.cfarDetector2D
    key = id(cfar.CUTIdx)
    rect = getattr(cfar, "_rect", None)
    if rect is None or rect[0] != key:
        rect = (key, _cut_rectangle(cfar.CUTIdx))
        cfar._rect = rect
    r0, r1, c0, c1 = rect[1]
    return L.CfarConfig(float(det.ProbabilityFalseAlarm), (C.c_int32 * 2)(*det.GuardBandSize), (C.c_int32 * 2)(*det.TrainingBandSize),
                        r0, r1, c0, c1)


def fft2D(radarEstParams, cfar, rxGrid, txGrid, *, ctx=None, return_debug=False, reuse_range=False):
    """estResults = sensing.estimation.fft2D(radarEstParams, cfar, rxGrid, txGrid).

    Returns a namespace with ``rngEst, velEst, aziEst, eleEst`` (fft2D.m:102,114-115).  Raises
    IsacError(NO_DETECTION) where the reference errors inside findpeaks (zero detections); the
    reference's caller turns any error into ``senResults = NaN`` (cellSimulation.m:196-202).
    Plotting (fft2D.m:119) is not part of the hot path.
    ``reuse_range=True`` (device grids): consume the range rows the preceding ``monoStaticSensing(..., fuse_fft2d=...)``
    call cached on this context (isac_fft2d_submit_cached_dev); an error if there are none."""
    lazy = hasattr(rxGrid, "materialize")                 # LazyEchoGrid of monoStaticSensing(..., lazy=True): rxGrid goes to the library as NULL
    if lazy and not reuse_range:
        raise ValueError("a lazy echo grid is consumed with reuse_range=True (the fused call has already run its range stage)")
    dev = isinstance(rxGrid, L.DeviceArray) or lazy
    if dev != isinstance(txGrid, L.DeviceArray):
        raise ValueError("rxGrid and txGrid must both be numpy arrays or both DeviceArrays")
    ctx = ctx or (rxGrid.ctx if dev else L.default_context())
    K, Lsym, A = (rxGrid.shape if dev else np.shape(rxGrid))
    if tuple(txGrid.shape) != (K, Lsym, A):
        raise ValueError("rxGrid and txGrid must have identical [nSc x nSym x nAnts] shape")
    cf = _cfar_block(cfar)
    ep = est_block(radarEstParams)
    res = L.EstResult()
    lib = ctx.lib
    if dev and reuse_range:
        st = lib.isac_fft2d_submit_cached_dev(ctx.handle, C.byref(ep), C.byref(cf), C.c_void_p(rxGrid.ptr or None), C.c_void_p(txGrid.ptr),
                                              C.c_int32(K), C.c_int32(Lsym), C.c_int32(A))
        if st == 0:
            st = lib.isac_fft2d_collect(ctx.handle, C.byref(res))
    elif reuse_range:
        raise ValueError("reuse_range needs the DeviceArray grids of a monoStaticSensing(fuse_fft2d=...) call")
    elif dev:
        st = lib.isac_fft2d_dev(ctx.handle, C.byref(ep), C.byref(cf), C.c_void_p(rxGrid.ptr), C.c_void_p(txGrid.ptr),
                                C.c_int32(K), C.c_int32(Lsym), C.c_int32(A), C.byref(res))
    else:
        rx, tx = L.as_c128_f(rxGrid), L.as_c128_f(txGrid)
        st = lib.isac_fft2d(ctx.handle, C.byref(ep), C.byref(cf), rx.ctypes.data_as(C.c_void_p), tx.ctypes.data_as(C.c_void_p),
                            C.c_int32(K), C.c_int32(Lsym), C.c_int32(A), C.byref(res))
    ctx.check(st)
    est = SimpleNamespace(rngEst=np.array(res.rng_est[: res.n_rng]), velEst=np.array(res.vel_est[: res.n_vel]),
                          aziEst=np.array(res.azi_est[: res.n_azi]), eleEst=np.array(res.ele_est[: res.n_azi]))
    if return_debug:
        return est, fft2D_debug(ctx, A)
    return est


def fft2D_submit(radarEstParams, cfar, rxGrid, txGrid, *, ctx=None, reuse_range=False):
    """Asynchronous half of fft2D for device-resident grids: enqueues every kernel and the result
    copy on ``ctx`` without waiting (isac_fft2d_submit_dev).  Pair with fft2D_collect(ctx).  Lets a
    host loop keep several cells / CPIs in flight on different contexts."""
    if not ((isinstance(rxGrid, L.DeviceArray) or (hasattr(rxGrid, "materialize") and reuse_range)) and isinstance(txGrid, L.DeviceArray)):
        raise ValueError("fft2D_submit needs DeviceArray grids (or a LazyEchoGrid with reuse_range=True)")
    ctx = ctx or rxGrid.ctx
    K, Lsym, A = rxGrid.shape
    if tuple(txGrid.shape) != (K, Lsym, A):
        raise ValueError("rxGrid and txGrid must have identical [nSc x nSym x nAnts] shape")
    cf = _cfar_block(cfar)
    ep = est_block(radarEstParams)
    fn = ctx.lib.isac_fft2d_submit_cached_dev if reuse_range else ctx.lib.isac_fft2d_submit_dev
    ctx.check(fn(ctx.handle, C.byref(ep), C.byref(cf), C.c_void_p(rxGrid.ptr or None), C.c_void_p(txGrid.ptr), C.c_int32(K), C.c_int32(Lsym), C.c_int32(A)))
    return ctx


def fft2D_collect(ctx):
    """Waits for the submitted fft2D on ``ctx`` and returns estResults (raises IsacError like fft2D)."""
    res = L.EstResult()
    ctx.check(ctx.lib.isac_fft2d_collect(ctx.handle, C.byref(res)))
    return SimpleNamespace(rngEst=np.array(res.rng_est[: res.n_rng]), velEst=np.array(res.vel_est[: res.n_vel]),
                           aziEst=np.array(res.azi_est[: res.n_azi]), eleEst=np.array(res.ele_est[: res.n_azi]))


def fft2D_debug(ctx, A):
    """Detection lists (CUT order, per antenna), the |rdm|^2 window, Ra and the MUSIC spectrum of the
    last fft2D call on ``ctx`` -- what the parity tests compare against the oracle."""
    lib = ctx.lib
    n_total = C.c_int32(0)
    off = np.zeros(A + 1, dtype=np.int32)
    ctx.check(lib.isac_fft2d_get_detections(ctx.handle, None, None, C.c_int32(1 << 30), off.ctypes.data_as(C.c_void_p), C.byref(n_total)))
    n = int(n_total.value)
    idx = np.zeros((2, max(n, 1)), dtype=np.int32, order="F")
    pw = np.zeros(max(n, 1), dtype=np.float64)
    ctx.check(lib.isac_fft2d_get_detections(ctx.handle, idx.ctypes.data_as(C.c_void_p), pw.ctypes.data_as(C.c_void_p),
                                            C.c_int32(max(n, 1)), off.ctypes.data_as(C.c_void_p), C.byref(n_total)))
    dets = [idx[:, off[a]:off[a + 1]].astype(np.int64) for a in range(A)]
    dims = (C.c_int32 * 3)()
    fr, fc = C.c_int32(0), C.c_int32(0)
    ctx.check(lib.isac_fft2d_get_power_window(ctx.handle, None, C.c_int64(0), dims, C.byref(fr), C.byref(fc)))
    pwin = np.zeros(tuple(dims), dtype=np.float64, order="F")
    ctx.check(lib.isac_fft2d_get_power_window(ctx.handle, pwin.ctypes.data_as(C.c_void_p), C.c_int64(pwin.size), dims, C.byref(fr), C.byref(fc)))
    ra = np.zeros((A, A), dtype=np.complex128, order="F")
    ctx.check(lib.isac_fft2d_get_covariance(ctx.handle, ra.ctypes.data_as(C.c_void_p), C.c_int32(A)))
    ns = C.c_int32(0)
    ctx.check(lib.isac_fft2d_get_music_spectrum(ctx.handle, None, C.c_int32(0), C.byref(ns)))
    spec = np.zeros(max(ns.value, 1), dtype=np.float64)
    ctx.check(lib.isac_fft2d_get_music_spectrum(ctx.handle, spec.ctypes.data_as(C.c_void_p), C.c_int32(spec.size), C.byref(ns)))
    return SimpleNamespace(detections=dets, det_pow=[pw[off[a]:off[a + 1]] for a in range(A)], power_window=pwin,
                           first_row=int(fr.value), first_col=int(fc.value), Ra=ra, spectrum_db=spec[: ns.value])
