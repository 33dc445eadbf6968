"""sensing.monoStaticSensing (+sensing/monoStaticSensing.m:1-23)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib as L
from ._marshal import ChannelBlock, carrier_block, los_array


class LazyEchoGrid:
    """echoGrid kept INSIDE a context by ``monoStaticSensing(..., fuse_fft2d=..., lazy=True)`` (isac_mono_static_sensing_fused_dev with d_echo_grid == NULL): hand it to
    ``fft2D(..., reuse_range=True)`` as rxGrid, or call ``.materialize()`` for the array (monoStaticSensing.m:1 returns it; cellSimulation.m:194-197 only passes it on to fft2D).
    Valid until the next echo call on the same context."""

    def __init__(self, ctx, shape):
        self.ctx, self.shape, self.ptr = ctx, tuple(int(v) for v in shape), 0

    def materialize(self, out=None):
        """The grid as a DeviceArray [nSc x nSym x nAnts] -- bit for bit what the non-lazy call stores (isac_echo_grid_materialize_dev)."""
        dims = (C.c_int32 * 3)()
        self.ctx.check(self.ctx.lib.isac_echo_grid_materialize_dev(self.ctx.handle, None, dims))
        if tuple(dims) != self.shape:
            raise L.IsacError(1, "the context's lazy echo grid has been replaced by a later call")
        if out is None:
            out = self.ctx.empty(self.shape)
        self.ctx.check(self.ctx.lib.isac_echo_grid_materialize_dev(self.ctx.handle, C.c_void_p(out.ptr), dims))
        return out

    def numpy(self):
        return self.materialize().numpy()


def monoStaticSensing(txWaveform, txDimension, carrierInfo, radarParams, targetLoSConditions, *,
                      noise=None, seed=None, nfft=None, ctx=None, out=None, fuse_fft2d=None, spectral_noise=None, noise_domain="time", lazy=False):
    """echoGrid = monoStaticSensing(txWaveform, txDimension, carrierInfo, radarParams, targetLoSConditions).

    Radar channel (:13) + OFDM demodulation (:16) + zero-padding of the symbol dimension up to
    ``txDimension(2)`` (:19-21), fused on the device: the time-domain echo is never written to HBM.
    numpy in -> numpy out; DeviceArray in -> DeviceArray out (``out``: optional pre-allocated
    DeviceArray to reuse between CPIs).  ``noise`` / ``seed`` as in basicRadarChannel.

    ``noise_domain="spectral"`` with ``seed`` (ISAC_NOISE_PHILOX_SPECTRAL) or ``spectral_noise`` = unit complex normals
    [nSc x nSym x nAnts] (ISAC_NOISE_INJECTED_SPECTRAL): the AWGN is drawn / supplied on the demodulated grid instead of
    per time sample -- the same distribution (the demodulator is unitary up to sqrt(Nfft) on disjoint windows), and the
    echo becomes sum_q a_q[r] D_q[k,l] + W with only Q demodulation FFTs per symbol (include/isac.h, isac_noise_mode).

    ``fuse_fft2d=(radarEstParams, cfar, txGrid)`` (device path): also run the range stage of the
    ``fft2D(radarEstParams, cfar, echoGrid, txGrid)`` call that follows while each echo column is still
    on chip (isac_mono_static_sensing_fused_dev); that fft2D call then skips re-reading echoGrid.
    Results are identical to the unfused sequence.

    ``lazy=True`` (with ``fuse_fft2d``): the echo grid stays inside the context and a LazyEchoGrid is returned -- on the spectral Philox route with 49..64 antennas and one or
    two LoS targets nothing is written to HBM at all (the covariance stage of fft2D re-forms the grid from its 12 MB of inputs); include/isac.h, 'LAZY echo grid'."""
    dev = isinstance(txWaveform, L.DeviceArray)
    ctx = ctx or (txWaveform.ctx if dev else L.default_context())
    T, A = (txWaveform.shape if dev else np.shape(txWaveform))
    cb = ChannelBlock(radarParams)
    if A != cb.block.n_ants:
        raise ValueError("txWaveform antenna dimension differs from radarParams.nTxAnts")
    los = los_array(targetLoSConditions, cb.block.n_targets)
    car = carrier_block(carrierInfo, nfft)
    if spectral_noise is not None:
        if noise is not None:
            raise ValueError("give either time-domain `noise` or `spectral_noise`, not both")
        mode, noise = L.NOISE_INJECTED_SPECTRAL, spectral_noise
    elif noise is not None:
        mode = L.NOISE_INJECTED
    elif seed is not None:
        if noise_domain not in ("time", "spectral"):
            raise ValueError("noise_domain must be 'time' or 'spectral'")
        mode = L.NOISE_PHILOX_SPECTRAL if noise_domain == "spectral" else L.NOISE_PHILOX
    else:
        mode = L.NOISE_NONE
    lib = ctx.lib
    lw = C.c_int32(0)
    st = lib.isac_ofdm_symbol_count(C.byref(car), C.c_int64(T), C.byref(lw))
    if st != 0:
        raise L.IsacError(st, "isac_ofdm_symbol_count failed")
    l_out = max(int(lw.value), int(txDimension[1]))
    lo = C.c_int32(0)
    if dev:
        nz = noise if (noise is None or isinstance(noise, L.DeviceArray)) else ctx.to_device(L.as_c128_f(noise))
        shape = (car.n_sc, max(l_out, 1), A)
        if lazy:
            if fuse_fft2d is None or out is not None:
                raise ValueError("lazy=True needs fuse_fft2d=(radarEstParams, cfar, txGrid) and no `out` array")
            out = LazyEchoGrid(ctx, shape)
        elif out is None:
            out = ctx.empty(shape)
        elif tuple(out.shape) != shape:
            raise ValueError(f"out must have shape {shape}")
        if fuse_fft2d is not None:
            from .estimation.fft2D import _cfar_block
            from ._marshal import est_block
            est_params, cfar, tx_grid = fuse_fft2d
            if not isinstance(tx_grid, L.DeviceArray) or tuple(tx_grid.shape) != shape:
                raise ValueError("fuse_fft2d needs a DeviceArray txGrid with the echo grid's shape")
            ep, cf = est_block(est_params), _cfar_block(cfar)
            ctx.check(lib.isac_mono_static_sensing_fused_dev(ctx.handle, C.c_void_p(txWaveform.ptr), C.c_int64(T), C.c_int32(int(txDimension[1])),
                                                             C.byref(car), C.byref(cb.block), los.ctypes.data_as(C.c_void_p), C.c_int(mode),
                                                             C.c_void_p(nz.ptr if nz is not None else 0), C.c_uint64(seed or 0),
                                                             C.c_void_p(out.ptr or None), C.byref(lo), C.byref(ep), C.byref(cf), C.c_void_p(tx_grid.ptr)))
            return out
        ctx.check(lib.isac_mono_static_sensing_dev(ctx.handle, C.c_void_p(txWaveform.ptr), C.c_int64(T), C.c_int32(int(txDimension[1])),
                                                   C.byref(car), C.byref(cb.block), los.ctypes.data_as(C.c_void_p), C.c_int(mode),
                                                   C.c_void_p(nz.ptr if nz is not None else 0), C.c_uint64(seed or 0),
                                                   C.c_void_p(out.ptr), C.byref(lo)))
        return out
    tx = L.as_c128_f(txWaveform)
    nz = None if noise is None else L.as_c128_f(noise)
    out = np.empty((car.n_sc, max(l_out, 1), A), dtype=np.complex128, order="F")
    ctx.check(lib.isac_mono_static_sensing(ctx.handle, tx.ctypes.data_as(C.c_void_p), C.c_int64(T), C.c_int32(int(txDimension[1])),
                                           C.byref(car), C.byref(cb.block), los.ctypes.data_as(C.c_void_p), C.c_int(mode),
                                           nz.ctypes.data_as(C.c_void_p) if nz is not None else C.c_void_p(0),
                                           C.c_uint64(seed or 0), out.ctypes.data_as(C.c_void_p), C.byref(lo)))
    return out
