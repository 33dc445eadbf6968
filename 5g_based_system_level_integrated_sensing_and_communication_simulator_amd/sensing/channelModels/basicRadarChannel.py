"""sensing.channelModels.basicRadarChannel (+sensing/+channelModels/basicRadarChannel.m:1-76)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ... import _lib as L
from .._marshal import ChannelBlock, los_array


def basicRadarChannel(txWaveform, radarParams, targetLoSConditions, *, noise=None, seed=None, ctx=None):
    """rxWaveform = basicRadarChannel(txWaveform, radarParams, targetLoSConditions).

    ``txWaveform`` [T x nTxAnts]: numpy array (host path, result is a numpy array) or
    DeviceArray (device path, result stays in HBM).

    The reference draws AWGN with ``randn`` (:68).  Here: ``noise`` = the
    ``randn(size)+1j*randn(size)`` array to inject (parity mode), or ``seed`` for the
    on-device Philox generator (performance mode), or neither for a noiseless run.
    All targets NLoS -> IsacError NO_LOS (the reference produces an empty rxWaveform, :59,:64)."""
    dev = isinstance(txWaveform, L.DeviceArray)
    ctx = ctx or (txWaveform.ctx if dev else L.default_context())
    T, A = (txWaveform.shape if dev else np.shape(txWaveform))
    cb = ChannelBlock(radarParams)
    if A != cb.block.n_ants:
        raise ValueError("txWaveform antenna dimension differs from radarParams.nTxAnts")
    los = los_array(targetLoSConditions, cb.block.n_targets)
    mode = L.NOISE_INJECTED if noise is not None else (L.NOISE_PHILOX if seed is not None else L.NOISE_NONE)
    lib = ctx.lib
    if dev:
        nz = noise if (noise is None or isinstance(noise, L.DeviceArray)) else ctx.to_device(L.as_c128_f(noise))
        out = ctx.empty((T, A))
        ctx.check(lib.isac_basic_radar_channel_dev(ctx.handle, C.c_void_p(txWaveform.ptr), C.c_int64(T), C.byref(cb.block),
                                                   los.ctypes.data_as(C.c_void_p), C.c_int(mode),
                                                   C.c_void_p(nz.ptr if nz is not None else 0), C.c_uint64(seed or 0),
                                                   C.c_void_p(out.ptr)))
        return out
    tx = L.as_c128_f(txWaveform)
    nz = None if noise is None else L.as_c128_f(noise)
    out = np.empty((T, A), dtype=np.complex128, order="F")
    ctx.check(lib.isac_basic_radar_channel(ctx.handle, tx.ctypes.data_as(C.c_void_p), C.c_int64(T), C.byref(cb.block),
                                           los.ctypes.data_as(C.c_void_p), C.c_int(mode),
                                           nz.ctypes.data_as(C.c_void_p) if nz is not None else C.c_void_p(0),
                                           C.c_uint64(seed or 0), out.ctypes.data_as(C.c_void_p)))
    return out
