"""Context preparation for the one-call-per-cell drop-in (isac_ctx_reserve; no counterpart in the reference, whose first call simply is slow)."""
from __future__ import annotations

import ctypes as C

from .. import _lib as L
from ._marshal import ChannelBlock, carrier_block, est_block


def reserve(waveformLength, txDimension, carrierInfo, radarParams, cfar, *, nfft=None, warm_ms=0.0, ctx=None) -> float:
    """Prepare ``ctx`` for ``monoStaticSensing(txWaveform [waveformLength x nTxAnts], txDimension, carrierInfo, radarParams, ...)`` followed by
    ``fft2D(radarParams, cfar, echoGrid, txGrid)``: one or more DRY runs of exactly that chain on grids the library generates itself (code objects
    loaded, scratch sized, tables built, LDS attributes set, pinned staging allocated), repeated until ``warm_ms`` of wall time have passed so that
    the device clocks are up as well.  The reference calls the chain once per cell and simulation (cellSimulation.m:189-202): call this while the
    scenario is being set up.  Returns the wall time of the call in ms."""
    from .estimation.fft2D import _cfar_block
    ctx = ctx or L.default_context()
    cb = ChannelBlock(radarParams)
    car = carrier_block(carrierInfo, nfft)
    ep, cf = est_block(radarParams), _cfar_block(cfar)
    ms = C.c_double(0.0)
    ctx.check(ctx.lib.isac_ctx_reserve(ctx.handle, C.c_int64(int(waveformLength)), C.c_int32(int(txDimension[1])), C.byref(car), C.byref(cb.block),
                                       C.byref(ep), C.byref(cf), C.c_double(float(warm_ms)), C.byref(ms)))
    return float(ms.value)
