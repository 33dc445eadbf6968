"""Shared marshalling between the MATLAB-style structs and the C parameter blocks."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib as L


def nr_ofdm_info(nrb: int, scs_khz: float):
    """Nfft / SampleRate as 5G Toolbox nrOFDMInfo(NRB, SCS) reports them (gNBPhy.m:772)."""
    import math
    from types import SimpleNamespace
    nfft = max(128, 2 ** math.ceil(math.log2(12 * nrb / 0.85)))
    return SimpleNamespace(Nfft=nfft, SampleRate=float(nfft * scs_khz * 1e3), SymbolsPerSlot=14)


def carrier_block(carrierInfo, nfft=None) -> L.Carrier:
    nrb = int(carrierInfo.NRBsDL)
    scs = int(carrierInfo.SubcarrierSpacing)
    if nfft is None:
        nfft = nr_ofdm_info(nrb, scs).Nfft
    return L.Carrier(12 * nrb, int(nfft), scs, 0)


class ChannelBlock:
    """Keeps the numpy buffers alive for the lifetime of the C struct."""

    def __init__(self, rp):
        self.range = L.as_f64(rp.range)
        self.velocity = L.as_f64(rp.velocity)
        self.lsf = L.as_f64(rp.largeScaleFading)
        self.steer = L.as_c128_f(rp.RxSteeringVec)
        q = int(rp.nTargets)
        a = int(rp.nTxAnts)
        if self.steer.shape != (a, q):
            raise ValueError("RxSteeringVec must be [nTxAnts x nTargets]")
        dp = C.POINTER(C.c_double)
        self.block = L.RadarChannelParams(float(rp.fc), float(rp.fs), float(rp.N0), a, q,
                                          self.range.ctypes.data_as(dp), self.velocity.ctypes.data_as(dp),
                                          self.lsf.ctypes.data_as(dp), self.steer.ctypes.data_as(C.c_void_p))


def los_array(targetLoSConditions, q: int) -> np.ndarray:
    los = np.ascontiguousarray(np.asarray(targetLoSConditions).reshape(-1) == 1, dtype=np.uint8)
    if los.size != q:
        raise ValueError("targetLoSConditions must have one entry per target")
    return los


def est_block(rp) -> L.EstParams:
    arr = getattr(rp, "antennaType", None)
    upa = 1 if getattr(arr, "kind", "ula") == "upa" else 0
    return L.EstParams(int(rp.nIFFT), int(rp.nFFT), float(rp.rRes), float(rp.vRes), upa,
                       int(getattr(arr, "nV", 0) or 0) if upa else 0, int(getattr(arr, "nH", 0) or 0) if upa else 0,
                       float(rp.azimuthScanScale), float(rp.azimuthScanGranularity),
                       float(rp.elevationScanScale), float(rp.elevationScanGranularity))
