"""Many cells' ``monoStaticSensing -> fft2D`` pairs in two library calls (isac_sensing_submit_n / isac_sensing_collect_n; no counterpart in the reference, which runs
the pair once per cell inside its per-cell worker: cellSimulation.m:189-202, networkSimulation.m:47-60).  At small arrays the host loop, not the GPU, bounds a
per-cell call sequence; here the loop over cells runs inside the library."""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np

from .. import _lib as L
from ._marshal import ChannelBlock, carrier_block, est_block, los_array


class SensingBatch:
    """The marshalled arguments of one ``submitN`` call (kept alive until ``collectN``) -- build once, submit many times: ``seed`` may be advanced between calls."""

    def __init__(self, ctxs, txWaveforms, txGrids, txDimension, carrierInfo, radarParamsList, losList, cfar, *, seeds, nfft=None, noise_domain="spectral", echoGrids=None,
                 pace_us=0.0):
        from .estimation.fft2D import _cfar_block
        n = len(ctxs)
        if not (len(txWaveforms) == len(txGrids) == len(radarParamsList) == len(losList) == len(seeds) == n):
            raise ValueError("submitN: one context, waveform, grid, radarParams, LoS vector and seed per job")
        self.ctxs, self.n = list(ctxs), n
        self.car = carrier_block(carrierInfo, nfft)
        self.ep, self.cf = est_block(radarParamsList[0]), _cfar_block(cfar)
        self.blocks = [ChannelBlock(rp) for rp in radarParamsList]
        self.los = [los_array(lo, b.block.n_targets) for lo, b in zip(losList, self.blocks)]
        self.T = int(txWaveforms[0].shape[0])
        self.tx_dim_l = int(txDimension[1])
        mode = L.NOISE_PHILOX_SPECTRAL if noise_domain == "spectral" else L.NOISE_PHILOX
        self.jobs = (L.SensingJob * n)()
        self.keep = (txWaveforms, txGrids, echoGrids)
        for i in range(n):
            j = self.jobs[i]
            j.d_tx_wave, j.d_tx_grid = txWaveforms[i].ptr, txGrids[i].ptr
            j.d_echo_grid = echoGrids[i].ptr if echoGrids is not None and echoGrids[i] is not None else None
            j.rp = C.addressof(self.blocks[i].block)
            j.los = self.los[i].ctypes.data
            j.d_noise_unit, j.seed, j.noise_mode = None, int(seeds[i]), mode
        self.handles = (C.c_void_p * n)(*[c.handle for c in self.ctxs])
        self.status = (C.c_int32 * n)()
        self.out = (L.EstResult * n)()
        self.pace_us = float(pace_us)

    def submit(self):
        lib = self.ctxs[0].lib
        st = lib.isac_sensing_submit_n(self.handles, C.c_int32(self.n), self.jobs, C.c_int64(self.T), C.c_int32(self.tx_dim_l), C.byref(self.car), C.byref(self.ep),
                                       C.byref(self.cf), C.c_double(self.pace_us), self.status)
        if st != 0:
            raise L.IsacError(st, "isac_sensing_submit_n: malformed call")
        return self

    def collect(self):
        """[estResults | IsacError] per job, in order (an error is returned, not raised: the reference maps a failed cell to senResults = NaN, cellSimulation.m:196-202)."""
        lib = self.ctxs[0].lib
        st = lib.isac_sensing_collect_n(self.handles, C.c_int32(self.n), self.out, self.status)
        if st != 0:
            raise L.IsacError(st, "isac_sensing_collect_n: malformed call")
        res = []
        for i in range(self.n):
            if self.status[i] != 0:
                res.append(L.IsacError(self.status[i], (lib.isac_last_error(self.ctxs[i].handle) or b"").decode()))
                continue
            r = self.out[i]
            res.append(SimpleNamespace(rngEst=np.array(r.rng_est[: r.n_rng]), velEst=np.array(r.vel_est[: r.n_vel]), aziEst=np.array(r.azi_est[: r.n_azi]),
                                       eleEst=np.array(r.ele_est[: r.n_azi])))
        return res


def submitN(ctxs, txWaveforms, txGrids, txDimension, carrierInfo, radarParamsList, losList, cfar, **kw):
    """Enqueue job i -- monoStaticSensing(txWaveforms[i], ...) -> fft2D(radarParamsList[i], cfar, echoGrid, txGrids[i]) -- on ctxs[i] (DeviceArrays; distinct idle contexts);
    returns the SensingBatch to call ``.collect()`` on.  ``echoGrids=None``: every job's echo grid stays lazy."""
    return SensingBatch(ctxs, txWaveforms, txGrids, txDimension, carrierInfo, radarParamsList, losList, cfar, **kw).submit()
