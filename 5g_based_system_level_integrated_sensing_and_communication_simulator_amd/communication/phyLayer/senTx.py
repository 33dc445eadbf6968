"""Mono-static sensing transmit accumulation of ``gNBPhy.phyTx`` (+communication/+phyLayer/gNBPhy.m:591-612), device resident.

``SenTx`` mirrors the two gNBPhy properties ``senTxGrid`` / ``senTxWave`` (gNBPhy.m:48-52): every ``append(txGrid, currSlot)``
is one phyTx call that carried PDSCH -- the slot grid is OFDM-modulated (nrOFDMModulate, :599), scaled by signalAmp (:592,:602)
and appended in a 'D' slot, zeros of the same size are appended otherwise (:605-612).  The arrays live in HBM and feed
``sensing.monoStaticSensing`` / ``sensing.estimation.fft2D`` without crossing PCIe."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ... import _lib as L
from ...sensing._marshal import carrier_block


def determineSlotType(tddPattern: str, slotIdx: int) -> str:
    """+communication/determineSlotType.m:5."""
    return tddPattern[slotIdx % len(tddPattern)]


def signalAmp(txPower_dBm: float, nfft: int, nSc: int, numTxAnts: int) -> float:
    """gNBPhy.m:592: db2mag(TxPower-30)*sqrt(Nfft^2/(size(txGrid,1)*NumTxAnts))."""
    return 10.0 ** ((txPower_dBm - 30.0) / 20.0) * float(np.sqrt(nfft ** 2 / (nSc * numTxAnts)))


def nrOFDMModulate(carrierInfo, grid, *, nSlot=0, windowing=0, amplitude=1.0, nfft=None, ctx=None):
    """waveform = amplitude * nrOFDMModulate(carrier, grid) with carrier.NSlot = nSlot and the toolbox's raised-cosine windowing over
    ``windowing`` samples (pass nrOFDMInfo(carrier).Windowing; 0 = plain CP-OFDM).  numpy in -> numpy out; DeviceArray in -> DeviceArray out."""
    dev = isinstance(grid, L.DeviceArray)
    ctx = ctx or (grid.ctx if dev else L.default_context())
    k, l, a = grid.shape
    car = carrier_block(carrierInfo, nfft)
    d_g = grid if dev else ctx.to_device(L.as_c128_f(grid))
    spf = int(carrierInfo.SubcarrierSpacing) // 15
    sym0 = (int(nSlot) % spf) * 14
    t0, t1 = C.c_int64(0), C.c_int64(0)
    ctx.lib.isac_ofdm_waveform_length(C.byref(car), C.c_int32(sym0 + l), C.byref(t1))
    ctx.lib.isac_ofdm_waveform_length(C.byref(car), C.c_int32(sym0), C.byref(t0))
    t_len = int(t1.value - t0.value)
    d_w = ctx.empty((t_len, a))
    ctx.check(ctx.lib.isac_ofdm_modulate_windowed_dev(ctx.handle, C.c_void_p(d_g.ptr), C.c_int32(l), C.c_int32(a), C.byref(car), C.c_double(amplitude),
                                                      C.c_int32(int(nSlot)), C.c_int32(int(windowing)), C.c_void_p(d_w.ptr), C.c_int64(t_len)))
    return d_w if dev else d_w.numpy()


class SenTx:
    """Device-resident senTxGrid [nSc x 14*maxSlots x A] / senTxWave [T x A].  ``maxSlots`` = number of PDSCH-carrying slots of the
    sensing interval (the arrays are sized once; MATLAB grows them with cat())."""

    def __init__(self, carrierInfo, numTxAnts: int, maxSlots: int, tddPattern: str = "DDDSU", txPower: float = 46.0, windowing: int = 0,
                 nfft=None, ctx=None):
        self.ctx = ctx or L.default_context()
        self.car = carrier_block(carrierInfo, nfft)
        self.carrierInfo, self.A, self.maxSlots = carrierInfo, int(numTxAnts), int(maxSlots)
        self.tdd, self.txPower, self.windowing = tddPattern, float(txPower), int(windowing)
        self.K = self.car.n_sc
        # capacity in samples: a slot's length depends on its position in the subframe only through the long CPs; the longest possible slot bounds it
        spf = int(carrierInfo.SubcarrierSpacing) // 15
        t = C.c_int64(0)
        self.ctx.lib.isac_ofdm_waveform_length(C.byref(self.car), C.c_int32(14 * spf), C.byref(t))
        self._t_cap = (int(t.value) // spf + self.car.nfft) * self.maxSlots
        self._grid = self.ctx.empty((self.K, 14 * self.maxSlots, self.A))
        self._wave = self.ctx.empty((self._t_cap, self.A))
        self.nSlots, self.T = 0, 0

    def append(self, txGrid, currSlot: int):
        """One phyTx call with PDSCH: txGrid [nSc x 14 x A] (numpy or DeviceArray)."""
        if self.nSlots >= self.maxSlots:
            raise L.IsacError(6, "SenTx: more slots appended than maxSlots")
        ctx = self.ctx
        g = txGrid if isinstance(txGrid, L.DeviceArray) else ctx.to_device(L.as_c128_f(txGrid))
        if tuple(g.shape) != (self.K, 14, self.A):
            raise ValueError(f"txGrid must be [{self.K} x 14 x {self.A}]")
        is_dl = 1 if determineSlotType(self.tdd, int(currSlot)) == "D" else 0
        amp = signalAmp(self.txPower, self.car.nfft, self.K, self.A)
        t_len = C.c_int64(0)
        ctx.check(ctx.lib.isac_sentx_append_dev(ctx.handle, C.byref(self.car), C.c_int32(self.A), C.c_int32(int(currSlot)), C.c_int32(is_dl),
                                                C.c_void_p(g.ptr), C.c_double(amp), C.c_int32(self.windowing), C.c_void_p(self._grid.ptr),
                                                C.c_int32(14 * self.maxSlots), C.c_int32(14 * self.nSlots), C.c_void_p(self._wave.ptr),
                                                C.c_int64(self._t_cap), C.c_int64(self.T), C.byref(t_len)))
        self.nSlots += 1
        self.T += int(t_len.value)

    @property
    def senTxGrid(self) -> np.ndarray:
        return self._grid.numpy()[:, : 14 * self.nSlots, :]

    @property
    def senTxWave(self) -> np.ndarray:
        return self._wave.numpy()[: self.T, :]

    def device_arrays(self):
        """(senTxGrid, senTxWave) as DeviceArrays for the sensing call chain; valid when every one of the maxSlots slots has been appended
        (the planes are laid out for the full capacity)."""
        if self.nSlots != self.maxSlots:
            raise L.IsacError(1, "SenTx.device_arrays: append all maxSlots slots first (or read .senTxGrid/.senTxWave on the host)")
        if self.T == self._t_cap:
            return self._grid, self._wave
        w = self.ctx.empty((self.T, self.A))                  # compact the waveform planes to T rows
        for a in range(self.A):
            src = self._wave.ptr + 16 * self._t_cap * a
            self.ctx.check(self.ctx.lib.isac_memcpy_d2d(self.ctx.handle, C.c_void_p(w.ptr + 16 * self.T * a), C.c_void_p(src), C.c_size_t(16 * self.T)))
        return self._grid, w
