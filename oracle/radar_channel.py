"""Oracle restatement of ``sensing.channelModels.basicRadarChannel``
(+sensing/+channelModels/basicRadarChannel.m:1-76) and ``sensing.monoStaticSensing``
(+sensing/monoStaticSensing.m:1-23).  TEST INFRASTRUCTURE ONLY.

The AWGN draw (``randn``, basicRadarChannel.m:68) is MATLAB's mt19937ar+ziggurat
stream and cannot be reproduced outside MATLAB; the oracle takes the noise as an
explicit argument (unit-variance real/imag parts, scaled by sqrt(N0/2) here).
"""
from __future__ import annotations

import numpy as np

from .matlab_compat import LIGHTSPEED
from .ofdm import ofdm_demodulate


def basic_radar_channel(tx_waveform: np.ndarray, rp, los, noise_unit: np.ndarray | None):
    """tx_waveform [T x A] complex128; ``rp`` from radar_params(); ``los`` [Q] 0/1;
    ``noise_unit`` [T x A] complex128 with N(0,1) real and imaginary parts (or None
    for a noiseless run).  Returns rxWaveform [T x A]."""
    tx = np.asarray(tx_waveform, dtype=np.complex128)
    t_len, n_ants = tx.shape                               # :8
    c = LIGHTSPEED                                         # :11
    fc = rp.fc
    lam = c / fc
    fs = rp.fs
    ts = 1.0 / fs                                          # :15
    q = int(rp.nTargets)
    path_delay = 2.0 * np.asarray(rp.range, dtype=np.float64) / c          # :21
    shift = np.ceil(path_delay / ts).astype(np.int64)                       # :22
    fd = 2.0 * np.asarray(rp.velocity, dtype=np.float64) / lam              # :25

    t_axis = np.arange(t_len, dtype=np.float64) * ts                        # :29  (0:Ts:Ts*(T-1)).'
    phase_tx = np.exp(2j * np.pi * fc * t_axis)                             # :30
    tx = tx * phase_tx[:, None]                                             # :31

    lsf = np.asarray(rp.largeScaleFading, dtype=np.float64)
    sv = np.asarray(rp.RxSteeringVec, dtype=np.complex128)                  # :35 [A x Q]
    echoes = []
    los = np.asarray(los).reshape(-1)
    for i in range(q):
        if los[i] == 1:                                                     # :40
            s = int(shift[i])
            e = np.concatenate([np.zeros((s, n_ants), dtype=np.complex128), tx[: t_len - s]], axis=0)  # :42
            ph = np.exp(2j * np.pi * fd[i] * (np.arange(e.shape[0], dtype=np.float64) * ts))          # :43-44
            e = e * ph[:, None]                                             # :45
            e = e * lsf[i]                                                  # :48
            e = (e @ sv[:, i])[:, None] * sv[:, i][None, :]                 # :51  (e*a)*a.'
            if e.shape[0] < t_len:                                          # :54-57
                e = np.concatenate([e, np.zeros((t_len - e.shape[0], n_ants), dtype=np.complex128)])
            echoes.append(e)
    if not echoes:
        # :59,:64  sum(cat(3, [] ...), 3) -> [] ; downstream demodulation errors
        raise ValueError("basicRadarChannel: no LoS target -> empty rxWaveform")
    rx = echoes[0]
    for e in echoes[1:]:                                                    # :64
        rx = rx + e
    if noise_unit is not None:                                              # :67-69
        n0 = np.sqrt(rp.N0 / 2.0)
        rx = rx + n0 * noise_unit
    phase_rx = np.exp(-2j * np.pi * fc * (np.arange(rx.shape[0], dtype=np.float64) * ts))   # :72-73
    return rx * phase_rx[:, None]                                           # :74


def mono_static_sensing(tx_waveform, tx_dimension, carrier_info, rp, los, noise_unit,
                        nfft: int = 4096):
    """monoStaticSensing.m:1-23 -> echoGrid [K x L x A]."""
    echo = basic_radar_channel(tx_waveform, rp, los, noise_unit)            # :13
    n_sc = carrier_info.NRBsDL * 12
    grid = ofdm_demodulate(echo, n_sc, nfft, carrier_info.SubcarrierSpacing)   # :16
    if grid.shape[1] < tx_dimension[1]:                                     # :19-21
        pad = np.zeros((grid.shape[0], tx_dimension[1] - grid.shape[1], grid.shape[2]), dtype=np.complex128)
        grid = np.concatenate([grid, pad], axis=1)
    return grid
