"""Oracle restatement of 5G Toolbox CP-OFDM (de)modulation as the sensing path uses it
(monoStaticSensing.m:8-16 -> nrOFDMDemodulate; gNBPhy.m:599 -> nrOFDMModulate).

Toolbox behaviour restated from its documentation: normal CP,
longer CP on the first symbol of every half-subframe (0.5 ms), demodulation FFT
window starting ``fix(0.5*CP)`` samples into the CP with the resulting circular
offset compensated per bin, unscaled ``fft``/scaled ``ifft``, central-K bins.
The modulator here deliberately applies NO inter-symbol windowing (synthetic
inputs are ours).  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import numpy as np
from scipy import fft as sfft

_WORKERS = -1  # scipy.fft: use every host core (CPU baseline states the count)


def cp_lengths(nfft: int, scs_khz: float, n_symbols: int, first_symbol: int = 0) -> np.ndarray:
    """CP length of each of ``n_symbols`` consecutive symbols (normal CP).

    TS 38.211 5.3.1: N_cp = 144*kappa*2^-mu, plus 16*kappa for l = 0 and l = 7*2^mu
    within a subframe; scaled to the actual Nfft (Nfft/2048 * 2^mu samples per 15 kHz unit).
    """
    mu = int(round(np.log2(scs_khz / 15.0)))
    scale = nfft / 2048.0
    base = int(round(144 * scale))
    extra = int(round(16 * scale * (2 ** mu)))
    sym_per_subframe = 14 * 2 ** mu
    l = (first_symbol + np.arange(n_symbols)) % sym_per_subframe
    long_cp = (l % (7 * 2 ** mu)) == 0
    return np.where(long_cp, base + extra, base).astype(np.int64)


def symbol_starts(nfft: int, scs_khz: float, n_symbols: int, first_symbol: int = 0) -> tuple[np.ndarray, np.ndarray]:
    """(start sample of each symbol's CP, CP length); ``first_symbol``: index of the first symbol inside its subframe
    (carrier.NSlot * 14 for a slot grid, gNBPhy.m:579)."""
    cps = cp_lengths(nfft, scs_khz, n_symbols, first_symbol)
    lens = cps + nfft
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    return starts.astype(np.int64), cps


def raised_cosine_edge(n_w: int) -> np.ndarray:
    """Rising edge of the OFDM symbol window, LTE/5G Toolbox form (lteOFDMModulate / nrOFDMModulate documentation):
    w(i) = 0.5 (1 - sin(pi (N_W + 1 - 2 i) / (2 N_W))), i = 1..N_W; the falling edge is its mirror, rise + fall = 1."""
    i = np.arange(1, n_w + 1, dtype=np.float64)
    return 0.5 * (1.0 - np.sin(np.pi * (n_w + 1 - 2.0 * i) / (2.0 * n_w)))


def ofdm_modulate(grid: np.ndarray, nfft: int, scs_khz: float, windowing: int = 0, first_symbol: int = 0) -> np.ndarray:
    """grid [K x L x A] -> waveform [T x A]; CP-OFDM, ifft (1/Nfft) scaling.

    ``windowing`` = N_W > 0 restates nrOFDMModulate's raised-cosine windowing and overlap (toolbox behaviour from its public
    documentation -- UNVERIFIABLE here): every symbol is cyclically extended by N_W samples in front of its CP, the first
    and last N_W samples of the extended symbol are tapered by the raised-cosine edge, consecutive symbols overlap-add over
    N_W samples, and the head of the first symbol wraps onto the tail of the last one (the waveform loops seamlessly).
    The reference calls nrOFDMModulate with the toolbox default (gNBPhy.m:599); that default value comes from
    nrOFDMInfo(carrier).Windowing on the MATLAB side and is passed in explicitly here.  0 = no windowing."""
    k, l, a = grid.shape
    starts, cps = symbol_starts(nfft, scs_khz, l, first_symbol)
    total = int(starts[-1] + cps[-1] + nfft) if l else 0
    wave = np.zeros((total, a), dtype=np.complex128)
    first = (nfft - k) // 2
    full = np.zeros((nfft, l, a), dtype=np.complex128)
    full[first:first + k] = grid
    td = sfft.ifft(sfft.ifftshift(full, axes=0), axis=0, workers=_WORKERS)
    for s in range(l):
        cp = int(cps[s])
        o = int(starts[s])
        wave[o:o + cp] = td[nfft - cp:, s, :]
        wave[o + cp:o + cp + nfft] = td[:, s, :]
    n_w = int(windowing)
    if n_w > 0 and l > 0:
        if n_w > int(cps.min()):
            raise ValueError("ofdm_modulate: windowing longer than the cyclic prefix")
        rise = raised_cosine_edge(n_w)[:, None]
        fall = rise[::-1]
        for s in range(l):
            cp = int(cps[s])
            nxt = (s + 1) % l                                     # the last symbol's tail takes the FIRST symbol's head
            cpn = int(cps[nxt])
            e = int(starts[s]) + cp + nfft                        # end of symbol s
            head = td[nfft - cpn - n_w:nfft - cpn, nxt, :]        # N_W samples in front of the next symbol's CP (cyclic extension)
            wave[e - n_w:e] = fall * wave[e - n_w:e] + rise * head
    return wave


def ofdm_demodulate(wave: np.ndarray, n_sc: int, nfft: int, scs_khz: float,
                    cp_fraction: float = 0.5) -> np.ndarray:
    """waveform [T x R] -> grid [K x L x R] for the whole symbols contained in T
    (``nrOFDMDemodulate(carrier, wave)`` with NSlot = 0; monoStaticSensing.m:16)."""
    t, r = wave.shape
    # number of whole symbols that fit
    n_max = int(t // nfft) + 1
    starts, cps = symbol_starts(nfft, scs_khz, n_max)
    ends = starts + cps + nfft
    l = int(np.searchsorted(ends, t, side="right"))
    if l == 0:
        raise ValueError("ofdm_demodulate: waveform shorter than one OFDM symbol")
    first = (nfft - n_sc) // 2
    kbin = np.arange(n_sc) + first - nfft // 2        # signed bin of each kept subcarrier
    grid = np.empty((n_sc, l, r), dtype=np.complex128)
    for s in range(l):
        cp = int(cps[s])
        off = int(np.fix(cp * cp_fraction))
        w0 = int(starts[s]) + off
        x = sfft.fft(wave[w0:w0 + nfft], axis=0, workers=_WORKERS)
        x = sfft.fftshift(x, axes=0)[first:first + n_sc]
        d = cp - off                                    # window leads the useful part by d samples
        grid[:, s, :] = x * np.exp(2j * np.pi * kbin * d / nfft)[:, None]
    return grid
