"""Oracle restatement of ``sensing.estimation.fft2D`` (+sensing/+estimation/fft2D.m:1-204).

``rdm_literal`` follows fft2D.m:37-46 line by line, including the all-dimension
``ifftshift``/``fftshift`` and the nIFFT-long "Doppler" window applied on the
(shifted) range axis.  ``rdm_explicit`` is the algebraically equivalent form the
HIP kernels implement; tests assert the two agree to 0.
TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
from scipy import fft as sfft

from .matlab_compat import kaiser, unique_stable
from .cfar import ca_cfar2d
from .music import music_doa

_WORKERS = -1


def rdm_literal(rx_grid: np.ndarray, tx_grid: np.ndarray, n_ifft: int, n_fft: int) -> np.ndarray:
    """fft2D.m:37-46 -> rdm [nIFFT x nFFT x nAnts]."""
    n_sc, n_sym, n_ants = rx_grid.shape
    channel = rx_grid * np.conj(tx_grid)                                   # :37 (pagectranspose(pagetranspose(.)) == conj)
    rng_win = kaiser(n_sc, 3.0)[:, None, None]                             # :40,:146  repmat(kaiser(nSc,3),[1 nSym])
    dop_win = kaiser(n_ifft, 3.0)[:, None, None]                           # :147      repmat(kaiser(nIFFT,3),[1 nSym])
    chl = channel * rng_win                                                # :43
    rng_ifft = sfft.ifft(chl, n=n_ifft, axis=0, workers=_WORKERS) * np.sqrt(n_ifft)
    rng_ifft = sfft.ifftshift(rng_ifft)                                    # :44  ALL dims
    rng_ifft = rng_ifft * dop_win                                          # :45
    rdm = sfft.fft(rng_ifft, n=n_fft, axis=1, workers=_WORKERS) / np.sqrt(n_fft)
    return sfft.fftshift(rdm)                                              # :46  ALL dims


def rdm_explicit(rx_grid: np.ndarray, tx_grid: np.ndarray, n_ifft: int, n_fft: int) -> np.ndarray:
    """Equivalent explicit form: effective range window indexed by the shifted bin,
    slow-time half-rotation before the zero-padded Doppler FFT, Doppler-axis fftshift,
    antenna order untouched."""
    n_sc, n_sym, n_ants = rx_grid.shape
    channel = rx_grid * np.conj(tx_grid) * kaiser(n_sc, 3.0)[:, None, None]
    r = sfft.ifft(channel, n=n_ifft, axis=0, workers=_WORKERS) * np.sqrt(n_ifft)
    # final[i] = R[i] * w[j] with j the index i lands on after fftshift(ifftshift-applied window):
    # out = fftshift(ifftshift(R) .* w)  ==  R .* fftshift(w)   (fftshift undoes ifftshift)
    w_eff = sfft.fftshift(kaiser(n_ifft, 3.0))
    r = r * w_eff[:, None, None]
    r = sfft.ifftshift(r, axes=1)
    d = sfft.fft(r, n=n_fft, axis=1, workers=_WORKERS) / np.sqrt(n_fft)
    return sfft.fftshift(d, axes=1)


def covariance(rx_grid: np.ndarray) -> np.ndarray:
    """fft2D.m:106-107.  ``reshape(rxGrid, nSc*nSym, nAnts)'`` is a CONJUGATE transpose,
    so Ra = X X^H / N with X = G^H, i.e. Ra[a,b] = sum_n conj(G[n,a]) G[n,b] / N."""
    n_sc, n_sym, n_ants = rx_grid.shape
    g = rx_grid.reshape(n_sc * n_sym, n_ants, order="F")
    ra = (g.conj().T @ g) / (n_sc * n_sym)
    return 0.5 * (ra + ra.conj().T)          # zherk returns an exactly Hermitian matrix


def detect_per_antenna(rdm: np.ndarray, cfar, r_res: float, v_res: float, n_fft: int):
    """fft2D.m:59-96 -> per-antenna detection lists (1-based [2 x D]) + sorted estimates."""
    n_ants = rdm.shape[2]
    all_rng, all_vel, dets = [], [], []
    for a in range(n_ants):
        rd = np.abs(rdm[:, :, a]) ** 2                                     # :61
        det = ca_cfar2d(rd, cfar.CUTIdx, cfar.Pfa, cfar.GuardBandSize, cfar.TrainingBandSize)   # :62
        dets.append(det)
        peaks = rd[det[0] - 1, det[1] - 1]                                 # :74
        rng_est = (det[0] - 1) * r_res                                     # :77,:81
        vel_est = (det[1] - n_fft / 2 - 1) * v_res                         # :78,:82
        order = np.argsort(-peaks, kind="stable")                          # :89 sort(peaks,'descend')
        all_rng.append(rng_est[order])
        all_vel.append(vel_est[order])
    return dets, np.concatenate(all_rng), np.concatenate(all_vel)


def fft2d(rp, cfar, rx_grid: np.ndarray, tx_grid: np.ndarray, return_debug: bool = False, rdm_fn=None):
    """fft2D.m:1-204 -> estResults{rngEst, velEst, aziEst, eleEst}.

    Raises ValueError when no CUT is detected (findpeaks 'NPeaks' = 0 error, music.m:102;
    cellSimulation.m:196-202 turns it into senResults = NaN).  ``rdm_fn``: ``rdm_literal`` (default, line-by-line) or
    ``rdm_explicit`` (bit-identical, without the all-dimension shift copies: the timed CPU baseline uses it)."""
    n_ifft, n_fft = int(rp.nIFFT), int(rp.nFFT)
    rdm = (rdm_fn or rdm_literal)(rx_grid, tx_grid, n_ifft, n_fft)
    dets, all_rng, all_vel = detect_per_antenna(rdm, cfar, rp.rRes, rp.vRes, n_fft)
    est = SimpleNamespace(rngEst=unique_stable(all_rng), velEst=unique_stable(all_vel))   # :99,:102
    ra = covariance(rx_grid)                                               # :106-107
    num_dets = est.rngEst.size                                             # :110
    _, azi, ele = music_doa(num_dets, rp, ra)                              # :111
    est.aziEst, est.eleEst = azi, ele                                      # :114-115
    if return_debug:
        return est, SimpleNamespace(rdm=rdm, detections=dets, Ra=ra)
    return est
