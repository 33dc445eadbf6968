"""Oracle restatement of the DoA / MUSIC estimators:

* ``doaEstimation.music``   (+sensing/+estimation/+doaEstimation/music.m:1-125)  ULA branch
  (the UPA branch ends in the non-existent ``tools.find2DPeaks`` -- music.m:69 -- and is
  restated only up to the 2-D spectrum);
* ``doaEstimation.digitalBF`` / ``mvdrBF``  (digitalBF.m:55-86, mvdrBF.m:55-86) ULA branch;
* ``estimation.music2D``    (+sensing/+estimation/music2D.m:1-123) -- dead code in the
  reference (no caller) but named by BASELINE.json's north_star.

TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
from scipy import linalg

from .matlab_compat import EPS, LIGHTSPEED, sind, cosd, mag2db, findpeaks


def determine_num_targets(v) -> int:
    """music.m:109-125, applied to eigenvalues in eig()'s ASCENDING order (called at :22
    before the sort at :26).  Returns the 1-based argmax like MATLAB's [~,L] = max(.)."""
    v = np.asarray(v, dtype=np.float64).ravel()
    delta = -np.diff(v)                                                   # :113
    n = delta.size                                                        # :116
    half_mean = np.mean(delta[math.ceil((n + 1) / 2) - 1:])               # :117
    eps_ = 1.0                                                            # :121
    return int(np.argmax(delta - (1 + eps_) * half_mean)) + 1             # :123 first maximiser


def _noise_projector(ra, n_sig):
    """music.m:19-29: eig -> descending sort -> Uan Uan^H."""
    ra = np.asarray(ra, dtype=np.complex128)
    va, ua = linalg.eigh(ra)                                              # :19-20 ascending, real
    order = np.argsort(-va, kind="stable")                                # :26
    ua = ua[:, order]                                                     # :27
    uan = ua[:, n_sig:]                                                   # :28 (empty when L >= nAnts)
    return uan @ uan.conj().T, va                                         # :29


def ula_scan_angles(rp):
    gran = rp.azimuthScanGranularity
    a_max = rp.azimuthScanScale
    steps = int(math.floor((a_max + 1) / gran))                           # music.m:79
    return np.arange(steps) * gran - a_max / 2.0                          # :88


def music_spectrum_ula(uann, n_ants, rp):
    """music.m:82-96 -> PmusicdB [aSteps]."""
    d = 0.5                                                               # :12
    nn = np.arange(n_ants, dtype=np.float64)
    angles = ula_scan_angles(rp)
    p = np.empty(angles.size, dtype=np.complex128)
    for i, ang in enumerate(angles):
        aa = np.exp(-2j * np.pi * nn * d * float(sind(ang)))             # :82,:89
        p[i] = 1.0 / (np.vdot(aa, uann @ aa) + EPS)                       # :90
    pm = np.abs(p)                                                        # :94
    with np.errstate(divide="ignore"):
        return mag2db(pm / pm.max())                                      # :95-96


def music_doa(num_dets, rp, ra):
    """[L, aziEst, eleEst] = music(numDets, radarEstParams, Ra)  (music.m:1)."""
    ra = np.asarray(ra, dtype=np.complex128)
    n_ants = ra.shape[0]
    if num_dets is None:                                                  # :21-22
        va = np.real(linalg.eigvalsh(ra))
        n_sig = determine_num_targets(va)
    else:
        n_sig = int(num_dets)
    uann, _ = _noise_projector(ra, n_sig)
    arr = rp.antennaType
    if getattr(arr, "kind", "ula") == "upa":
        raise NotImplementedError("music.m:69 calls tools.find2DPeaks, which does not exist in the reference")
    pdb = music_spectrum_ula(uann, n_ants, rp)
    _, locs = findpeaks(pdb, npeaks=n_sig, sort_descend=True)             # :102 (L = 0 raises)
    azi = locs * rp.azimuthScanGranularity - rp.azimuthScanScale / 2.0    # :103  (azi-1)*g - aMax/2, locs 0-based
    ele = np.full(azi.shape, np.nan)                                      # :104
    return n_sig, azi.astype(np.float64), ele


def digital_bf(num_dets, rp, ra):
    """digitalBF.m:55-86 (ULA)."""
    ra = np.asarray(ra, dtype=np.complex128)
    n_ants = ra.shape[0]
    nn = np.arange(n_ants, dtype=np.float64)
    angles = ula_scan_angles(rp)
    p = np.array([np.vdot(aa, ra @ aa) for aa in
                  (np.exp(-2j * np.pi * nn * 0.5 * float(sind(a))) for a in angles)])   # :72
    pm = np.abs(p)
    pdb = mag2db(pm / pm.max())
    _, locs = findpeaks(pdb, npeaks=int(num_dets), sort_descend=True)     # :84
    azi = locs * rp.azimuthScanGranularity - rp.azimuthScanScale / 2.0
    return azi.astype(np.float64), np.full(azi.shape, np.nan)


def mvdr_bf(num_dets, rp, ra):
    """mvdrBF.m:55-86 (ULA):  1 / (a^H Ra^-1 a + eps)."""
    ra = np.asarray(ra, dtype=np.complex128)
    n_ants = ra.shape[0]
    nn = np.arange(n_ants, dtype=np.float64)
    ra_inv = np.linalg.inv(ra)
    angles = ula_scan_angles(rp)
    p = np.array([1.0 / (np.vdot(aa, ra_inv @ aa) + EPS) for aa in
                  (np.exp(-2j * np.pi * nn * 0.5 * float(sind(a))) for a in angles)])   # :72
    pm = np.abs(p)
    pdb = mag2db(pm / pm.max())
    _, locs = findpeaks(pdb, npeaks=int(num_dets), sort_descend=True)
    azi = locs * rp.azimuthScanGranularity - rp.azimuthScanScale / 2.0
    return azi.astype(np.float64), np.full(azi.shape, np.nan)


def music2d(rp, scs_khz, rx_grid, tx_grid, return_debug: bool = False):
    """music2D.m:1-123 -> {aziEst, eleEst, rngEst, velEst}."""
    n_sc, n_sym, n_ants = rx_grid.shape                                   # :33
    scs = scs_khz * 1e3                                                   # :34
    c = LIGHTSPEED
    lam = c / rp.fc                                                       # :37
    t_sym = rp.Tsri                                                       # :38
    r_max = float(rp.cfarEstZone[0, 1])                                   # :41
    v_max = float(rp.cfarEstZone[1, 1]) * 2.0                             # :42
    r_gran = v_gran = 0.5                                                 # :43-44
    r_steps = int(math.floor((r_max + 1) / r_gran))                       # :45
    v_steps = int(math.floor((v_max + 1) / v_gran))                       # :46

    g = rx_grid.reshape(n_sc * n_sym, n_ants, order="F")                  # :57
    ra = (g.conj().T @ g) / (n_sc * n_sym)                                # :58
    ra = 0.5 * (ra + ra.conj().T)
    n_sig, azi, ele = music_doa(None, rp, ra)                             # :61

    h = (rx_grid * np.conj(tx_grid))[:, :, 0]                             # :67-68
    rr = (h @ h.conj().T) / n_sym                                         # :71
    rv = (h.T @ np.conj(h)) / n_sc                                        # :72
    rr = 0.5 * (rr + rr.conj().T)
    rv = 0.5 * (rv + rv.conj().T)

    def noise_proj(r):
        w, u = linalg.eigh(r)
        u = u[:, np.argsort(-w, kind="stable")]
        un = u[:, n_sig:]
        return un                                                          # Urnn = un un^H (applied factored)

    urn = noise_proj(rr)                                                  # :77-82
    uvn = noise_proj(rv)                                                  # :84-89
    nn = np.arange(n_sc, dtype=np.float64)
    mm = np.arange(n_sym, dtype=np.float64)
    pr = np.empty(r_steps)
    for i in range(r_steps):                                              # :98-102
        ar = np.exp(-2j * np.pi * scs * 2 * (i * r_gran) * nn / c)        # :92
        y = urn.conj().T @ ar
        pr[i] = 1.0 / np.real(np.vdot(y, y))                              # a^H Urn Urn^H a
    pv = np.empty(v_steps)
    for i in range(v_steps):                                              # :104-108
        av = np.exp(2j * np.pi * t_sym * 2 * (i * v_gran - v_max / 2) * mm / lam)   # :93
        y = uvn.conj().T @ av
        pv[i] = 1.0 / np.real(np.vdot(y, y))
    pr_db = mag2db(np.abs(pr) / np.abs(pr).max())                         # :111-113
    pv_db = mag2db(np.abs(pv) / np.abs(pv).max())                         # :115-117
    _, rloc = findpeaks(pr_db, npeaks=n_sig, sort_descend=True)           # :120
    _, vloc = findpeaks(pv_db, npeaks=n_sig, sort_descend=True)           # :121
    est = SimpleNamespace(aziEst=azi, eleEst=ele,
                          rngEst=rloc * r_gran,                           # :122
                          velEst=vloc * v_gran - v_max / 2)               # :123
    if return_debug:
        return est, SimpleNamespace(L=n_sig, Ra=ra, Rr=rr, Rv=rv, PrdB=pr_db, PvdB=pv_db)
    return est
