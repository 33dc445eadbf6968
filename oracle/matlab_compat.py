"""MATLAB base / toolbox helper semantics used by the sensing path (oracle only).

Restated from published MATLAB behaviour; the
toolbox sources are not under /root/reference, so each choice that could not be
observed is flagged "oracle-defined".
"""
from __future__ import annotations

import numpy as np
from scipy import special

# physconst('Lightspeed'), physconst('Boltzmann')  (radarParams.m:27,40; basicRadarChannel.m:11)
LIGHTSPEED = 299792458.0
BOLTZMANN = 1.380649e-23
EPS = float(np.finfo(np.float64).eps)  # eps(1), music.m:56,90


def db2pow(x):
    return 10.0 ** (np.asarray(x, dtype=np.float64) / 10.0)


def db2mag(x):
    return 10.0 ** (np.asarray(x, dtype=np.float64) / 20.0)


def pow2db(x):
    return 10.0 * np.log10(np.asarray(x, dtype=np.float64))


def mag2db(x):
    return 20.0 * np.log10(np.asarray(x, dtype=np.float64))


def _fold_deg(x):
    """Reduce degrees to [-90, 90] keeping sin() unchanged; exact for fp-representable angles."""
    x = np.fmod(np.asarray(x, dtype=np.float64), 360.0)          # (-360, 360)
    x = np.where(x > 180.0, x - 360.0, x)
    x = np.where(x < -180.0, x + 360.0, x)                        # [-180, 180]
    x = np.where(x > 90.0, 180.0 - x, x)
    x = np.where(x < -90.0, -180.0 - x, x)                        # [-90, 90]
    return x


def sind(x):
    """sin of an angle in degrees, exact at multiples of 90 and mirror-symmetric:
    sind(180-p) == sind(p) bit-for-bit (oracle-defined reduction).

    Used at radarParams.m:95,109 and music.m:44,82.
    """
    x = _fold_deg(x)
    ax = np.abs(x)
    small = np.sin(np.deg2rad(x))
    big = np.sign(x) * np.cos(np.deg2rad(90.0 - ax))
    return np.where(ax <= 45.0, small, big)


def cosd(x):
    """cos of an angle in degrees via sind(90 - |x|) (exact zeros at +-90)."""
    x = np.fmod(np.abs(np.asarray(x, dtype=np.float64)), 360.0)   # [0, 360)
    return sind(90.0 - x)


def kaiser(n: int, beta: float) -> np.ndarray:
    """Signal Processing Toolbox ``kaiser(n, beta)`` (fft2D.m:135 uses beta = 3).

    Half-window evaluation mirrored about the centre, so the result is exactly
    symmetric (the classic kaiser.m construction).
    """
    nw = int(round(n))
    if nw == 1:
        return np.ones(1)
    bes = abs(float(beta))
    odd = nw % 2
    xind = float((nw - 1) ** 2)
    half = (nw + 1) // 2
    xi = np.arange(half, dtype=np.float64) + 0.5 * (1 - odd)
    xi = 4.0 * xi * xi
    w = special.i0(bes * np.sqrt(1.0 - xi / xind)) / special.i0(bes)
    w = np.abs(np.concatenate([w[::-1][: half - odd], w]))
    return w


def findpeaks(y, npeaks=None, sort_descend=True):
    """Signal Processing Toolbox ``findpeaks(y,'NPeaks',L,'SortStr','descend')``
    (music.m:102; music2D.m:120-121).  Returns (pks, locs) with 0-based locs.

    Strict local maxima; a flat peak reports the first sample of the plateau;
    end points are never peaks; descending sort is stable (ties keep index
    order); NPeaks <= 0 raises like MATLAB's input validation does.
    """
    y = np.asarray(y, dtype=np.float64).ravel()
    if npeaks is not None and (int(npeaks) != npeaks or npeaks < 1):
        raise ValueError("findpeaks: NPeaks must be a positive integer")
    n = y.size
    if n < 3:
        return np.zeros(0), np.zeros(0, dtype=np.int64)
    # collapse runs of equal values onto their first sample
    keep = np.concatenate([[True], y[1:] != y[:-1]])
    idx = np.flatnonzero(keep)
    v = y[idx]
    if v.size < 3:
        return np.zeros(0), np.zeros(0, dtype=np.int64)
    s = np.sign(np.diff(v))
    # NaNs give sign NaN and never satisfy the comparison
    with np.errstate(invalid="ignore"):
        imax = 1 + np.flatnonzero(np.diff(s) < 0)
    locs = idx[imax]
    # a run that reaches the last sample has no falling edge -> already excluded
    pks = y[locs]
    if sort_descend:
        order = np.argsort(-pks, kind="stable")
        pks, locs = pks[order], locs[order]
    if npeaks is not None:
        pks, locs = pks[: int(npeaks)], locs[: int(npeaks)]
    return pks, locs.astype(np.int64)


def unique_stable(x):
    """``unique(x,'stable')`` (fft2D.m:99): first occurrences, original order."""
    x = np.asarray(x).ravel()
    _, first = np.unique(x, return_index=True)
    return x[np.sort(first)]
