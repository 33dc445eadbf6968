"""Oracle restatement of ``sensing.detection.cfar2D`` (+sensing/+detection/cfar2D.m:1-39)
and of the ``phased.CFARDetector2D`` step it configures (stepped at fft2D.m:62).

Toolbox behaviour: cell-averaging over the (2(G+T)+1)^2 block
minus the (2G+1)^2 guard block, alpha = N (Pfa^(-1/N) - 1), strict '>' test,
detections reported in CUTIdx column order.  The *summation order* inside the
toolbox is unobservable; the oracle DEFINES it as: column-major walk over the
outer block (column offset slowest, row offset fastest), skipping guard/CUT cells,
left-to-right fp64 accumulation -- the HIP kernel reproduces exactly this order so
that detection indices are bit-identical on identical power maps.
TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np


def cfar2d_config(rp) -> SimpleNamespace:
    """cfar2D.m:17-37.  Returns CUTIdx [2 x nCUT] (1-based, rows fastest) + detector config."""
    n_ifft, n_fft = int(rp.nIFFT), int(rp.nFFT)
    rng_grid = np.arange(n_ifft, dtype=np.float64) * rp.rRes                 # :17
    dop_grid = np.arange(-n_fft // 2, n_fft // 2, dtype=np.float64) * rp.vRes   # :18
    rng_detec = np.asarray(rp.cfarEstZone)[0]                                # :19
    dop_detec = np.asarray(rp.cfarEstZone)[1]                                # :20
    # [~, idx] = min(abs(grid - edge))  -> first minimiser, 1-based           :21-22
    rng_idx = [int(np.argmin(np.abs(rng_grid - e))) + 1 for e in rng_detec]
    dop_idx = [int(np.argmin(np.abs(dop_grid - e))) + 1 for e in dop_detec]
    rows = np.arange(rng_idx[0], rng_idx[1] + 1)
    cols = np.arange(dop_idx[0], dop_idx[1] + 1)
    col_idxs, row_idxs = np.meshgrid(cols, rows)                             # :23
    cut = np.stack([row_idxs.ravel(order="F"), col_idxs.ravel(order="F")]).astype(np.int64)   # :24
    return SimpleNamespace(CUTIdx=cut, Method="CA", Pfa=float(rp.Pfa),
                           GuardBandSize=(2, 2), TrainingBandSize=(1, 1),   # :32-33
                           rowRange=(rng_idx[0], rng_idx[1]), colRange=(dop_idx[0], dop_idx[1]))


def cfar_threshold_factor(n_train: int, pfa: float) -> float:
    """ThresholdFactor='Auto', CA:  alpha = N (Pfa^(-1/N) - 1)."""
    return n_train * (pfa ** (-1.0 / n_train) - 1.0)


def training_offsets(guard=(2, 2), train=(1, 1)):
    """Oracle-defined training-cell order: column offset slowest, row offset fastest."""
    gr, gc = guard
    tr, tc = train
    offs = []
    for dc in range(-(gc + tc), gc + tc + 1):
        for dr in range(-(gr + tr), gr + tr + 1):
            if abs(dr) <= gr and abs(dc) <= gc:
                continue
            offs.append((dr, dc))
    return offs


def ca_cfar2d(power: np.ndarray, cut_idx: np.ndarray, pfa: float,
              guard=(2, 2), train=(1, 1), return_threshold: bool = False):
    """``detections = cfarDetector(P, CUTIdx)`` with OutputFormat 'Detection index'.

    power [nRows x nCols] fp64; cut_idx [2 x nCUT] 1-based.  Returns [2 x D] 1-based
    indices in CUTIdx order (and optionally the per-CUT thresholds).
    """
    p = np.asarray(power, dtype=np.float64)
    cut = np.asarray(cut_idx, dtype=np.int64)
    r = cut[0] - 1
    c = cut[1] - 1
    offs = training_offsets(guard, train)
    hr = guard[0] + train[0]
    hc = guard[1] + train[1]
    if cut.shape[1] and (r.min() - hr < 0 or r.max() + hr >= p.shape[0]
                         or c.min() - hc < 0 or c.max() + hc >= p.shape[1]):
        raise ValueError("CFARDetector2D: CUT training window exceeds the input matrix")
    acc = np.zeros(cut.shape[1], dtype=np.float64)
    for dr, dc in offs:                       # fixed left-to-right accumulation order
        acc = acc + p[r + dr, c + dc]
    n = len(offs)
    noise = acc / n
    alpha = cfar_threshold_factor(n, pfa)
    thr = alpha * noise
    det = p[r, c] > thr
    out = cut[:, det]
    if return_threshold:
        return out, thr
    return out
