"""sensing.detection.cfar2D (+sensing/+detection/cfar2D.m:1-39) and the phased.CFARDetector2D
step it configures."""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np

from ... import _lib as L


class CFARDetector2D:
    """The subset of phased.CFARDetector2D the reference uses (cfar2D.m:27-33): Method 'CA',
    ThresholdFactor 'Auto', OutputFormat 'Detection index'.  Calling the object runs the detector
    on the GPU: ``detections = cfarDetector2D(P, CUTIdx)`` -> [2 x D] 1-based indices in CUT order."""

    def __init__(self, ProbabilityFalseAlarm=1e-9, GuardBandSize=(2, 2), TrainingBandSize=(1, 1)):
        self.Method = "CA"
        self.ThresholdFactor = "Auto"
        self.ProbabilityFalseAlarm = float(ProbabilityFalseAlarm)
        self.OutputFormat = "Detection index"
        self.GuardBandSize = tuple(int(g) for g in GuardBandSize)
        self.TrainingBandSize = tuple(int(t) for t in TrainingBandSize)

    def __call__(self, P, CUTIdx, ctx=None):
        ctx = ctx or L.default_context()
        p = np.asfortranarray(np.asarray(P, dtype=np.float64))
        cut = np.asfortranarray(np.asarray(CUTIdx, dtype=np.int32))       # [2 x nCUT] column-major == interleaved (row, col)
        n_cut = cut.shape[1] if cut.ndim == 2 else 0
        det = np.zeros((2, max(n_cut, 1)), dtype=np.int32, order="F")
        n_det = C.c_int32(0)
        g = (C.c_int32 * 2)(*self.GuardBandSize)
        t = (C.c_int32 * 2)(*self.TrainingBandSize)
        ctx.check(ctx.lib.isac_cfar2d_ca(ctx.handle, p.ctypes.data_as(C.c_void_p), C.c_int32(p.shape[0]), C.c_int32(p.shape[1]),
                                         cut.ctypes.data_as(C.c_void_p), C.c_int32(n_cut), g, t,
                                         C.c_double(self.ProbabilityFalseAlarm), det.ctypes.data_as(C.c_void_p),
                                         C.c_int32(max(n_cut, 1)), C.byref(n_det)))
        return det[:, : n_det.value].astype(np.int64)


def cfar2D(radaParams):
    """cfarConfig = sensing.detection.cfar2D(radarParams): CUT index list of the detection zone
    (cfar2D.m:17-24, rows fastest) and the configured detector (cfar2D.m:27-33)."""
    nIFFT, nFFT = int(radaParams.nIFFT), int(radaParams.nFFT)
    rngGrid = np.arange(nIFFT, dtype=np.float64) * radaParams.rRes                 # :17
    dopGrid = np.arange(-nFFT // 2, nFFT // 2, dtype=np.float64) * radaParams.vRes  # :18
    zone = np.asarray(radaParams.cfarEstZone, dtype=np.float64)
    rngIdx = [int(np.argmin(np.abs(rngGrid - e))) + 1 for e in zone[0]]            # :21  first minimiser, 1-based
    dopIdx = [int(np.argmin(np.abs(dopGrid - e))) + 1 for e in zone[1]]            # :22
    columnIdxs, rowIdxs = np.meshgrid(np.arange(dopIdx[0], dopIdx[1] + 1), np.arange(rngIdx[0], rngIdx[1] + 1))   # :23
    CUTIdx = np.stack([rowIdxs.ravel(order="F"), columnIdxs.ravel(order="F")]).astype(np.int64)               # :24
    det = CFARDetector2D(radaParams.Pfa, (2, 2), (1, 1))                            # :27-33
    return SimpleNamespace(CUTIdx=CUTIdx, cfarDetector2D=det)
