"""ctypes binding of the C++/OpenMP CPU port (oracle/cpu_port/isac_cpu.cpp) -- TEST / BASELINE INFRASTRUCTURE ONLY.

Two uses, both outside the product path: (1) `bench.py`'s ``cpu_baseline`` leg (kind "port": the MATLAB reference
cannot run here or on the GPU box) and (2) a second independent implementation of the reference's chain that the NumPy
oracle is cross-checked against (tests/test_cpu_port.py).  Built by ``__graft_entry__.build()`` into
``oracle/_build/libisac_cpu.so`` (git-ignored, travels to the GPU box with the snapshot).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "isac_cpu.cpp")
LIB = os.path.join(os.path.dirname(HERE), "_build", "libisac_cpu.so")
_lib = None


def build(force: bool = False) -> str:
    """g++ -O3 -mavx2 -mfma -fcx-limited-range -fopenmp (x86-64-v3: runs on the build container and on the GPU box's host alike;
    complex products without the Inf/NaN recovery branches, like every FFT library)."""
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    cmd = ["g++", "-O3", "-mavx2", "-mfma", "-fcx-limited-range", "-fopenmp", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wextra", "-o", LIB, SRC]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("CPU port failed to compile:\n" + r.stderr)
    return LIB


class _Radar(C.Structure):
    _fields_ = [("fc", C.c_double), ("fs", C.c_double), ("n0", C.c_double), ("n_ants", C.c_int32), ("n_targets", C.c_int32),
                ("range", C.c_void_p), ("velocity", C.c_void_p), ("lsf", C.c_void_p), ("steering", C.c_void_p)]


class _Cfg(C.Structure):
    _fields_ = [("n_ifft", C.c_int32), ("n_fft", C.c_int32), ("r_res", C.c_double), ("v_res", C.c_double), ("pfa", C.c_double),
                ("guard", C.c_int32 * 2), ("train", C.c_int32 * 2), ("row0", C.c_int32), ("row1", C.c_int32), ("col0", C.c_int32),
                ("col1", C.c_int32), ("az_scale", C.c_double), ("az_gran", C.c_double)]


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.isac_cpu_threads.restype = C.c_int
        _lib.isac_cpu_mono_static_sensing.restype = C.c_int
        _lib.isac_cpu_fft2d.restype = C.c_int
        _lib.isac_cpu_symbol_count.restype = C.c_int
        _lib.isac_cpu_symbol_count.argtypes = [C.c_int, C.c_int, C.c_longlong]
    return _lib


def threads() -> int:
    return int(load().isac_cpu_threads())


def _f(a, dtype):
    return np.ascontiguousarray(np.asarray(a, dtype=dtype).reshape(-1))


def mono_static_sensing(tx_waveform, tx_dimension, carrier_info, rp, los, noise_unit=None, nfft=4096, seed=0, out=None):
    """Same contract as oracle.mono_static_sensing; ``noise_unit`` [T x A] injected, or ``seed`` != 0 for the port's own AWGN.
    ``out``: an echo grid to overwrite (a caller that processes CPI after CPI keeps its buffer: a fresh 0.75 GB array per call costs
    0.2 s of page faults on a 128-thread host)."""
    lib = load()
    tx = np.asfortranarray(np.asarray(tx_waveform, dtype=np.complex128))
    t_len, a = tx.shape
    k = int(carrier_info.NRBsDL) * 12
    rng_, vel_, lsf_ = _f(rp.range, np.float64), _f(rp.velocity, np.float64), _f(rp.largeScaleFading, np.float64)
    sv = np.asfortranarray(np.asarray(rp.RxSteeringVec, dtype=np.complex128))
    rad = _Radar(float(rp.fc), float(rp.fs), float(rp.N0), a, int(rp.nTargets), rng_.ctypes.data, vel_.ctypes.data, lsf_.ctypes.data, sv.ctypes.data)
    los = _f(np.asarray(los).reshape(-1) == 1, np.uint8)
    lw = lib.isac_cpu_symbol_count(int(nfft), int(carrier_info.SubcarrierSpacing), t_len)
    l_out = max(lw, int(tx_dimension[1]), 1)
    if out is None:
        out = np.empty((k, l_out, a), dtype=np.complex128, order="F")
    elif out.shape != (k, l_out, a) or out.dtype != np.complex128 or not out.flags.f_contiguous:
        raise ValueError("out must be a Fortran-ordered complex128 array of the echo grid's shape")
    nz = None if noise_unit is None else np.asfortranarray(np.asarray(noise_unit, dtype=np.complex128))
    lo = C.c_int(0)
    st = lib.isac_cpu_mono_static_sensing(tx.ctypes.data_as(C.c_void_p), C.c_longlong(t_len), C.c_int(int(tx_dimension[1])), C.c_int(k), C.c_int(int(nfft)),
                                          C.c_int(int(carrier_info.SubcarrierSpacing)), C.byref(rad), los.ctypes.data_as(C.c_void_p),
                                          nz.ctypes.data_as(C.c_void_p) if nz is not None else None, C.c_uint64(int(seed)),
                                          out.ctypes.data_as(C.c_void_p), C.byref(lo))
    if st != 0:
        raise ValueError(f"cpu_port.mono_static_sensing: status {st}")
    return out


def fft2d(rp, cfar, rx_grid, tx_grid, return_debug=False):
    """Same contract as oracle.fft2d (ValueError when nothing is detected)."""
    lib = load()
    rx = np.asfortranarray(np.asarray(rx_grid, dtype=np.complex128))
    tx = np.asfortranarray(np.asarray(tx_grid, dtype=np.complex128))
    k, l, a = rx.shape
    cut = np.asarray(cfar.CUTIdx)
    cfg = _Cfg(int(rp.nIFFT), int(rp.nFFT), float(rp.rRes), float(rp.vRes), float(cfar.Pfa), (C.c_int32 * 2)(*cfar.GuardBandSize),
               (C.c_int32 * 2)(*cfar.TrainingBandSize), int(cut[0].min()), int(cut[0].max()), int(cut[1].min()), int(cut[1].max()),
               float(rp.azimuthScanScale), float(rp.azimuthScanGranularity))
    cap = int(cut.shape[1]) * a
    det = np.zeros((2, max(cap, 1)), dtype=np.int32, order="F")
    off = np.zeros(a + 1, dtype=np.int32)
    ecap = 4096
    rng_, vel_, azi_ = np.zeros(ecap), np.zeros(ecap), np.zeros(ecap)
    nr_, nv_, na_ = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    ra = np.zeros((a, a), dtype=np.complex128, order="F")
    hr, hc = cfar.GuardBandSize[0] + cfar.TrainingBandSize[0], cfar.GuardBandSize[1] + cfar.TrainingBandSize[1]
    nr, nc = cfg.row1 - cfg.row0 + 1 + 2 * hr, cfg.col1 - cfg.col0 + 1 + 2 * hc
    pw = np.zeros((nr, nc, a), dtype=np.float64, order="F") if return_debug else None
    st = lib.isac_cpu_fft2d(rx.ctypes.data_as(C.c_void_p), tx.ctypes.data_as(C.c_void_p), C.c_int(k), C.c_int(l), C.c_int(a), C.byref(cfg),
                            det.ctypes.data_as(C.c_void_p), C.c_int(cap), off.ctypes.data_as(C.c_void_p), C.byref(nr_), rng_.ctypes.data_as(C.c_void_p),
                            C.byref(nv_), vel_.ctypes.data_as(C.c_void_p), C.byref(na_), azi_.ctypes.data_as(C.c_void_p), C.c_int(ecap),
                            ra.ctypes.data_as(C.c_void_p), pw.ctypes.data_as(C.c_void_p) if pw is not None else None)
    if st == 4:
        raise ValueError("findpeaks: NPeaks must be a positive integer (no CFAR detection)")
    if st != 0:
        raise ValueError(f"cpu_port.fft2d: status {st}")
    est = SimpleNamespace(rngEst=rng_[: nr_.value].copy(), velEst=vel_[: nv_.value].copy(), aziEst=azi_[: na_.value].copy(),
                          eleEst=np.full(na_.value, np.nan))
    if return_debug:
        dets = [det[:, off[i]:off[i + 1]].astype(np.int64) for i in range(a)]
        return est, SimpleNamespace(detections=dets, Ra=ra, power_window=pw, first_row=cfg.row0 - hr, first_col=cfg.col0 - hc)
    return est
