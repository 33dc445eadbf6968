"""Generates tests/golden/*.npz from the CPU ORACLE (the MATLAB reference cannot run here and
ships no vectors of its own -- see oracle/__init__.py, "parity unpinned").  Inputs are
re-derived from the seed by tests/conftest.make_scene; outputs are the oracle's.

    python tests/golden/make_golden.py [chain_small] [config1_a16] [config2_a64] [config4_a256]

chain_small   24 PRB / 8 antennas: full echo sub-sample + detection list (CPU suite re-checks it in seconds).
config1_a16   BASELINE.json configs[0] at its full size: 273 PRB, 224 symbols, the reference's default 16-element ULA (ula.m:45).
config2_a64   configs[1], the benchmark shape: 273 PRB, 224 symbols, 64 antennas, 2 targets, injected AWGN.
config4_a256  configs[3] array side: 256-element ULA at the full 273-PRB bandwidth (28 symbols to bound the oracle's memory).
The full-size fixtures hold SURVEY 8(c)'s "one full-size hash per config": SHA-256 of the per-antenna CFAR detection index
lists, plus the estimates, Ra, a strided echo-grid sub-sample and three |rdm|^2 planes -- everything a parity test needs
without re-running the oracle at 1 GB sizes.
"""
from __future__ import annotations

import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as O  # noqa: E402
from conftest import make_scene  # noqa: E402

FULL = {
    "config1_a16": dict(n_ants=16, n_slots=16, nrb=273, targets=((100.0, 20.0, 1.5),), velocity=(7.0,), seed=101),
    "config2_a64": dict(n_ants=64, n_slots=16, nrb=273, targets=((100.0, 20.0, 1.5), (180.0, -150.0, 1.5)), velocity=(7.0, -4.0), seed=102),
    "config4_a256": dict(n_ants=256, n_slots=2, nrb=273, targets=((150.0, 40.0, 1.5),), velocity=(3.0,), seed=104, num_slots_param=3,
                         zero_s_slots=False),
}
ECHO_STRIDE = (97, 13, 5)        # sub-sample of the echo grid kept in the fixture (subcarrier, symbol, antenna)


def detection_digest(dets) -> str:
    """SHA-256 over [n_ants, offsets..., row/col pairs...] as little-endian int32 (the per-antenna lists in CUT order)."""
    off = np.concatenate([[0], np.cumsum([d.shape[1] for d in dets])]).astype("<i4")
    flat = np.concatenate([d.astype("<i4").reshape(-1, order="F") for d in dets]) if len(dets) else np.zeros(0, "<i4")
    h = hashlib.sha256()
    h.update(np.array([len(dets)], "<i4").tobytes())
    h.update(off.tobytes())
    h.update(flat.tobytes())
    return h.hexdigest()


def estimate_digest(est) -> str:
    h = hashlib.sha256()
    for v in (est.rngEst, est.velEst, est.aziEst):
        h.update(np.asarray(v, "<f8").tobytes())
    return h.hexdigest()


def chain_small():
    kw = dict(n_ants=8, n_slots=4, nrb=24, targets=((150.0, 40.0, 1.5), (-90.0, 70.0, 5.0)), velocity=(0.0, 6.0),
              num_slots_param=6, seed=21)
    sc = make_scene(**kw)
    echo = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, sc.noise, nfft=sc.wave.Nfft)
    est, dbg = O.fft2d(sc.rp, O.cfar2d_config(sc.rp), echo, sc.tx_grid, return_debug=True)
    np.savez_compressed(os.path.join(HERE, "chain_small.npz"), n_ants=kw["n_ants"], n_slots=kw["n_slots"], nrb=kw["nrb"],
                        targets=np.array(kw["targets"]), velocity=np.array(kw["velocity"]), num_slots_param=kw["num_slots_param"],
                        seed=kw["seed"], tx_grid_sha256=hashlib.sha256(np.ascontiguousarray(sc.tx_grid).tobytes()).hexdigest(),
                        echo_grid_sub=echo[::5, ::3, :], rngEst=est.rngEst, velEst=est.velEst,
                        aziEst=est.aziEst, det_idx=np.concatenate(dbg.detections, axis=1), Ra=dbg.Ra)
    print("chain_small:", est)


def full_config(name):
    kw = FULL[name]
    t0 = time.perf_counter()
    sc = make_scene(**kw)
    t1 = time.perf_counter()
    echo = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, sc.noise, nfft=sc.wave.Nfft)
    t2 = time.perf_counter()
    cf = O.cfar2d_config(sc.rp)
    est, dbg = O.fft2d(sc.rp, cf, echo, sc.tx_grid, return_debug=True, rdm_fn=O.rdm_explicit)   # == rdm_literal bit for bit (KAT-4)
    t3 = time.perf_counter()
    # how far is the closest CUT from its threshold?  (a parity test is only meaningful when rounding cannot flip a detection)
    margin = np.inf
    planes = sorted({0, sc.A // 2, sc.A - 1})
    for a in range(sc.A):
        p = np.abs(dbg.rdm[:, :, a]) ** 2
        _, thr = O.ca_cfar2d(p, cf.CUTIdx, cf.Pfa, return_threshold=True)
        pc = p[cf.CUTIdx[0] - 1, cf.CUTIdx[1] - 1]
        margin = min(margin, float(np.min(np.abs(pc - thr) / np.maximum(np.abs(thr), 1e-300))))
    hr, hc = 3, 3
    r0, r1 = int(cf.CUTIdx[0].min()) - hr, int(cf.CUTIdx[0].max()) + hr
    c0, c1 = int(cf.CUTIdx[1].min()) - hc, int(cf.CUTIdx[1].max()) + hc
    pw = np.stack([np.abs(dbg.rdm[r0 - 1:r1, c0 - 1:c1, a]) ** 2 for a in planes], axis=2)
    s = ECHO_STRIDE
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), name=name, n_ants=kw["n_ants"], n_slots=kw["n_slots"], nrb=kw["nrb"],
        targets=np.array(kw["targets"]), velocity=np.array(kw["velocity"]), seed=kw["seed"],
        num_slots_param=kw.get("num_slots_param", -1), zero_s_slots=int(kw.get("zero_s_slots", True)),
        tx_grid_sha256=hashlib.sha256(np.ascontiguousarray(sc.tx_grid[:, :, 0]).tobytes()).hexdigest(),
        echo_stride=np.array(s), echo_grid_sub=echo[::s[0], ::s[1], ::s[2]], echo_max=float(np.abs(echo).max()),
        det_sha256=detection_digest(dbg.detections), det_counts=np.array([d.shape[1] for d in dbg.detections]),
        est_sha256=estimate_digest(est), rngEst=est.rngEst, velEst=est.velEst, aziEst=est.aziEst,
        Ra=dbg.Ra, pw_planes=np.array(planes), pw_first=np.array([r0, c0]), power_window=pw, cfar_margin=margin)
    print(f"{name}: scene {t1 - t0:.1f}s echo {t2 - t1:.1f}s fft2d {t3 - t2:.1f}s | dets/ant {np.array([d.shape[1] for d in dbg.detections])[:8]}... "
          f"rng {est.rngEst} vel {est.velEst} azi {est.aziEst} | closest CUT-to-threshold margin {margin:.2e}")


if __name__ == "__main__":
    which = sys.argv[1:] or ["chain_small", *FULL]
    for w in which:
        chain_small() if w == "chain_small" else full_config(w)
