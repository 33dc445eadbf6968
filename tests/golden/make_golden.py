"""Generates tests/golden/*.npz from the CPU ORACLE (the MATLAB reference cannot run here and
ships no vectors of its own -- see oracle/__init__.py, "parity unpinned").  Inputs are
re-derived from the seed by tests/conftest.make_scene; outputs are the oracle's.

    python tests/golden/make_golden.py [chain_small] [config1_a16] [config2_a64] [config4_a256] [config4_a256_full] [config3_7cells] [music2d_k3276]

chain_small   24 PRB / 8 antennas: full echo sub-sample + detection list (CPU suite re-checks it in seconds).
config1_a16   BASELINE.json configs[0] at its full size: 273 PRB, 224 symbols, the reference's default 16-element ULA (ula.m:45).
config2_a64   configs[1], the benchmark shape: 273 PRB, 224 symbols, 64 antennas, 2 targets, injected AWGN.
config4_a256  configs[3] array side: 256-element ULA at the full 273-PRB bandwidth (28 symbols to bound the oracle's memory).
config4_a256_full  configs[3] at its stated size: 256-element ULA, 273 PRB, 224 symbols (X = 256 x 733 824); the oracle runs the
              range-Doppler / CFAR stage antenna plane by antenna plane, so only hashes, Ra and three |rdm|^2 planes are ever held.
config3_7cells     configs[2]: seven cells, each = config 2 (A = 64, 16 slots), own targets / data / AWGN per cell; the AWGN is a seeded
              field on the DEMODULATED grid (the bench path's noise domain) mapped to the equivalent time-domain noise for the oracle
              (conftest.spectral_to_time_noise).  Holds every cell's estimates + the SHA-256 of its per-antenna CFAR lists.
music2d_k3276      music2D.m:67-123 at the numerology north_star names: K = 3276, L = 224 (Rr is 3276 x 3276), A = 16, two targets.
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
FULL["config4_a256_full"] = dict(n_ants=256, n_slots=16, nrb=273, targets=((150.0, 40.0, 1.5), (90.0, -120.0, 1.5)), velocity=(3.0, -8.0), seed=105)
# configs[2]: 7 cells, each = config 2.  Per-cell targets (x, y, z) / velocities; data seed 300 + c, spectral AWGN seed 400 + c.
CELLS7 = [
    dict(targets=((100.0, 20.0, 1.5),), velocity=(7.0,)),
    dict(targets=((180.0, -150.0, 1.5),), velocity=(-4.0,)),
    dict(targets=((60.0, 70.0, 1.5), (250.0, 60.0, 1.5)), velocity=(2.0, -9.0)),
    dict(targets=((-40.0, 120.0, 1.5),), velocity=(10.0,)),
    dict(targets=((300.0, -90.0, 1.5),), velocity=(-1.0,)),
    dict(targets=((140.0, 140.0, 1.5), (75.0, -30.0, 1.5)), velocity=(5.0, 0.0)),
    dict(targets=((220.0, 10.0, 1.5),), velocity=(-6.0,)),
]
MUSIC2D = dict(n_ants=16, n_slots=16, nrb=273, targets=((120.0, 60.0, 1.5), (-250.0, 80.0, 1.5)), velocity=(10.0, -6.0), seed=106)
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


def cell7_kwargs(c):
    return dict(n_ants=64, n_slots=16, nrb=273, seed=300 + c, with_noise=False, **CELLS7[c])


def cell7_spectral_noise(c, k, l, a):
    """Unit AWGN field on the demodulated grid of cell c [K x L x A] (drawn plane by plane: bounds the temporaries)."""
    rng = np.random.default_rng(400 + c)
    w = np.empty((k, l, a), dtype=np.complex128, order="F")
    for i in range(a):
        w[:, :, i] = rng.standard_normal((k, l)) + 1j * rng.standard_normal((k, l))
    return w


def streamed_fft2d(sc, echo, keep_planes):
    """O.fft2d with the range-Doppler / CFAR stage run one antenna plane at a time (fft2D.m:59-96 is a per-antenna loop; every FFT
    line is the same 1-D transform whatever the batch): returns est, detections, Ra, kept |rdm|^2 windows, CUT-threshold margin."""
    from oracle.fft2d import detect_per_antenna
    cf = O.cfar2d_config(sc.rp)
    n_ifft, n_fft = int(sc.rp.nIFFT), int(sc.rp.nFFT)
    hr, hc = 3, 3
    r0, r1 = int(cf.CUTIdx[0].min()) - hr, int(cf.CUTIdx[0].max()) + hr
    c0, c1 = int(cf.CUTIdx[1].min()) - hc, int(cf.CUTIdx[1].max()) + hc
    dets, rng_l, vel_l, pw, margin = [], [], [], {}, np.inf
    for a in range(sc.A):
        rdm = O.rdm_explicit(echo[:, :, a:a + 1], sc.tx_grid[:, :, a:a + 1], n_ifft, n_fft)
        d, r, v = detect_per_antenna(rdm, cf, sc.rp.rRes, sc.rp.vRes, n_fft)
        dets.append(d[0]); rng_l.append(r); vel_l.append(v)
        p = np.abs(rdm[:, :, 0]) ** 2
        _, thr = O.ca_cfar2d(p, cf.CUTIdx, cf.Pfa, return_threshold=True)
        pc = p[cf.CUTIdx[0] - 1, cf.CUTIdx[1] - 1]
        margin = min(margin, float(np.min(np.abs(pc - thr) / np.maximum(np.abs(thr), 1e-300))))
        if a in keep_planes:
            pw[a] = p[r0 - 1:r1, c0 - 1:c1].copy()
    from types import SimpleNamespace
    est = SimpleNamespace(rngEst=O.unique_stable(np.concatenate(rng_l)), velEst=O.unique_stable(np.concatenate(vel_l)))
    ra = O.covariance(echo)
    _, est.aziEst, est.eleEst = O.music_doa(est.rngEst.size, sc.rp, ra)
    return est, dets, ra, pw, margin, (r0, c0)


def config4_full(name="config4_a256_full"):
    kw = FULL[name]
    t0 = time.perf_counter()
    sc = make_scene(**kw)
    echo = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, sc.noise, nfft=sc.wave.Nfft)
    sc.noise = None; sc.tx_wave = None
    t1 = time.perf_counter()
    planes = sorted({0, sc.A // 2, sc.A - 1})
    est, dets, ra, pw, margin, (r0, c0) = streamed_fft2d(sc, echo, planes)
    s = ECHO_STRIDE
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), name=name, n_ants=kw["n_ants"], n_slots=kw["n_slots"], nrb=kw["nrb"],
        targets=np.array(kw["targets"]), velocity=np.array(kw["velocity"]), seed=kw["seed"], num_slots_param=-1, zero_s_slots=1,
        tx_grid_sha256=hashlib.sha256(np.ascontiguousarray(sc.tx_grid[:, :, 0]).tobytes()).hexdigest(),
        echo_stride=np.array(s), echo_grid_sub=echo[::s[0], ::s[1], ::s[2]], echo_max=float(np.abs(echo).max()),
        det_sha256=detection_digest(dets), det_counts=np.array([d.shape[1] for d in dets]),
        est_sha256=estimate_digest(est), rngEst=est.rngEst, velEst=est.velEst, aziEst=est.aziEst,
        Ra=ra, pw_planes=np.array(planes), pw_first=np.array([r0, c0]), power_window=np.stack([pw[a] for a in planes], axis=2), cfar_margin=margin)
    print(f"{name}: scene+echo {t1 - t0:.1f}s fft2d {time.perf_counter() - t1:.1f}s | dets/ant {np.array([d.shape[1] for d in dets])[:8]}... "
          f"rng {est.rngEst} vel {est.velEst} azi {est.aziEst} | margin {margin:.2e}")


def config3_7cells():
    from conftest import spectral_to_time_noise
    out = {}
    for c in range(len(CELLS7)):
        t0 = time.perf_counter()
        sc = make_scene(**cell7_kwargs(c))
        w = cell7_spectral_noise(c, sc.K, sc.L, sc.A)
        tnoise = spectral_to_time_noise(w, sc.T, sc.wave.Nfft, 30, sc.rp.fc, sc.rp.fs)
        del w
        echo = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, tnoise, nfft=sc.wave.Nfft)
        del tnoise
        est, dets, ra, _, margin, _ = streamed_fft2d(sc, echo, ())
        out[f"c{c}_rngEst"], out[f"c{c}_velEst"], out[f"c{c}_aziEst"] = est.rngEst, est.velEst, est.aziEst
        out[f"c{c}_det_sha256"] = detection_digest(dets)
        out[f"c{c}_n_det"] = sum(d.shape[1] for d in dets)
        out[f"c{c}_margin"] = margin
        out[f"c{c}_tx_grid_sha256"] = hashlib.sha256(np.ascontiguousarray(sc.tx_grid[:, :, 0]).tobytes()).hexdigest()
        out[f"c{c}_ra_trace"] = float(np.trace(ra).real)
        print(f"config3 cell {c}: {time.perf_counter() - t0:.1f}s rng {est.rngEst} vel {est.velEst} azi {est.aziEst} dets {out[f'c{c}_n_det']} margin {margin:.2e}")
    np.savez_compressed(os.path.join(HERE, "config3_7cells.npz"), n_cells=len(CELLS7), **out)


def music2d_full():
    kw = MUSIC2D
    t0 = time.perf_counter()
    sc = make_scene(**kw)
    echo = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, sc.noise, nfft=sc.wave.Nfft)
    est, dbg = O.music2d(sc.rp, 30, echo, sc.tx_grid, return_debug=True)
    s = ECHO_STRIDE
    np.savez_compressed(os.path.join(HERE, "music2d_k3276.npz"), L=dbg.L, aziEst=est.aziEst, rngEst=est.rngEst, velEst=est.velEst,
                        PrdB=dbg.PrdB, PvdB=dbg.PvdB, echo_grid_sub=echo[::s[0], ::s[1], ::s[2]], echo_max=float(np.abs(echo).max()),
                        tx_grid_sha256=hashlib.sha256(np.ascontiguousarray(sc.tx_grid[:, :, 0]).tobytes()).hexdigest())
    print(f"music2d_k3276: {time.perf_counter() - t0:.1f}s L {dbg.L} azi {est.aziEst} rng {est.rngEst} vel {est.velEst}")


SPECIAL = {"chain_small": lambda: chain_small(), "config4_a256_full": lambda: config4_full(), "config3_7cells": lambda: config3_7cells(),
           "music2d_k3276": lambda: music2d_full()}

if __name__ == "__main__":
    which = sys.argv[1:] or ["chain_small", *FULL, "config3_7cells", "music2d_k3276"]
    for w in which:
        SPECIAL[w]() if w in SPECIAL else full_config(w)
