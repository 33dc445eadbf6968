"""Seeded differential fuzz of the whole chain: random small scenes (numerology, antennas, slots, targets, Doppler, LoS
pattern, noise) through monoStaticSensing -> fft2D on the GPU against the oracle.  Same tolerances as test_gpu_parity.py:
fields <= 1e-10 relative, CFAR detections / range-velocity bins / integer azimuths exact.  Scenes in which some CUT lies
within 1e-9 (relative) of its CFAR threshold are skipped -- there a rounding-level difference may legitimately flip a
detection.  ISAC_FUZZ_N=<n> runs more seeds."""
from __future__ import annotations

import os

import numpy as np
import pytest

import oracle as O
from conftest import load_pkg, make_scene
from test_gpu_parity import RTOL, _guard_band_ok, rel

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("ISAC_FUZZ_N", "16"))


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


def _scene(seed):
    rng = np.random.default_rng(1000 + seed)
    nrb = int(rng.choice([24, 51, 106, 133, 273]))
    n_ants = int(rng.choice([1, 2, 3, 4, 5, 8]))
    n_slots = int(rng.choice([2, 3, 4, 6]))
    q = int(rng.integers(1, 4))
    r = rng.uniform(60.0, 400.0, q)
    az = np.deg2rad(rng.uniform(-70.0, 70.0, q))
    targets = tuple((float(r[i] * np.cos(az[i])), float(r[i] * np.sin(az[i])), 1.5) for i in range(q))
    vel = tuple(float(v) for v in rng.integers(-12, 13, q))
    sc = make_scene(n_ants=n_ants, n_slots=n_slots, nrb=nrb, targets=targets, velocity=vel, seed=seed,
                    zero_s_slots=bool(rng.integers(0, 2)), with_noise=bool(rng.integers(0, 4) > 0),
                    num_slots_param=int(rng.choice([n_slots, n_slots + 2, max(2, n_slots - 1)])))
    los = (rng.random(q) < 0.8).astype(np.uint8)
    if not los.any():
        los[int(rng.integers(0, q))] = 1
    return sc, los


@pytest.mark.parametrize("seed", range(N_CASES))
def test_chain_matches_oracle_on_random_scene(pkg, seed):
    sc, los = _scene(seed)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    echo = pkg.sensing.monoStaticSensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, rp, los, noise=sc.noise, nfft=sc.wave.Nfft)
    ref_echo = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, los, sc.noise, nfft=sc.wave.Nfft)
    assert rel(echo, ref_echo) < RTOL
    ocf = O.cfar2d_config(sc.rp)
    try:
        want, dbg = O.fft2d(sc.rp, ocf, ref_echo, sc.tx_grid, return_debug=True)
    except ValueError:                                   # zero detections: findpeaks(NPeaks = 0) errors in the reference
        with pytest.raises(pkg.IsacError) as ei:
            pkg.sensing.estimation.fft2D(rp, cf, ref_echo, sc.tx_grid)
        assert ei.value.name == "NO_DETECTION"
        return
    for a in range(sc.A):
        if not _guard_band_ok(np.abs(dbg.rdm[:, :, a]) ** 2, ocf.CUTIdx, ocf.Pfa):
            pytest.skip("a CUT sits within 1e-9 of its threshold")
    got, gd = pkg.sensing.estimation.fft2D(rp, cf, ref_echo, sc.tx_grid, return_debug=True)
    r0, c0 = gd.first_row - 1, gd.first_col - 1
    nr, nc, _ = gd.power_window.shape
    assert rel(gd.power_window, np.abs(dbg.rdm[r0:r0 + nr, c0:c0 + nc, :]) ** 2) < RTOL
    assert rel(gd.Ra, dbg.Ra) < RTOL
    for a in range(sc.A):
        assert np.array_equal(gd.detections[a], dbg.detections[a]), f"antenna {a}"
    assert np.array_equal(got.rngEst, want.rngEst) and np.array_equal(got.velEst, want.velEst)
    # MUSIC separates the L = numDets largest eigenvalues from the rest (music.m:21-25).  When that split falls inside a
    # numerically degenerate cluster (noise-free scene with more detections than sources: the "noise" eigenvalues are
    # rounding residue ~1e-18), WHICH eigenvectors land on the signal side is decided by rounding, the pseudo-spectrum is
    # not a function of Ra any more, and there is nothing to compare.
    w = np.sort(np.linalg.eigvalsh(dbg.Ra))[::-1]
    n_sig = min(int(want.rngEst.size), sc.A - 1)
    if n_sig >= 1 and (w[n_sig - 1] - w[n_sig]) < 1e-9 * w[0]:
        pytest.skip("signal/noise split inside a degenerate eigenvalue cluster: MUSIC peaks undefined")
    assert np.array_equal(got.aziEst, want.aziEst)
