"""The C++/OpenMP CPU port (oracle/cpu_port: bench.py's `cpu_baseline`, kind "port") against the NumPy oracle: two
independent restatements of the same reference lines (monoStaticSensing.m:1-23, basicRadarChannel.m:21-74, fft2D.m:37-115,
music.m:19-104) must agree -- echo grid / |rdm|^2 / Ra to 1e-10, CFAR index lists and all estimates exactly."""
import numpy as np
import pytest

import oracle as O
from oracle import cpu_port as P
from conftest import make_scene


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


@pytest.mark.parametrize("kw", [
    dict(n_ants=4, n_slots=4, nrb=24, targets=((150.0, 40.0, 1.5),), velocity=(0.0,), num_slots_param=6, zero_s_slots=False),
    dict(n_ants=6, n_slots=8, nrb=51, targets=((120.0, 60.0, 1.5), (-250.0, 80.0, 1.5)), velocity=(10.0, -6.0), num_slots_param=12, seed=5),
    dict(n_ants=3, n_slots=2, nrb=133, targets=((120.0, 30.0, 1.5),), velocity=(3.0,), num_slots_param=3, zero_s_slots=False),
    dict(n_ants=8, n_slots=16, nrb=273, targets=((100.0, 20.0, 1.5), (180.0, -150.0, 1.5)), velocity=(7.0, -4.0), seed=11),
])
def test_cpu_port_matches_oracle(kw):
    sc = make_scene(**kw)
    los = sc.los
    want_echo = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, los, sc.noise, nfft=sc.wave.Nfft)
    got_echo = P.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, los, sc.noise, nfft=sc.wave.Nfft)
    assert got_echo.shape == want_echo.shape and rel(got_echo, want_echo) < 1e-10
    cf = O.cfar2d_config(sc.rp)
    want, odbg = O.fft2d(sc.rp, cf, want_echo, sc.tx_grid, return_debug=True, rdm_fn=O.rdm_explicit)
    got, gdbg = P.fft2d(sc.rp, cf, want_echo, sc.tx_grid, return_debug=True)
    r0, c0 = gdbg.first_row - 1, gdbg.first_col - 1
    nr, nc, _ = gdbg.power_window.shape
    assert rel(gdbg.power_window, np.abs(odbg.rdm[r0:r0 + nr, c0:c0 + nc, :]) ** 2) < 1e-10
    for a in range(sc.A):
        assert np.array_equal(gdbg.detections[a], odbg.detections[a]), f"antenna {a}"
    assert np.array_equal(got.rngEst, want.rngEst) and np.array_equal(got.velEst, want.velEst)
    assert np.array_equal(got.aziEst, want.aziEst) and got.rngEst.size >= 1
    assert rel(gdbg.Ra, odbg.Ra) < 1e-10 and np.array_equal(gdbg.Ra, gdbg.Ra.conj().T)


def test_cpu_port_edge_cases():
    sc = make_scene(n_ants=2, n_slots=1, nrb=24, with_noise=False)
    with pytest.raises(ValueError):
        P.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, np.zeros(1), None, nfft=sc.wave.Nfft)        # all NLoS
    with pytest.raises(ValueError):
        P.mono_static_sensing(sc.tx_wave[:100], sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, None, nfft=sc.wave.Nfft)       # < 1 symbol
    wave = sc.tx_wave[:-7]                                       # partial last symbol + padding (monoStaticSensing.m:19-21)
    got = P.mono_static_sensing(wave, (sc.K, 20, sc.A), sc.carrier, sc.rp, sc.los, None, nfft=sc.wave.Nfft)
    want = O.mono_static_sensing(wave, (sc.K, 20, sc.A), sc.carrier, sc.rp, sc.los, None, nfft=sc.wave.Nfft)
    assert got.shape == (sc.K, 20, sc.A) and rel(got, want) < 1e-10 and np.all(got[:, 13:, :] == 0)
    sc2 = make_scene(n_ants=2, n_slots=4, nrb=24, num_slots_param=6, with_noise=False)
    with pytest.raises(ValueError):
        P.fft2d(sc2.rp, O.cfar2d_config(sc2.rp), np.zeros_like(sc2.tx_grid), sc2.tx_grid)                                    # no detection
    # the port's own AWGN: unit variance after removing the clean echo
    clean = P.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, None, nfft=sc.wave.Nfft)
    noisy = P.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, None, nfft=sc.wave.Nfft, seed=7)
    nzv = (noisy - clean) / (np.sqrt(sc.rp.N0 / 2.0) * np.sqrt(sc.wave.Nfft))
    assert abs(nzv.real.std() - 1) < 0.03 and abs(nzv.imag.std() - 1) < 0.03 and abs(nzv.mean()) < 0.02
    assert P.threads() >= 1


@pytest.mark.parametrize("seed", range(8))
def test_cpu_port_matches_oracle_random_scenes(seed):
    """Seeded random scenes (array size, bandwidth, CPI length, target count / geometry / speed, S-slot pattern): the two restatements
    agree on every field, or fail the same way where the scene yields no detection (fft2D.m errors inside findpeaks there)."""
    rng = np.random.default_rng(4200 + seed)
    n_t = int(rng.integers(1, 4))
    r = rng.uniform(60.0, 400.0, n_t)
    az = np.deg2rad(rng.uniform(-70.0, 70.0, n_t))
    kw = dict(n_ants=int(rng.integers(1, 10)), n_slots=int(rng.integers(2, 9)), nrb=int(rng.choice([24, 51, 106])),
              targets=tuple((float(r[i] * np.cos(az[i])), float(r[i] * np.sin(az[i])), 1.5) for i in range(n_t)),
              velocity=tuple(float(v) for v in rng.integers(-12, 13, n_t)), seed=seed, zero_s_slots=bool(rng.integers(0, 2)))
    kw["num_slots_param"] = kw["n_slots"] + int(rng.integers(0, 3))
    sc = make_scene(**kw)
    want_echo = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, sc.noise, nfft=sc.wave.Nfft)
    got_echo = P.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, sc.noise, nfft=sc.wave.Nfft)
    assert rel(got_echo, want_echo) < 1e-10
    cf = O.cfar2d_config(sc.rp)
    try:
        want, odbg = O.fft2d(sc.rp, cf, want_echo, sc.tx_grid, return_debug=True, rdm_fn=O.rdm_explicit)
    except ValueError:
        with pytest.raises(ValueError):
            P.fft2d(sc.rp, cf, want_echo, sc.tx_grid)
        return
    got, gdbg = P.fft2d(sc.rp, cf, want_echo, sc.tx_grid, return_debug=True)
    for a in range(sc.A):
        assert np.array_equal(gdbg.detections[a], odbg.detections[a]), f"antenna {a}"
    assert np.array_equal(got.rngEst, want.rngEst) and np.array_equal(got.velEst, want.velEst) and np.array_equal(got.aziEst, want.aziEst)
    assert rel(gdbg.Ra, odbg.Ra) < 1e-10
