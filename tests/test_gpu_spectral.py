"""Spectral noise modes of monoStaticSensing (ISAC_NOISE_INJECTED_SPECTRAL / ISAC_NOISE_PHILOX_SPECTRAL) and the fused
synthesis + range kernel, against the oracle's TIME-DOMAIN path: a unit noise field W on the demodulated grid is mapped
back to the time-domain noise that produces exactly W after the oracle's carrier rotation and OFDM demodulation
(conftest.spectral_to_time_noise), so `HIP(spectral W)` and `oracle(time noise)` must agree to the usual 1e-10 --
the linearity argument of include/isac.h (isac_noise_mode) checked numerically, not assumed."""
from __future__ import annotations

import numpy as np
import pytest

import oracle as O
from conftest import load_pkg, make_scene, spectral_to_time_noise

pytestmark = pytest.mark.gpu
RTOL = 1e-10


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


@pytest.fixture(scope="module")
def ctx(pkg):
    return pkg.default_context()


def unit_noise(shape, seed):
    rng = np.random.default_rng(seed)
    return np.asfortranarray(rng.standard_normal(shape) + 1j * rng.standard_normal(shape))


@pytest.mark.parametrize("nrb,n_ants,n_slots,targets,vel,los", [
    (24, 4, 2, ((150.0, 40.0, 1.5),), (0.0,), (1,)),
    (24, 5, 2, ((150.0, 40.0, 1.5), (-80.0, 60.0, 10.0), (300.0, -20.0, 1.5)), (4.0, -9.0, 2.0), (1, 0, 1)),
    (51, 3, 1, tuple((60.0 + 45.0 * i, 20.0 * (-1) ** i, 1.5) for i in range(6)), tuple(float(v) for v in (3, -4, 5, -6, 7, -8)), (1,) * 6),
    (273, 2, 2, ((100.0, 20.0, 1.5), (260.0, -200.0, 1.5)), (7.0, -10.0), (1, 1)),
    (133, 2, 2, ((120.0, 30.0, 1.5),), (3.0,), (1,)),
])
def test_injected_spectral_noise_matches_time_domain_oracle(pkg, ctx, nrb, n_ants, n_slots, targets, vel, los):
    sc = make_scene(n_ants=n_ants, n_slots=n_slots, nrb=nrb, targets=targets, velocity=vel, with_noise=False)
    los = np.array(los)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    w = unit_noise(sc.tx_grid.shape, nrb + n_ants)
    tnoise = spectral_to_time_noise(w, sc.T, sc.wave.Nfft, 30, sc.rp.fc, sc.rp.fs)
    want = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, los, tnoise, nfft=sc.wave.Nfft)
    got = pkg.sensing.monoStaticSensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, rp, los, spectral_noise=w, nfft=sc.wave.Nfft)
    assert got.shape == want.shape and rel(got, want) < RTOL
    # noiseless: the rank-Q factorisation alone (sum_q a_q D_q) against the per-antenna demodulation
    want0 = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, los, None, nfft=sc.wave.Nfft)
    got0 = pkg.sensing.monoStaticSensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, rp, los, spectral_noise=np.zeros_like(w), nfft=sc.wave.Nfft)
    assert rel(got0, want0) < RTOL
    # ... and against the library's own time-domain kernels with the equivalent injected noise
    got_t = pkg.sensing.monoStaticSensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, rp, los, noise=tnoise, nfft=sc.wave.Nfft)
    assert rel(got, got_t) < RTOL


def test_padding_and_partial_symbol_spectral(pkg, ctx):
    sc = make_scene(n_ants=2, n_slots=1, nrb=24, with_noise=False)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    wave = sc.tx_wave[:-7]                                   # 13 whole symbols, padded to 20 (monoStaticSensing.m:19-21)
    w = unit_noise((sc.K, 20, sc.A), 3)
    got = pkg.sensing.monoStaticSensing(wave, (sc.K, 20, sc.A), sc.carrier, rp, sc.los, spectral_noise=w, nfft=sc.wave.Nfft)
    tnoise = spectral_to_time_noise(w[:, :13, :], wave.shape[0], sc.wave.Nfft, 30, sc.rp.fc, sc.rp.fs)
    want = O.mono_static_sensing(wave, (sc.K, 20, sc.A), sc.carrier, sc.rp, sc.los, tnoise, nfft=sc.wave.Nfft)
    assert got.shape == (sc.K, 20, sc.A) and rel(got, want) < RTOL and np.all(got[:, 13:, :] == 0)


@pytest.mark.parametrize("nrb,n_ants,n_slots", [(24, 3, 2), (273, 2, 1)])
def test_philox_spectral_matches_restated_generator(pkg, ctx, nrb, n_ants, n_slots):
    sc = make_scene(n_ants=n_ants, n_slots=n_slots, nrb=nrb, with_noise=False)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    seed = 0x5EED0002ABCD
    w = O.philox_spectral_noise(sc.K, sc.L, sc.A, seed)
    got = pkg.sensing.monoStaticSensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, seed=seed, noise_domain="spectral", nfft=sc.wave.Nfft)
    inj = pkg.sensing.monoStaticSensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, spectral_noise=w, nfft=sc.wave.Nfft)
    # the generator's Box-Muller runs on the float32 hardware transcendentals (csrc/echo_dev.hpp): device field and restated field agree to a
    # float32 bound on the unit samples (O.philox.SPECTRAL_NOISE_ATOL), not bit for bit -- the injected modes above / below are the exact ones
    sig = np.sqrt(sc.rp.N0 / 2.0) * np.sqrt(sc.wave.Nfft)
    atol = sig * O.philox.SPECTRAL_NOISE_ATOL
    assert np.abs(got - inj).max() < atol
    want = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los,
                                 spectral_to_time_noise(w, sc.T, sc.wave.Nfft, 30, sc.rp.fc, sc.rp.fs), nfft=sc.wave.Nfft)
    assert np.abs(got - want).max() < atol + RTOL * np.abs(want).max()
    # the noise the device added, recovered: unit variance, matches the restatement element by element
    clean = pkg.sensing.monoStaticSensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=sc.wave.Nfft)
    nz = (got - clean) / sig
    assert np.abs(nz - w).max() < O.philox.SPECTRAL_NOISE_ATOL and abs(nz.real.std() - 1) < 0.02 and abs(nz.imag.std() - 1) < 0.02
    assert np.abs(nz).max() <= O.philox.SPECTRAL_NOISE_MAX + 1e-5
    # another seed gives another field; basicRadarChannel (time-domain output) rejects the spectral modes
    got2 = pkg.sensing.monoStaticSensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, seed=seed + 1, noise_domain="spectral", nfft=sc.wave.Nfft)
    assert not np.array_equal(got2, got)


def test_fused_spectral_path_is_identical_and_reuse_is_explicit(pkg, ctx):
    """Full numerology (Nfft == nIFFT == 4096): echo_range_kernel (synthesis + range stage) + fft2D(reuse_range=True) must give
    bit-identical echo grid, |rdm|^2 window and detections to the unfused sequence, for injected and Philox spectral noise."""
    sc = make_scene(n_ants=3, n_slots=4, nrb=273, targets=((100.0, 20.0, 1.5), (180.0, -150.0, 1.5)), velocity=(7.0, -4.0), seed=9, with_noise=False)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    d_wave, d_txg = ctx.to_device(sc.tx_wave), ctx.to_device(sc.tx_grid)
    w = unit_noise(sc.tx_grid.shape, 77)
    for kw in (dict(spectral_noise=ctx.to_device(w)), dict(seed=1234, noise_domain="spectral")):
        e0 = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, **kw)
        est0, dbg0 = pkg.sensing.estimation.fft2D(rp, cf, e0, d_txg, return_debug=True)
        e1 = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, fuse_fft2d=(rp, cf, d_txg), **kw)
        est1, dbg1 = pkg.sensing.estimation.fft2D(rp, cf, e1, d_txg, return_debug=True, reuse_range=True)
        assert np.array_equal(e0.numpy(), e1.numpy())
        assert np.array_equal(dbg0.power_window, dbg1.power_window)
        assert all(np.array_equal(a, b) for a, b in zip(dbg0.detections, dbg1.detections))
        assert np.array_equal(est0.rngEst, est1.rngEst) and np.array_equal(est0.velEst, est1.velEst) and np.array_equal(est0.aziEst, est1.aziEst)
        assert est0.rngEst.size >= 1
    # injected case against the oracle end to end (detections exact)
    tnoise = spectral_to_time_noise(w, sc.T, 4096, 30, sc.rp.fc, sc.rp.fs)
    want_echo = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, tnoise, nfft=4096)
    e1 = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, fuse_fft2d=(rp, cf, d_txg), spectral_noise=ctx.to_device(w))
    est1, dbg1 = pkg.sensing.estimation.fft2D(rp, cf, e1, d_txg, return_debug=True, reuse_range=True)
    assert rel(e1.numpy(), want_echo) < RTOL
    want, odbg = O.fft2d(sc.rp, O.cfar2d_config(sc.rp), want_echo, sc.tx_grid, return_debug=True, rdm_fn=O.rdm_explicit)
    for a in range(sc.A):
        assert np.array_equal(dbg1.detections[a], odbg.detections[a])
    assert np.array_equal(est1.rngEst, want.rngEst) and np.array_equal(est1.velEst, want.velEst) and np.array_equal(est1.aziEst, want.aziEst)
    # reuse is explicit: a plain fft2D ignores the cache; reuse_range without a matching cache is an error, not stale rows
    e2 = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, fuse_fft2d=(rp, cf, d_txg), seed=5, noise_domain="spectral")
    other = ctx.to_device(np.asfortranarray(2.0 * e2.numpy()))
    with pytest.raises(pkg.IsacError) as ei:
        pkg.sensing.estimation.fft2D(rp, cf, other, d_txg, reuse_range=True)
    assert ei.value.name == "INVALID_ARG"
    with pytest.raises(pkg.IsacError):
        pkg.sensing.estimation.fft2D(rp, cf, e2, d_txg, reuse_range=True)   # single use: the failed attempt consumed it
    _, dbg3 = pkg.sensing.estimation.fft2D(rp, cf, other, d_txg, return_debug=True)
    _, dbg2 = pkg.sensing.estimation.fft2D(rp, cf, e2, d_txg, return_debug=True)
    assert np.allclose(dbg3.power_window, 4.0 * dbg2.power_window, rtol=1e-12)
    # padded symbol dimension through the fused spectral path
    e3 = pkg.sensing.monoStaticSensing(ctx.to_device(sc.tx_wave[:-9]), sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, fuse_fft2d=(rp, cf, d_txg), seed=5, noise_domain="spectral")
    e4 = pkg.sensing.monoStaticSensing(ctx.to_device(sc.tx_wave[:-9]), sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, seed=5, noise_domain="spectral")
    assert np.array_equal(e3.numpy(), e4.numpy()) and np.all(e3.numpy()[:, -1, :] == 0)


def test_basic_radar_channel_rejects_spectral_modes(pkg, ctx):
    import ctypes as C
    sc = make_scene(n_ants=2, n_slots=1, nrb=24, with_noise=False)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    from importlib import import_module
    mm = import_module(pkg.__name__ + ".sensing._marshal")
    cb = mm.ChannelBlock(rp)
    d_w, d_o = ctx.to_device(sc.tx_wave), ctx.empty((sc.T, sc.A))
    los = np.ones(1, dtype=np.uint8)
    st = ctx.lib.isac_basic_radar_channel_dev(ctx.handle, C.c_void_p(d_w.ptr), C.c_int64(sc.T), C.byref(cb.block), los.ctypes.data_as(C.c_void_p),
                                              C.c_int(3), C.c_void_p(0), C.c_uint64(1), C.c_void_p(d_o.ptr))
    assert st == 1


@pytest.mark.parametrize("n_ants,targets,vel", [(40, ((80.0, 20.0, 1.5), (60.0, -40.0, 1.5)), (7.0, -3.0)),
                                                 (64, ((90.0, 40.0, 1.5),), (0.0,)),
                                                 (49, ((70.0, 20.0, 1.5), (60.0, -50.0, 1.5), (90.0, 10.0, 1.5)), (7.0, -3.0, 1.0))])
def test_fused_synthesis_covariance_kernel(pkg, ctx, n_ants, targets, vel):
    """33..64 antennas: the fused entry may run the synthesis inside the covariance kernel (echo_cov_kernel).  Whatever the
    library picks, the echo grid is bit-identical to the unfused spectral synthesis, Ra matches X X^H / N of that grid, and the
    cached fft2D gives the estimates of the plain call sequence."""
    sc = make_scene(n_ants=n_ants, n_slots=4, nrb=273, targets=targets, velocity=vel, seed=31, with_noise=False)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    rng = np.random.default_rng(5)
    w = np.asfortranarray(rng.standard_normal(sc.tx_grid.shape) + 1j * rng.standard_normal(sc.tx_grid.shape))
    d_wave, d_txg, d_w = ctx.to_device(sc.tx_wave), ctx.to_device(sc.tx_grid), ctx.to_device(w)
    for kw in (dict(spectral_noise=d_w), dict(seed=77, noise_domain="spectral")):
        e0 = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, **kw)
        est0, dbg0 = pkg.sensing.estimation.fft2D(rp, cf, e0, d_txg, return_debug=True)
        e1 = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, fuse_fft2d=(rp, cf, d_txg), **kw)
        est1, dbg1 = pkg.sensing.estimation.fft2D(rp, cf, e1, d_txg, return_debug=True, reuse_range=True)
        h0 = e0.numpy()
        assert np.array_equal(h0, e1.numpy())
        g = h0.reshape(-1, sc.A, order="F")
        ra = g.conj().T @ g / g.shape[0]
        assert rel(dbg1.Ra, ra) < RTOL and rel(dbg0.Ra, ra) < RTOL and np.array_equal(dbg1.Ra, dbg1.Ra.conj().T)
        assert np.array_equal(dbg0.power_window, dbg1.power_window)
        assert all(np.array_equal(a, b) for a, b in zip(dbg0.detections, dbg1.detections))
        assert np.array_equal(est0.rngEst, est1.rngEst) and np.array_equal(est0.velEst, est1.velEst) and np.array_equal(est0.aziEst, est1.aziEst)


@pytest.mark.parametrize("nrb", [24, 106, 133])
def test_fused_contract_holds_for_every_carrier(pkg, ctx, nrb):
    """Nfft != 4096 (24 / 106 / 133 PRB -> 512 / 2048 / 2048): the fused entry has no fused kernel for these, but the contract is the
    same -- monoStaticSensing(fuse_fft2d=...) followed by fft2D(reuse_range=True) -- and the results equal the plain sequence bit for bit."""
    sc = make_scene(n_ants=3, n_slots=4, nrb=nrb, targets=((90.0, 20.0, 1.5),), velocity=(5.0,), seed=41, with_noise=False, num_slots_param=6)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    d_wave, d_txg = ctx.to_device(sc.tx_wave), ctx.to_device(sc.tx_grid)
    for kw in (dict(seed=3, noise_domain="spectral"), dict(seed=3)):
        e0 = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=sc.wave.Nfft, **kw)
        try:
            est0, dbg0 = pkg.sensing.estimation.fft2D(rp, cf, e0, d_txg, return_debug=True)
        except pkg.IsacError as err:
            assert err.name == "NO_DETECTION"
            est0 = None
        # (the cache is single-use and per context: the fused call and its cached fft2D are consecutive)
        e1 = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=sc.wave.Nfft, fuse_fft2d=(rp, cf, d_txg), **kw)
        assert np.array_equal(e0.numpy(), e1.numpy())
        if est0 is None:
            with pytest.raises(pkg.IsacError) as ei:
                pkg.sensing.estimation.fft2D(rp, cf, e1, d_txg, reuse_range=True)
            assert ei.value.name == "NO_DETECTION"
            continue
        est1, dbg1 = pkg.sensing.estimation.fft2D(rp, cf, e1, d_txg, return_debug=True, reuse_range=True)
        assert np.array_equal(dbg0.power_window, dbg1.power_window)
        assert np.array_equal(est0.rngEst, est1.rngEst) and np.array_equal(est0.aziEst, est1.aziEst)
