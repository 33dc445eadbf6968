"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on identical
seeded inputs.  Tolerances (stated once): complex fp64 fields (waveforms, grids, |rdm|^2, Ra)
<= 1e-10 relative to the field's max magnitude; CFAR detection indices, range/velocity bin
estimates and integer-degree azimuths: exact."""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest
from scipy import linalg

import oracle as O
import os

from conftest import PKG_NAME, load_pkg, make_scene

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


def _guard_band_ok(p_map, cut, pfa, margin=1e-9):
    """True when no CUT sits within `margin` (relative) of its threshold, so that rounding-level
    differences in |rdm|^2 cannot flip a detection (SURVEY.md 7 'hard parts')."""
    _, thr = O.ca_cfar2d(p_map, cut, pfa, return_threshold=True)
    pc = p_map[cut[0] - 1, cut[1] - 1]
    return bool(np.all(np.abs(pc - thr) > margin * np.maximum(np.abs(thr), 1e-300)))


# ------------------------------------------------------------------ OFDM
@pytest.mark.parametrize("nrb,n_slots,n_ants", [(24, 2, 3), (106, 1, 2), (273, 2, 2)])
def test_ofdm_modulate_demodulate(pkg, ctx, nrb, n_slots, n_ants):
    sc = make_scene(n_ants=n_ants, n_slots=n_slots, nrb=nrb, with_noise=False, zero_s_slots=False)
    car = pkg._lib.Carrier(sc.K, sc.wave.Nfft, 30, 0)
    d_grid = ctx.to_device(sc.tx_grid)
    d_wave = ctx.empty((sc.T, sc.A))
    ctx.check(ctx.lib.isac_ofdm_modulate_dev(ctx.handle, C.c_void_p(d_grid.ptr), C.c_int32(sc.L), C.c_int32(sc.A), C.byref(car),
                                             C.c_double(sc.amp), C.c_void_p(d_wave.ptr), C.c_int64(sc.T)))
    assert rel(d_wave.numpy(), sc.tx_wave) < RTOL
    d_back = ctx.empty((sc.K, sc.L, sc.A))
    ctx.check(ctx.lib.isac_ofdm_demodulate_dev(ctx.handle, C.c_void_p(d_wave.ptr), C.c_int64(sc.T), C.c_int32(sc.A), C.byref(car),
                                               C.c_void_p(d_back.ptr), C.c_int32(sc.L)))
    want = O.ofdm_demodulate(sc.tx_wave, sc.K, sc.wave.Nfft, 30)
    assert rel(d_back.numpy(), want) < RTOL
    assert rel(d_back.numpy() / sc.amp, sc.tx_grid) < 1e-9      # encode -> decode round trip


# ------------------------------------------------------------------ echo synthesis
@pytest.mark.parametrize("nrb,n_ants,targets,vel,los", [
    (24, 4, ((150.0, 40.0, 1.5),), (0.0,), (1,)),
    (24, 5, ((150.0, 40.0, 1.5), (-80.0, 60.0, 10.0), (300.0, -20.0, 1.5)), (4.0, -9.0, 2.0), (1, 0, 1)),
    (273, 2, ((100.0, 20.0, 1.5), (260.0, -200.0, 1.5)), (7.0, -10.0), (1, 1)),
])
def test_basic_radar_channel_and_mono_static(pkg, ctx, nrb, n_ants, targets, vel, los):
    sc = make_scene(n_ants=n_ants, n_slots=2, nrb=nrb, targets=targets, velocity=vel)
    los = np.array(los)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    want_rx = O.basic_radar_channel(sc.tx_wave, sc.rp, los, sc.noise)
    got_rx = pkg.sensing.channelModels.basicRadarChannel(sc.tx_wave, rp, los, noise=sc.noise)
    assert rel(got_rx, want_rx) < RTOL
    want = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, los, sc.noise, nfft=sc.wave.Nfft)
    got = pkg.sensing.monoStaticSensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, rp, los, noise=sc.noise, nfft=sc.wave.Nfft)
    assert got.shape == want.shape and rel(got, want) < RTOL
    # device-resident path gives the same bits as the host-pointer path
    d = pkg.sensing.monoStaticSensing(ctx.to_device(sc.tx_wave), sc.tx_grid.shape, sc.carrier, rp, los,
                                      noise=ctx.to_device(sc.noise), nfft=sc.wave.Nfft)
    assert np.array_equal(d.numpy(), got)
    # noiseless + linearity in the waveform (size-independent property)
    g1 = pkg.sensing.monoStaticSensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, rp, los, nfft=sc.wave.Nfft)
    g2 = pkg.sensing.monoStaticSensing(2.0 * sc.tx_wave, sc.tx_grid.shape, sc.carrier, rp, los, nfft=sc.wave.Nfft)
    assert rel(g2, 2.0 * g1) < 1e-14


def test_symbol_padding_and_partial_symbol(pkg, ctx):
    sc = make_scene(n_ants=2, n_slots=1, nrb=24, with_noise=False)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    wave = sc.tx_wave[:-7]                                  # last symbol incomplete -> 13 whole symbols
    got = pkg.sensing.monoStaticSensing(wave, (sc.K, 20, sc.A), sc.carrier, rp, sc.los, nfft=sc.wave.Nfft)
    want = O.mono_static_sensing(wave, (sc.K, 20, sc.A), sc.carrier, sc.rp, sc.los, None, nfft=sc.wave.Nfft)
    assert got.shape == (sc.K, 20, sc.A) and rel(got, want) < RTOL
    assert np.all(got[:, 13:, :] == 0)                      # monoStaticSensing.m:19-21


def test_philox_noise_mode_matches_restated_generator(pkg, ctx):
    sc = make_scene(n_ants=3, n_slots=1, nrb=24, with_noise=False)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    seed = 0x5EED0002
    e = np.arange(sc.T * sc.A, dtype=np.uint64).reshape(sc.T, sc.A, order="F")
    noise = O.philox_normal_pairs(e, seed)
    want = O.basic_radar_channel(sc.tx_wave, sc.rp, sc.los, noise)
    got = pkg.sensing.channelModels.basicRadarChannel(sc.tx_wave, rp, sc.los, seed=seed)
    assert rel(got, want) < RTOL
    assert abs(np.std(noise.real) - 1) < 0.02 and abs(np.std(noise.imag) - 1) < 0.02


def test_all_nlos_and_short_waveform_errors(pkg, ctx):
    sc = make_scene(n_ants=2, n_slots=1, nrb=24, with_noise=False)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    with pytest.raises(pkg.IsacError) as ei:
        pkg.sensing.monoStaticSensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, rp, np.zeros(1), nfft=sc.wave.Nfft)
    assert ei.value.name == "NO_LOS"
    with pytest.raises(pkg.IsacError) as ei:
        pkg.sensing.monoStaticSensing(sc.tx_wave[:100], sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=sc.wave.Nfft)
    assert ei.value.name == "SHORT_WAVEFORM"


# ------------------------------------------------------------------ CFAR alone: bit-exact on identical power maps
def test_cfar_detector_bit_exact(pkg, ctx):
    rng = np.random.default_rng(7)
    det = pkg.sensing.detection.CFARDetector2D(1e-4, (2, 2), (1, 1))
    for shape in [(64, 40), (4096, 256)]:
        p = rng.exponential(1.0, shape)
        for _ in range(40):
            p[rng.integers(5, shape[0] - 5), rng.integers(5, shape[1] - 5)] *= rng.uniform(20, 400)
        rows, cols = np.arange(4, shape[0] - 3), np.arange(4, shape[1] - 3)
        cc, rr = np.meshgrid(cols, rows)
        cut = np.stack([rr.ravel(order="F"), cc.ravel(order="F")])
        want = O.ca_cfar2d(p, cut, 1e-4)
        got = det(p, cut)
        assert want.shape[1] > 10 and np.array_equal(got, want)
    with pytest.raises(pkg.IsacError) as ei:
        det(np.ones((20, 20)), np.array([[3], [10]]))
    assert ei.value.name == "CFAR_WINDOW"
    assert det(np.zeros((20, 20)), np.array([[10], [10]])).shape == (2, 0)
    det2 = pkg.sensing.detection.CFARDetector2D(1e-3, (1, 3), (2, 1))     # other band sizes
    p = rng.exponential(1.0, (50, 60)); p[25, 30] = 300
    cut = np.stack([np.repeat(np.arange(6, 44), 1), np.full(38, 30)])
    assert np.array_equal(det2(p, cut), O.ca_cfar2d(p, cut, 1e-3, (1, 3), (2, 1)))


# ------------------------------------------------------------------ fft2D end to end
def _run_fft2d_case(pkg, sc, los=None, hip_echo=False):
    los = sc.los if los is None else los
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    rx = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, los, sc.noise, nfft=sc.wave.Nfft)
    if hip_echo:   # the grid fft2D consumes is the HIP echo path's own output (checked against the oracle's first)
        got_rx = pkg.sensing.monoStaticSensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, rp, los, noise=sc.noise, nfft=sc.wave.Nfft)
        assert rel(got_rx, rx) < RTOL
    ocf = O.cfar2d_config(sc.rp)
    want, dbg = O.fft2d(sc.rp, ocf, rx, sc.tx_grid, return_debug=True)
    got, gd = pkg.sensing.estimation.fft2D(rp, cf, got_rx if hip_echo else rx, sc.tx_grid, return_debug=True)
    # |rdm|^2 window
    r0, c0 = gd.first_row - 1, gd.first_col - 1
    nr, nc, na = gd.power_window.shape
    p_ref = np.abs(dbg.rdm[r0:r0 + nr, c0:c0 + nc, :]) ** 2
    assert rel(gd.power_window, p_ref) < RTOL
    assert rel(gd.Ra, dbg.Ra) < RTOL and np.array_equal(gd.Ra, gd.Ra.conj().T)
    for a in range(sc.A):
        assert _guard_band_ok(np.abs(dbg.rdm[:, :, a]) ** 2, ocf.CUTIdx, ocf.Pfa), "scene too close to a threshold"
        assert np.array_equal(gd.detections[a], dbg.detections[a]), f"antenna {a}"
    assert np.array_equal(got.rngEst, want.rngEst) and np.array_equal(got.velEst, want.velEst)
    assert np.array_equal(got.aziEst, want.aziEst) and np.isnan(got.eleEst).all()
    return got, gd, rx


def test_fft2d_small(pkg, ctx):
    sc = make_scene(n_ants=4, n_slots=4, nrb=24, targets=((150.0, 40.0, 1.5),), velocity=(0.0,), num_slots_param=6,
                    zero_s_slots=False)
    got, gd, rx = _run_fft2d_case(pkg, sc)
    assert got.rngEst.size >= 1
    # device-resident inputs give identical results
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    got2 = pkg.sensing.estimation.fft2D(rp, cf, ctx.to_device(rx), ctx.to_device(sc.tx_grid))
    assert np.array_equal(got2.rngEst, got.rngEst) and np.array_equal(got2.aziEst, got.aziEst)


def test_fft2d_every_other_cut_detects(pkg, ctx):
    """Pfa = 0.5: more than 4096 detections per antenna (the default zone has 8 510 CUTs) -- phased.CFARDetector2D reports
    every one of them (fft2D.m:62-78), so must the device path: detection lists, their CUT order and the estimates stay exact."""
    sc = make_scene(n_ants=2, n_slots=4, nrb=273, targets=((100.0, 20.0, 1.5),), velocity=(7.0,), seed=21)
    sc.rp.Pfa = 0.5
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    rp.Pfa = 0.5
    cf = pkg.sensing.detection.cfar2D(rp)
    ocf = O.cfar2d_config(sc.rp)
    rx = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, sc.noise, nfft=sc.wave.Nfft)
    want, dbg = O.fft2d(sc.rp, ocf, rx, sc.tx_grid, return_debug=True)
    got, gd = pkg.sensing.estimation.fft2D(rp, cf, rx, sc.tx_grid, return_debug=True)
    assert min(d.shape[1] for d in dbg.detections) > 4096
    for a in range(sc.A):
        assert _guard_band_ok(np.abs(dbg.rdm[:, :, a]) ** 2, ocf.CUTIdx, ocf.Pfa), "scene too close to a threshold"
        assert np.array_equal(gd.detections[a], dbg.detections[a]), f"antenna {a}"
    assert np.array_equal(got.rngEst, want.rngEst) and np.array_equal(got.velEst, want.velEst)
    assert np.array_equal(got.aziEst, want.aziEst)
    assert got.rngEst.size > 300                      # practically every range row of the zone


@pytest.mark.parametrize("area,blocks", [(((650.0, 1000.0), (-50.0, 50.0)), "one block, not the first"),
                                         (((500.0, 800.0), (-50.0, 50.0)), "straddles two 512-row blocks"),
                                         (((50.0, 500.0), (-50.0, 50.0)), "default: inside the first block")] +
                                        [(((625.0 * i + 20, 625.0 * (i + 1) - 25), (-50.0, 50.0)), f"block {i}") for i in range(2, 8)])
def test_fft2d_range_window_position(pkg, ctx, area, blocks):
    """The range transform forms only the 512-row output block the CUT rows need when they fit in one (pruned last radix-8 pass),
    all eight otherwise: zones in a later block and across a block boundary against the oracle, fused and unfused (Pfa = 0.5 so that
    far zones detect at all)."""
    sc = make_scene(n_ants=3, n_slots=4, nrb=273, targets=((100.0, 20.0, 1.5),), velocity=(7.0,), seed=17, detection_area=area)
    sc.rp.Pfa = 0.5
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    rp.Pfa = 0.5
    cf = pkg.sensing.detection.cfar2D(rp)
    ocf = O.cfar2d_config(sc.rp)
    rows = ocf.CUTIdx[0]
    if "not the first" in blocks:
        assert rows.min() - 3 >= 513 and rows.max() + 3 <= 1024
    if "straddles" in blocks:
        assert rows.min() < 512 < rows.max()
    if blocks.startswith("block "):                              # every output digit of the single-output last pass (dft8_one)
        i = int(blocks.split()[1])
        assert (rows.min() - 4) // 512 == i == (rows.max() + 2) // 512
    rx = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, sc.noise, nfft=sc.wave.Nfft)
    want, dbg = O.fft2d(sc.rp, ocf, rx, sc.tx_grid, return_debug=True)
    d_wave, d_noise, d_txg = ctx.to_device(sc.tx_wave), ctx.to_device(sc.noise), ctx.to_device(sc.tx_grid)
    e0 = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, noise=d_noise, nfft=4096)
    got, gd = pkg.sensing.estimation.fft2D(rp, cf, e0, d_txg, return_debug=True)
    r0, c0 = gd.first_row - 1, gd.first_col - 1
    nr, nc, _ = gd.power_window.shape
    assert rel(gd.power_window, np.abs(dbg.rdm[r0:r0 + nr, c0:c0 + nc, :]) ** 2) < RTOL
    for a in range(sc.A):
        assert _guard_band_ok(np.abs(dbg.rdm[:, :, a]) ** 2, ocf.CUTIdx, ocf.Pfa), "scene too close to a threshold"
        assert np.array_equal(gd.detections[a], dbg.detections[a]), f"antenna {a}"
    assert np.array_equal(got.rngEst, want.rngEst) and np.array_equal(got.velEst, want.velEst) and np.array_equal(got.aziEst, want.aziEst)
    # the fused spectral kernel takes the same decision about the block: bit-identical to its unfused sequence
    e1 = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, seed=5, noise_domain="spectral")
    _, g1 = pkg.sensing.estimation.fft2D(rp, cf, e1, d_txg, return_debug=True)
    e2 = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, seed=5, noise_domain="spectral", fuse_fft2d=(rp, cf, d_txg))
    _, g2 = pkg.sensing.estimation.fft2D(rp, cf, e2, d_txg, return_debug=True, reuse_range=True)
    assert np.array_equal(g1.power_window, g2.power_window)


def test_fused_range_stage_is_identical(pkg, ctx):
    """monoStaticSensing(fuse_fft2d=...) + fft2D must give bit-identical echo grid, |rdm|^2 window and
    detections to the unfused call sequence (full-size numerology: the fused kernel needs Nfft == nIFFT == 4096)."""
    sc = make_scene(n_ants=3, n_slots=4, nrb=273, targets=((100.0, 20.0, 1.5),), velocity=(7.0,), seed=9)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    d_wave, d_noise, d_txg = ctx.to_device(sc.tx_wave), ctx.to_device(sc.noise), ctx.to_device(sc.tx_grid)
    e0 = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, noise=d_noise, nfft=4096)
    est0, dbg0 = pkg.sensing.estimation.fft2D(rp, cf, e0, d_txg, return_debug=True)
    e1 = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, noise=d_noise, nfft=4096,
                                       fuse_fft2d=(rp, cf, d_txg))
    est1, dbg1 = pkg.sensing.estimation.fft2D(rp, cf, e1, d_txg, return_debug=True, reuse_range=True)
    assert np.array_equal(e0.numpy(), e1.numpy())
    assert np.array_equal(dbg0.power_window, dbg1.power_window)
    assert all(np.array_equal(a, b) for a, b in zip(dbg0.detections, dbg1.detections))
    assert np.array_equal(est0.rngEst, est1.rngEst) and np.array_equal(est0.aziEst, est1.aziEst)
    # reuse is explicit and keyed on the grids: a different rxGrid cannot consume the cache, a plain fft2D never does
    other = ctx.to_device(np.asfortranarray(2.0 * e0.numpy()))
    e2 = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, noise=d_noise, nfft=4096,
                                       fuse_fft2d=(rp, cf, d_txg))
    with pytest.raises(pkg.IsacError) as ei:
        pkg.sensing.estimation.fft2D(rp, cf, other, d_txg, reuse_range=True)
    assert ei.value.name == "INVALID_ARG"
    _, dbg2 = pkg.sensing.estimation.fft2D(rp, cf, other, d_txg, return_debug=True)
    assert np.allclose(dbg2.power_window, 4.0 * dbg0.power_window, rtol=1e-12)
    # padded symbol dimension (txDimension(2) > whole symbols) through the fused path
    e3 = pkg.sensing.monoStaticSensing(ctx.to_device(sc.tx_wave[:-9]), sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096,
                                       fuse_fft2d=(rp, cf, d_txg))
    e4 = pkg.sensing.monoStaticSensing(ctx.to_device(sc.tx_wave[:-9]), sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096)
    assert np.array_equal(e3.numpy(), e4.numpy()) and np.all(e3.numpy()[:, -1, :] == 0)


def test_fft2d_multi_target_odd_antennas(pkg, ctx):
    sc = make_scene(n_ants=6, n_slots=8, nrb=51, targets=((120.0, 60.0, 1.5), (-250.0, 80.0, 1.5)), velocity=(10.0, -6.0),
                    num_slots_param=12, seed=5)
    _run_fft2d_case(pkg, sc)


def test_fft2d_full_size_grid(pkg, ctx):
    """273 PRB / 4096-point range IFFT / 256 Doppler bins (the benchmark shape), 4 antennas."""
    sc = make_scene(n_ants=4, n_slots=16, nrb=273, targets=((100.0, 20.0, 1.5), (-180.0, 150.0, 1.5)), velocity=(7.0, -4.0),
                    seed=11)
    got, gd, _ = _run_fft2d_case(pkg, sc)
    assert gd.power_window.shape[:2] == (376, 29) and gd.first_row == 39 and gd.first_col == 115


def test_fft2d_256_element_array(pkg, ctx):
    """Config 4 shape on the array side: 256-element ULA (generic MFMA covariance, global-memory Jacobi)."""
    sc = make_scene(n_ants=256, n_slots=2, nrb=24, targets=((150.0, 40.0, 1.5),), velocity=(0.0,), num_slots_param=3,
                    zero_s_slots=False, seed=13)
    _run_fft2d_case(pkg, sc, hip_echo=True)


def test_fft2d_zero_detections_is_an_error(pkg, ctx):
    sc = make_scene(n_ants=2, n_slots=4, nrb=24, num_slots_param=6, with_noise=False)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    with pytest.raises(pkg.IsacError) as ei:
        pkg.sensing.estimation.fft2D(rp, cf, np.zeros_like(sc.tx_grid), sc.tx_grid)
    assert ei.value.name == "NO_DETECTION"


# ------------------------------------------------------------------ covariance / eig / MUSIC
@pytest.mark.parametrize("n,a", [(8064, 4), (4096 + 37, 19), (733824, 16), (65536, 64), (20000, 100), (8192, 256), (5003, 65), (3001, 130), (40000, 200),
                                 (1, 3), (15, 5), (17, 64), (1, 70), (31, 129), (16, 256),
                                 (5000, 40), (777, 33), (12345, 48), (100003, 17), (4099, 32)])     # 2 and 3 antenna blocks, ragged sample counts
def test_covariance_mfma(pkg, ctx, n, a):
    rng = np.random.default_rng(a)
    g = np.asfortranarray(rng.standard_normal((n, a)) + 1j * rng.standard_normal((n, a)))
    g[:, 0] *= 3.0                                           # asymmetric: catches row/col swaps
    d_ra = ctx.empty((a, a))
    d_g = ctx.to_device(g)                                   # keep alive until the result is read back
    ctx.check(ctx.lib.isac_covariance_dev(ctx.handle, C.c_void_p(d_g.ptr), C.c_int64(n), C.c_int32(a), C.c_void_p(d_ra.ptr)))
    ra = d_ra.numpy()
    want = g.conj().T @ g / n
    assert rel(ra, want) < 1e-12 and np.array_equal(ra, ra.conj().T)


@pytest.mark.parametrize("a", [1, 2, 5, 16, 33, 64, 65, 96, 256, 320, 640])
def test_eigh_jacobi(pkg, ctx, a):
    rng = np.random.default_rng(a)
    m = rng.standard_normal((a, a)) + 1j * rng.standard_normal((a, a))
    h = np.asfortranarray(m @ m.conj().T / a + np.diag(rng.uniform(0, 3, a)))
    w = np.zeros(a)
    v = np.zeros((a, a), dtype=np.complex128, order="F")
    ctx.check(ctx.lib.isac_eigh(ctx.handle, h.ctypes.data_as(C.c_void_p), C.c_int32(a), w.ctypes.data_as(C.c_void_p),
                                v.ctypes.data_as(C.c_void_p)))
    wr = linalg.eigvalsh(h)
    assert np.abs(w - wr).max() < 1e-12 * np.abs(wr).max()
    assert np.abs(v.conj().T @ v - np.eye(a)).max() < 1e-12
    assert np.abs(h @ v - v * w).max() < 1e-11 * np.abs(wr).max()


_RECOVER_SNIPPET = r"""
import ctypes as C, importlib, sys
import numpy as np
sys.path.insert(0, %r)
pkg = importlib.import_module(%r)
ctx = pkg._lib.Context(0)
for a in (33, 64, 100, 256):
    rng = np.random.default_rng(a)
    m = rng.standard_normal((a, a)) + 1j * rng.standard_normal((a, a))
    h = np.asfortranarray(m @ m.conj().T / a + np.diag(rng.uniform(0, 3, a)))
    w = np.zeros(a); v = np.zeros((a, a), dtype=np.complex128, order="F")
    ctx.check(ctx.lib.isac_eigh(ctx.handle, h.ctypes.data_as(C.c_void_p), C.c_int32(a), w.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)))
    wr = np.linalg.eigvalsh(h)
    assert np.abs(w - wr).max() < 1e-12 * np.abs(wr).max(), a
    assert np.abs(v.conj().T @ v - np.eye(a)).max() < 1e-12, a
    assert np.abs(h @ v - v * w).max() < 1e-11 * np.abs(wr).max(), a
print("recovered")
"""


def test_eigh_live_replay_timeout_is_recovered():
    """A live replay block that gives up waiting (info[0] = -2) must not fail the call: the recorded rotations are replayed by a launch of its own.
    ISAC_EIG_FORCE_REPLAY_TIMEOUT (read once per process, hence the subprocess) destroys the eigenvectors of every QL-pipeline call and takes
    that path; ISAC_EIG_QL sends A <= 64 through the pipeline too."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ISAC_EIG_FORCE_REPLAY_TIMEOUT="1", ISAC_EIG_QL="1")
    r = subprocess.run([sys.executable, "-c", _RECOVER_SNIPPET % (root, PKG_NAME)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "recovered" in r.stdout, r.stdout + r.stderr


_TRIDIAG_TIMEOUT_SNIPPET = """
import ctypes as C, importlib, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
pkg = importlib.import_module(%r)
from conftest import make_scene
ctx = pkg._lib.Context(0)
def herm(a):
    rng = np.random.default_rng(a)
    m = rng.standard_normal((a, a)) + 1j * rng.standard_normal((a, a))
    return np.asfortranarray(m @ m.conj().T / a + np.diag(rng.uniform(0, 3, a)))
def eigh(a):
    h = herm(a); w = np.zeros(a); v = np.zeros((a, a), dtype=np.complex128, order="F")
    ctx.check(ctx.lib.isac_eigh(ctx.handle, h.ctypes.data_as(C.c_void_p), C.c_int32(a), w.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)))
    return h, w, v
def eigh_top(a, k):
    h = herm(a); w = np.zeros(a); v = np.zeros((a, k), dtype=np.complex128, order="F")     # (w: ALL eigenvalues)
    ctx.check(ctx.lib.isac_eigh_top(ctx.handle, h.ctypes.data_as(C.c_void_p), C.c_int32(a), C.c_int32(k), w.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)))
n_err = 0
for i, call in enumerate((lambda: eigh(100), lambda: eigh_top(100, 3), lambda: eigh(256))):
    print("call", i, flush=True)
    try:
        call()
    except pkg.IsacError as e:
        assert "tridiagonalisation" in str(e), str(e)
        n_err += 1
assert n_err == 3, n_err
print("eigh calls failed as they must", flush=True)
# the one-workgroup route (A <= 64) of the same context is untouched by the stale status
h, w, v = eigh(48)
assert np.abs(w - np.linalg.eigvalsh(h)).max() < 1e-12 * np.abs(w).max()
print("A = 48 fine", flush=True)
# the whole chain at 72 antennas: the CPI must fail, not return azimuths computed from a half-written tridiagonal form
sc = make_scene(n_ants=72, n_slots=2, nrb=24, targets=((120.0, 60.0, 1.5),), velocity=(5.0,), seed=3)
rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
cf = pkg.sensing.detection.cfar2D(rp)
echo = pkg.sensing.monoStaticSensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, rp, np.ones(1, np.uint8), noise=sc.noise, nfft=sc.wave.Nfft)
try:
    pkg.sensing.estimation.fft2D(rp, cf, echo, sc.tx_grid)
    raise SystemExit("fft2D returned estimates after a timed-out tridiagonalisation")
except pkg.IsacError as e:
    assert e.name != "NO_DETECTION", str(e)
print("surfaced")
"""


def test_distributed_tridiagonalisation_timeout_surfaces():
    """eigh_tridiag_dist_kernel reports an exchange time-out in info[0] AND in the sticky word info[6]: the kernels that follow on the stream (QL pipeline,
    subspace kernel, replay) overwrite info[0] and must keep the -4.  ISAC_EIG_FORCE_TRIDIAG_TIMEOUT (read once per process, hence the subprocess) makes
    every distributed reduction report one: isac_eigh, isac_eigh_top and the fft2D chain at A > 64 must fail; A <= 64 on the same context must not."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ISAC_EIG_FORCE_TRIDIAG_TIMEOUT="1")
    r = subprocess.run([sys.executable, "-c", _TRIDIAG_TIMEOUT_SNIPPET % (root, root, PKG_NAME)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "surfaced" in r.stdout, r.stdout + r.stderr


_TRIDIAG_SNIPPET = """
import ctypes as C, hashlib, importlib, sys
import numpy as np
sys.path.insert(0, %r)
pkg = importlib.import_module(%r)
ctx = pkg._lib.Context(0)
for a in (65, 97, 128, 200, 255, 256):
    rng = np.random.default_rng(a)
    m = rng.standard_normal((a, a)) + 1j * rng.standard_normal((a, a))
    h = np.asfortranarray(m @ m.conj().T / a + np.diag(rng.uniform(0, 3, a)))
    w = np.zeros(a); v = np.zeros((a, a), dtype=np.complex128, order="F")
    for rep in range(3):                                   # (repeated launches: the exchange tags carry the launch epoch, nothing is reset in between)
        ctx.check(ctx.lib.isac_eigh(ctx.handle, h.ctypes.data_as(C.c_void_p), C.c_int32(a), w.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)))
        wr = np.linalg.eigvalsh(h)
        assert np.abs(w - wr).max() < 1e-12 * np.abs(wr).max() and np.abs(h @ v - v * w).max() < 1e-11 * np.abs(wr).max(), (a, rep)
    print("digest", a, hashlib.sha256(w.tobytes() + v.tobytes()).hexdigest())
"""


def test_distributed_tridiagonalisation_modes_agree():
    """eigh_tridiag_dist_kernel (64 < n <= 256) under its development switches (read once per process, hence the subprocesses): the default (every working
    workgroup on one XCD, exchange resident in its L2), write-through exchange on one XCD ("far"), workgroups dealt over all XCDs ("s1": the kernel sees
    different XCC ids and uses the write-through stores by itself) -- the SAME bits, as placement must never change a result; the one-workgroup kernels
    ("0") agree to rounding (checked against numpy.linalg in every mode)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for mode in ("", "far", "s1", "0"):
        env = dict(os.environ, ISAC_EIG_TRIDIAG_DIST=mode)
        r = subprocess.run([sys.executable, "-c", _TRIDIAG_SNIPPET % (root, PKG_NAME)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        out[mode] = [ln for ln in r.stdout.splitlines() if ln.startswith("digest")]
        assert len(out[mode]) == 6
    assert out[""] == out["far"] == out["s1"]
    assert out["0"] != out[""]                              # (the switch did switch)


def test_distributed_tridiagonalisation_concurrent_contexts(pkg):
    """Twelve host threads, each with its own context, run isac_eigh at n = 256 / 200 at the same time: up to twelve distributed reductions in flight, each of
    which needs its 13-16 workgroups resident together (consecutive launches go to consecutive XCDs).  No exchange may time out (status -4) and every result must
    be the single-context result."""
    import threading
    rng = np.random.default_rng(5)
    mats = {}
    for a in (256, 200):
        m = rng.standard_normal((a, a)) + 1j * rng.standard_normal((a, a))
        mats[a] = np.asfortranarray(m @ m.conj().T / a + np.diag(rng.uniform(0, 3, a)))
    ref = {a: np.linalg.eigvalsh(h) for a, h in mats.items()}
    errors = []

    def worker(tid):
        try:
            c = pkg._lib.Context(0)
            for rep in range(6):
                a = 256 if (tid + rep) % 2 == 0 else 200
                h = mats[a]
                w = np.zeros(a); v = np.zeros((a, a), dtype=np.complex128, order="F")
                c.check(c.lib.isac_eigh(c.handle, h.ctypes.data_as(C.c_void_p), C.c_int32(a), w.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)))
                if not (np.abs(w - ref[a]).max() < 1e-12 * np.abs(ref[a]).max() and np.abs(h @ v - v * w).max() < 1e-11 * np.abs(ref[a]).max()):
                    errors.append((tid, rep, "wrong result"))
        except Exception as e:                                # noqa: BLE001 -- collected and reported by the main thread
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(12)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:4]


@pytest.mark.parametrize("kind,a", [("identity", 100), ("rank2", 130), ("diag_repeated", 96), ("tiny", 72), ("huge", 65),
                                    ("clustered", 200), ("tridiag_zero_blocks", 128),
                                    ("identity", 12), ("rank2", 40), ("tiny", 33), ("huge", 64), ("clustered", 48), ("zero", 20), ("zero", 80)])
def test_eigh_degenerate_spectra(pkg, ctx, kind, a):
    """Edge cases of both eigensolvers (Jacobi A <= 64, tridiagonal/QL pipeline beyond): exact splits, zero and repeated
    eigenvalues, scales near the fp64 range limits (zheev-style safe scaling), tight clusters, the zero matrix."""
    rng = np.random.default_rng(a)
    def rand_unitary(n):
        q, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
        return q
    if kind == "identity":
        h = np.eye(a, dtype=np.complex128)
    elif kind == "rank2":
        x = rng.standard_normal((a, 2)) + 1j * rng.standard_normal((a, 2))
        h = x @ x.conj().T
    elif kind == "zero":
        h = np.zeros((a, a), dtype=np.complex128)
    elif kind == "diag_repeated":
        h = np.diag(np.repeat([3.0, -1.0, 0.0, 7.5], a // 4)).astype(np.complex128)
    elif kind == "tiny":
        q = rand_unitary(a)
        h = (q * rng.uniform(0.5, 2.0, a)) @ q.conj().T * 1e-170
    elif kind == "huge":
        q = rand_unitary(a)
        h = (q * rng.uniform(0.5, 2.0, a)) @ q.conj().T * 1e150
    elif kind == "clustered":
        q = rand_unitary(a)
        w0 = np.concatenate([[100.0, 37.0, 5.0], 1.0 + 1e-13 * rng.standard_normal(a - 3)])
        h = (q * w0) @ q.conj().T
    else:
        h = np.zeros((a, a), dtype=np.complex128)
        for b0 in range(0, a, 16):                   # decoupled 16 x 16 Hermitian blocks
            m = rng.standard_normal((16, 16)) + 1j * rng.standard_normal((16, 16))
            h[b0:b0 + 16, b0:b0 + 16] = m + m.conj().T
    h = np.asfortranarray((h + h.conj().T) / 2)
    w = np.zeros(a)
    v = np.zeros((a, a), dtype=np.complex128, order="F")
    ctx.check(ctx.lib.isac_eigh(ctx.handle, h.ctypes.data_as(C.c_void_p), C.c_int32(a), w.ctypes.data_as(C.c_void_p),
                                v.ctypes.data_as(C.c_void_p)))
    wr = linalg.eigvalsh(h)
    scale = max(np.abs(wr).max(), 1e-300)
    assert np.all(np.isfinite(w)) and np.all(np.isfinite(v))
    assert np.abs(w - wr).max() < 1e-12 * scale
    assert np.abs(v.conj().T @ v - np.eye(a)).max() < 1e-12
    assert np.abs(h @ v - v * w).max() < 1e-11 * scale


def test_music_doa_kat_and_mirror_ties(pkg, ctx):
    sc = make_scene(n_ants=16, n_slots=1, nrb=24, with_noise=False)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    m = np.arange(16)
    for phi0 in (20, 37, -30, -61, 45):
        a = np.exp(-2j * np.pi * m * 0.5 * float(O.sind(phi0)))
        ra = np.outer(a, a.conj()) + 1e-3 * np.eye(16)
        for l in (1, 2):
            want = O.music_doa(l, sc.rp, ra)
            got = pkg.sensing.estimation.doaEstimation.music(l, rp, ra)
            assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.isnan(got[2]).all()
    # two sources + model-order estimate ([] -> determineNumTargets)
    a1 = np.exp(-2j * np.pi * m * 0.5 * float(O.sind(15)))
    a2 = np.exp(-2j * np.pi * m * 0.5 * float(O.sind(-40)))
    # distinct noise eigenvalues: with a flat floor the model-order rule argmaxes over rounding noise
    ra = 4 * np.outer(a1, a1.conj()) + np.outer(a2, a2.conj()) + np.diag(np.random.default_rng(3).uniform(0.01, 0.03, 16))
    want = O.music_doa(None, sc.rp, ra)
    got = pkg.sensing.estimation.doaEstimation.music(None, rp, ra)
    assert got[0] == want[0] and np.array_equal(got[1], want[1])
    with pytest.raises(pkg.IsacError) as ei:
        pkg.sensing.estimation.doaEstimation.music(0, rp, np.eye(16))
    assert ei.value.name == "NO_DETECTION"
    assert pkg.sensing.estimation.doaEstimation.music(16, rp, np.eye(16))[1].size == 0


# ------------------------------------------------------------------ whole chain on the device, golden fixture
def test_chain_against_golden_fixture(pkg, ctx):
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "chain_small.npz")
    g = np.load(path)
    sc = make_scene(n_ants=int(g["n_ants"]), n_slots=int(g["n_slots"]), nrb=int(g["nrb"]), targets=tuple(map(tuple, g["targets"])),
                    velocity=tuple(g["velocity"]), num_slots_param=int(g["num_slots_param"]), seed=int(g["seed"]))
    import hashlib
    assert hashlib.sha256(np.ascontiguousarray(sc.tx_grid).tobytes()).hexdigest() == str(g["tx_grid_sha256"])   # seeded generator is stable
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    echo = pkg.sensing.monoStaticSensing(ctx.to_device(sc.tx_wave), sc.tx_grid.shape, sc.carrier, rp, sc.los,
                                         noise=ctx.to_device(sc.noise), nfft=sc.wave.Nfft)
    assert rel(echo.numpy()[::5, ::3, :], g["echo_grid_sub"]) < RTOL
    est, dbg = pkg.sensing.estimation.fft2D(rp, cf, echo, ctx.to_device(sc.tx_grid), return_debug=True)
    assert np.array_equal(est.rngEst, g["rngEst"]) and np.array_equal(est.velEst, g["velEst"])
    assert np.array_equal(est.aziEst, g["aziEst"])
    assert np.array_equal(np.concatenate(dbg.detections, axis=1), g["det_idx"])


# ------------------------------------------------------------------ CDL MIMO channel apply
UE_ARRAY, GNB64 = (1, 1, 2, 1, 1), (1, 32, 2, 1, 1)


@pytest.mark.parametrize("profile,tx_size,rx_size,t_len,t0", [
    ("CDL-D", (1, 4, 2, 1, 1), UE_ARRAY, 7680 + 65, 0.0),
    ("CDL-A", (1, 8, 2, 1, 1), UE_ARRAY, 3000, 0.0),
    ("CDL-A", (1, 3, 2, 1, 1), UE_ARRAY, 4097, 1.0 / 640 - 1000 / 15.36e6),       # crosses a path-gain refresh inside the block
    ("CDL-D", GNB64, UE_ARRAY, 7680 + 65, 0.25),                                   # DL: 64 transmit elements (the benchmark array), cdl.m:57-64
    ("CDL-D", UE_ARRAY, GNB64, 7680 + 65, 0.0),                                    # UL: Nt = 2 -> Nr = 64 (cdl.m:78-85, stepped at gNBPhy.m:838-840)
    ("CDL-A", UE_ARRAY, GNB64, 4097, 1.0 / 640 - 1000 / 15.36e6),                  # UL, NLoS profile, across a path-gain refresh
    ("CDL-A", UE_ARRAY, (1, 8, 2, 1, 1), 3000, 0.1),                               # UL into the reference's default 16-element array
])
def test_cdl_apply_matches_oracle(pkg, ctx, profile, tx_size, rx_size, t_len, t0):
    import oracle.cdl as OC
    fs = 15.36e6
    cfg = OC.cdl_config(profile, 3.5e9, tx_size, rx_size, fs)
    ch = pkg.communication.channelModels.CDLChannel(profile, 300e-9, 3.5e9, tx_size, rx_size, fs)
    ch.time = t0
    nt, nr = int(np.prod(tx_size)), int(np.prod(rx_size))
    rng = np.random.default_rng(nt + nr)
    x = np.asfortranarray(rng.standard_normal((t_len, nt)) + 1j * rng.standard_normal((t_len, nt)))
    want = OC.apply_cdl(cfg, x, t0)
    got = pkg.communication.channelModels.applyCDL(ch, x)
    assert got.shape == (t_len, nr) and rel(got, want) < RTOL
    assert ch.time == pytest.approx(t0 + t_len / fs)
    # device-resident call continues from the advanced channel time
    got2 = pkg.communication.channelModels.applyCDL(ch, ctx.to_device(x)).numpy()
    assert rel(got2, OC.apply_cdl(cfg, x, t0 + t_len / fs)) < RTOL


@pytest.mark.parametrize("profile,tx_size,rx_size", [("CDL-D", GNB64, UE_ARRAY), ("CDL-A", (1, 8, 2, 1, 1), UE_ARRAY), ("CDL-A", UE_ARRAY, (1, 8, 2, 1, 1))])
def test_cdl_batch_apply_matches_oracle(pkg, ctx, profile, tx_size, rx_size):
    """isac_cdl_apply_batch_dev + isac_cdl_path_gains_dev: five (UE, slot) jobs in one call -- three UEs on ONE waveform at different channel times
    (one of them across a path-gain refresh), two more on a second waveform -- every output against the oracle's apply for that channel time;
    the device path gains against the host evaluation; channel time advances as with the single-job call."""
    import oracle.cdl as OC
    CM = pkg.communication.channelModels
    fs, t_len = 15.36e6, 4097
    cfg = OC.cdl_config(profile, 3.5e9, tx_size, rx_size, fs)
    nt, nr = int(np.prod(tx_size)), int(np.prod(rx_size))
    rng = np.random.default_rng(7 * nt + nr)
    xs = [np.asfortranarray(rng.standard_normal((t_len, nt)) + 1j * rng.standard_normal((t_len, nt))) for _ in range(2)]
    d_xs = [ctx.to_device(x) for x in xs]
    t0s = [0.0, 1.0 / 640 - 1000 / fs, 0.25, 0.0, 0.5]
    which = [0, 0, 0, 1, 1]
    chans = []
    for t0 in t0s:
        ch = CM.CDLChannel(profile, 300e-9, 3.5e9, tx_size, rx_size, fs)
        ch.time = t0
        chans.append(ch)
    # device path gains == host path gains
    snaps = np.array([0.0, 0.1, 1.0 / 640, 0.3333])
    h_dev = chans[0].path_gains_device(snaps, ctx).numpy().reshape(snaps.size, *chans[0]._static().base.shape[:1], nt, nr)
    h_host = np.stack([chans[0].path_gains(t) for t in snaps])
    assert rel(h_dev, h_host) < 1e-12
    outs = CM.applyCDLBatch(chans, [d_xs[w] for w in which], ctx=ctx)
    for t0, w, ch, o in zip(t0s, which, chans, outs):
        want = OC.apply_cdl(cfg, xs[w], t0)
        assert o.shape == (t_len, nr) and rel(o.numpy(), want) < RTOL, (t0, w)
        assert ch.time == pytest.approx(t0 + t_len / fs)
    # a second batch continues from the advanced channel times
    outs2 = CM.applyCDLBatch(chans[:2], [d_xs[0], d_xs[1]], ctx=ctx)
    assert rel(outs2[0].numpy(), OC.apply_cdl(cfg, xs[0], t0s[0] + t_len / fs)) < RTOL
    assert rel(outs2[1].numpy(), OC.apply_cdl(cfg, xs[1], t0s[1] + t_len / fs)) < RTOL
    # one channel twice in a batch = two consecutive slots of that UE: its time advances from job to job
    t_now = chans[2].time
    outs3 = CM.applyCDLBatch([chans[2], chans[2]], [d_xs[0], d_xs[1]], ctx=ctx)
    assert rel(outs3[0].numpy(), OC.apply_cdl(cfg, xs[0], t_now)) < RTOL and rel(outs3[1].numpy(), OC.apply_cdl(cfg, xs[1], t_now + t_len / fs)) < RTOL
    assert chans[2].time == pytest.approx(t_now + 2 * t_len / fs)


_CDL_FIR_SNIPPET = """
import hashlib, importlib, sys
import numpy as np
sys.path.insert(0, %r)
pkg = importlib.import_module(%r)
ctx = pkg._lib.Context(0)
CM = pkg.communication.channelModels
fs, t_len = 15.36e6, 5003
for profile, tx, rx in (("CDL-D", (4, 8, 2, 1, 1), (1, 1, 2, 1, 1)), ("CDL-A", (2, 4, 2, 1, 1), (1, 1, 2, 1, 1)), ("CDL-A", (1, 1, 2, 1, 1), (2, 4, 2, 1, 1))):
    nt = int(np.prod(tx))
    rng = np.random.default_rng(nt)
    xs = [ctx.to_device(np.asfortranarray(rng.standard_normal((t_len, nt)) + 1j * rng.standard_normal((t_len, nt)))) for _ in range(2)]
    chans = [CM.CDLChannel(profile, 300e-9, 3.5e9, tx, rx, fs, Seed=70 + u) for u in range(4)]
    chans[3].time = 1.0 / 640 - 2000 / fs                # one job crosses a path-gain refresh: two segments with their own output rows
    outs = CM.applyCDLBatch(chans, [xs[0], xs[0], xs[1], xs[1]], ctx=ctx)
    print("digest", profile, nt, hashlib.sha256(b"".join(o.numpy().tobytes() for o in outs)).hexdigest())
"""


def test_cdl_delay_filter_forms_same_bits():
    """cdl_fir4_kernel (four outputs per thread, window samples in registers; the model's 16-tap filters) against cdl_fir_kernel (ISAC_CDL_FIR1, read once per
    process, hence the subprocesses): downlink contract-then-filter and uplink filter-then-contract, ragged lengths, a job split at a path-gain refresh -- the
    additions of every output run in the same order: the same bits."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for extra in ({"ISAC_CDL_UNFUSED": "1"}, {"ISAC_CDL_UNFUSED": "1", "ISAC_CDL_FIR1": "1"}):
        r = subprocess.run([sys.executable, "-c", _CDL_FIR_SNIPPET % (root, PKG_NAME)], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        out.append([ln for ln in r.stdout.splitlines() if ln.startswith("digest")])
    assert len(out[0]) == 3 and out[0] == out[1]


_CDL_FUSED_SNIPPET = """
import hashlib, importlib, sys
import numpy as np
sys.path.insert(0, %r)
pkg = importlib.import_module(%r)
ctx = pkg._lib.Context(0)
CM = pkg.communication.channelModels
fs = 122.88e6
UE = (1, 1, 2, 1, 1)
for profile, tx, rx, t_len in (("CDL-A", (4, 8, 2, 1, 1), UE, 20011), ("CDL-D", (1, 8, 2, 1, 1), UE, 9000), ("CDL-A", (1, 3, 2, 1, 1), UE, 1531),
                               ("CDL-A", UE, (4, 8, 2, 1, 1), 9001), ("CDL-D", UE, (1, 12, 2, 1, 1), 4000)):        # (the last two: uplink, 2 -> 64 and 2 -> 24 elements)
    nt = int(np.prod(tx)) + 100 * int(np.prod(rx))
    rng = np.random.default_rng(nt)
    xs = [ctx.to_device(np.asfortranarray(rng.standard_normal((t_len, int(np.prod(tx)))) + 1j * rng.standard_normal((t_len, int(np.prod(tx)))))) for _ in range(2)]
    def chans():
        c = [CM.CDLChannel(profile, 300e-9, 3.5e9, tx, rx, fs, Seed=70 + u) for u in range(4)]
        c[3].time = 1.0 / 640 - (t_len // 2) / fs            # one job crosses a path-gain refresh: two segments, the second starts inside the waveform
        return c
    outs = CM.applyCDLBatch(chans(), [xs[0], xs[0], xs[1], xs[1]], ctx=ctx)
    print("digest batch", profile, nt, hashlib.sha256(b"".join(o.numpy().tobytes() for o in outs)).hexdigest())
    single = [CM.applyCDLBatch([c], [x], ctx=ctx)[0] for c, x in zip(chans(), [xs[0], xs[0], xs[1], xs[1]])]
    print("digest single", profile, nt, hashlib.sha256(b"".join(o.numpy().tobytes() for o in single)).hexdigest())
    np.save(sys.argv[1] + "_%%s_%%d.npy" %% (profile, nt), np.stack([o.numpy() for o in outs]))
"""


def test_cdl_fused_apply_bits_do_not_depend_on_the_grid(tmp_path):
    """cdl_fused_kernel (downlink: contraction + delay filters in one persistent launch, Z never in HBM) and cdl_fused_ul_kernel (uplink: filters + contraction, the
    filtered signals never in HBM): the same bits whether the tile sequence is walked by
    1, 7 or one-per-CU workgroups (ranges that start inside a segment re-create their predecessor's partial sums with warm-up tiles), the same bits for a job
    alone or in a batch; against the unfused kernels (other summation order) <= 1e-12 relative.  122.88 MHz sampling: delays up to 355 samples (three tiles back)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = {}
    td = {"ISAC_CDL_TIME_DOMAIN": "1"}            # (round 6: long downlink waveforms take the overlap-save path by default -- this test is about the time-domain kernels; "os" = the default)
    for tag, extra in (("cu", td), ("one", dict(td, ISAC_CDL_FUSED_WGS="1")), ("seven", dict(td, ISAC_CDL_FUSED_WGS="7")), ("unfused", dict(td, ISAC_CDL_UNFUSED="1")), ("os", {})):
        r = subprocess.run([sys.executable, "-c", _CDL_FUSED_SNIPPET % (root, PKG_NAME), str(tmp_path / tag)], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        runs[tag] = [ln.split()[1:] for ln in r.stdout.splitlines() if ln.startswith("digest")]
    assert len(runs["cu"]) == 10
    assert runs["cu"] == runs["one"] == runs["seven"]
    for i in range(0, 10, 2):
        assert runs["cu"][i][1:] == runs["cu"][i + 1][1:], runs["cu"][i]                  # batch == single
    for f in sorted(os.listdir(tmp_path)):
        if f.startswith("cu_"):
            a, b = np.load(tmp_path / f), np.load(tmp_path / f.replace("cu_", "unfused_"))
            assert rel(a, b) < 1e-12, f
            c = np.load(tmp_path / f.replace("cu_", "os_"))          # frequency-domain overlap-save apply (cdl_os.hip) where the shape qualifies, the same kernels elsewhere
            assert rel(c, a) < 1e-12, f
    assert runs["os"][0] != runs["cu"][0] and runs["os"][2] != runs["cu"][2]      # 64 and 16 transmit elements, T >= two windows: really another code path
    assert runs["os"][4:6] == runs["cu"][4:6] and runs["os"][8:] == runs["cu"][8:]  # 6 transmit elements / a 4 000-sample uplink (less than two windows): the time-domain kernels either way
    assert runs["os"][6] != runs["cu"][6]                                          # uplink 2 -> 64, T = 9 001: the overlap-save uplink kernel (cdl_os_ul_kernel)
    for i in range(0, 10, 2):
        assert runs["os"][i][1:] == runs["os"][i + 1][1:]                          # overlap-save: a job alone == the job in a batch, bit for bit


# ------------------------------------------------------------------ SINR -> CQI
@pytest.mark.parametrize("nr,p,nl", [(2, 4, 1), (2, 4, 2), (4, 8, 4), (8, 32, 8)])
def test_precoded_sinr_and_cqi(pkg, ctx, nr, p, nl):
    import oracle.cqi as OQ
    rng = np.random.default_rng(nr * 100 + p + nl)
    n_re = 52 * 12
    h = np.asfortranarray((rng.standard_normal((n_re, nr, p)) + 1j * rng.standard_normal((n_re, nr, p))) * 3.0)
    w, _ = np.linalg.qr(rng.standard_normal((p, nl)) + 1j * rng.standard_normal((p, nl)))
    w = w / np.sqrt(nl)
    sigma = 0.7
    want = OQ.precoded_sinr_batch(h, sigma, w)
    got = pkg.communication.phyLayer.precodedSINR(h, sigma, w)
    assert got.shape == (n_re,) and np.abs(got - want).max() <= 1e-10 * np.abs(want).max()
    assert pkg.communication.phyLayer.precodedSINR(h[0], sigma, w) == pytest.approx(OQ.precoded_sinr(h[0], sigma, w), rel=1e-10)
    for table in (OQ.DOWNLINK_SINR90PC, OQ.UPLINK_SINR90PC):
        cqi, mean = pkg.communication.phyLayer.cqiFromChannel(h, sigma, w, table)
        assert mean == pytest.approx(want.mean(), rel=1e-12) and cqi == OQ.get_cqi(want.mean(), table)
    assert pkg.communication.phyLayer.getCQI(1e-3, OQ.DOWNLINK_SINR90PC) == 0 == OQ.get_cqi(1e-3, OQ.DOWNLINK_SINR90PC)
    assert pkg.communication.phyLayer.getCQI(10 ** 3.6, OQ.DOWNLINK_SINR90PC) == 15


# ------------------------------------------------------------------ music2D
@pytest.mark.parametrize("n_ants,targets,vel", [(8, ((150.0, 40.0, 1.5),), (0.0,)), (16, ((120.0, 60.0, 1.5), (-250.0, 80.0, 1.5)), (10.0, -6.0))])
def test_music2d_matches_oracle(pkg, ctx, n_ants, targets, vel):
    from types import SimpleNamespace
    sc = make_scene(n_ants=n_ants, n_slots=2, nrb=24, targets=targets, velocity=vel, num_slots_param=3, zero_s_slots=False, seed=4)
    rx = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, sc.noise, nfft=sc.wave.Nfft)
    want, dbg = O.music2d(sc.rp, 30, rx, sc.tx_grid, return_debug=True)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    got = pkg.sensing.estimation.music2D(rp, SimpleNamespace(scs=30), rx, sc.tx_grid)
    assert got.L == dbg.L
    assert np.array_equal(got.aziEst, want.aziEst)
    assert np.array_equal(got.rngEst, want.rngEst) and np.array_equal(got.velEst, want.velEst)


def test_digital_and_mvdr_beamforming(pkg, ctx):
    sc = make_scene(n_ants=16, n_slots=1, nrb=24, with_noise=False)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    m = np.arange(16)
    a1 = np.exp(-2j * np.pi * m * 0.5 * float(O.sind(15)))
    a2 = np.exp(-2j * np.pi * m * 0.5 * float(O.sind(-40)))
    ra = 4 * np.outer(a1, a1.conj()) + np.outer(a2, a2.conj()) + np.diag(np.random.default_rng(3).uniform(0.01, 0.03, 16))
    for l in (1, 2, 4):
        wa, _ = O.digital_bf(l, sc.rp, ra)
        ga, ge = pkg.sensing.estimation.doaEstimation.digitalBF(l, rp, ra)
        assert np.array_equal(ga, wa) and np.isnan(ge).all()
        wa, _ = O.mvdr_bf(l, sc.rp, ra)
        ga, _ = pkg.sensing.estimation.doaEstimation.mvdrBF(l, rp, ra)
        assert np.array_equal(ga, wa)


# ------------------------------------------------------------------ edge cases the reference's parameter space allows
def test_many_targets_and_single_antenna(pkg, ctx):
    """11 LoS targets exercise the 8 + 2 + 1 beam-sum tiling; a 1-antenna array exercises the NB = 1 MFMA plan."""
    rng = np.random.default_rng(17)
    tg = tuple((float(r * np.cos(a)), float(r * np.sin(a)), 1.5) for r, a in zip(rng.uniform(60, 400, 11), rng.uniform(-1, 1, 11)))
    vel = tuple(float(v) for v in rng.integers(-10, 11, 11))
    los = np.ones(11, dtype=int); los[4] = 0
    sc = make_scene(n_ants=3, n_slots=1, nrb=24, targets=tg, velocity=vel, seed=8)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    want = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, los, sc.noise, nfft=sc.wave.Nfft)
    got = pkg.sensing.monoStaticSensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, rp, los, noise=sc.noise, nfft=sc.wave.Nfft)
    assert rel(got, want) < RTOL
    sc1 = make_scene(n_ants=1, n_slots=4, nrb=24, targets=((150.0, 40.0, 1.5),), velocity=(0.0,), num_slots_param=6, zero_s_slots=False)
    _run_fft2d_case(pkg, sc1)


def test_doppler_truncation_when_more_symbols_than_nfft(pkg, ctx):
    """fft(., nFFT, 2) truncates when L > nFFT (fft2D.m:46): 56 symbols against nFFT = 32."""
    sc = make_scene(n_ants=2, n_slots=4, nrb=24, targets=((150.0, 40.0, 1.5),), velocity=(0.0,), num_slots_param=3, zero_s_slots=False)
    assert sc.rp.nFFT == 32 and sc.L == 56
    _run_fft2d_case(pkg, sc)


@pytest.mark.parametrize("nrb,nfft", [(133, 2048), (51, 1024)])
def test_other_numerologies_stockham_path(pkg, ctx, nrb, nfft):
    sc = make_scene(n_ants=2, n_slots=2, nrb=nrb, targets=((120.0, 30.0, 1.5),), velocity=(3.0,), num_slots_param=3, zero_s_slots=False)
    assert sc.wave.Nfft == nfft
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    want = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, sc.noise, nfft=nfft)
    got = pkg.sensing.monoStaticSensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, noise=sc.noise, nfft=nfft)
    assert rel(got, want) < RTOL
    _run_fft2d_case(pkg, sc)


def test_full_size_round_trip_properties(pkg, ctx):
    """Size-independent properties at the benchmark's full shape (K=3276, L=224, A=8 here to bound host memory):
    modulate -> demodulate is the identity; the device generator's grid is unit-modulus QPSK with zeroed S slots;
    covariance is Hermitian PSD with trace = mean power."""
    K, L, A = 3276, 224, 8
    car = pkg._lib.Carrier(K, 4096, 30, 0)
    d_grid, d_wave, d_back = ctx.empty((K, L, A)), ctx.empty((983040, A)), ctx.empty((K, L, A))
    ctx.check(ctx.lib.isac_synth_qpsk_grid_dev(ctx.handle, C.c_void_p(d_grid.ptr), K, L, A, C.c_uint64(5), 1))
    ctx.check(ctx.lib.isac_ofdm_modulate_dev(ctx.handle, C.c_void_p(d_grid.ptr), L, A, C.byref(car), C.c_double(2.5), C.c_void_p(d_wave.ptr), C.c_int64(983040)))
    ctx.check(ctx.lib.isac_ofdm_demodulate_dev(ctx.handle, C.c_void_p(d_wave.ptr), C.c_int64(983040), A, C.byref(car), C.c_void_p(d_back.ptr), L))
    g, b = d_grid.numpy(), d_back.numpy()
    assert rel(b / 2.5, g) < 1e-11
    mod = np.abs(g)
    s_slots = np.zeros(L, bool)
    for s in range(3, 16, 4):
        s_slots[14 * s:14 * (s + 1)] = True
    assert np.all(mod[:, s_slots, :] == 0) and np.allclose(mod[:, ~s_slots, :], 1.0, atol=1e-15)
    d_ra = ctx.empty((A, A))
    ctx.check(ctx.lib.isac_covariance_dev(ctx.handle, C.c_void_p(d_back.ptr), C.c_int64(K * L), C.c_int32(A), C.c_void_p(d_ra.ptr)))
    ra = d_ra.numpy()
    assert np.array_equal(ra, ra.conj().T) and np.linalg.eigvalsh(ra).min() > -1e-9
    assert np.trace(ra).real == pytest.approx((np.abs(b) ** 2).sum() / (K * L), rel=1e-12)


def test_eigh_nan_input_is_an_error_not_a_hang(pkg, ctx):
    """A NaN matrix never converges: the QL recurrence must stop at its sweep budget and the call must fail loudly."""
    a = 100
    h = np.full((a, a), np.nan, dtype=np.complex128, order="F")
    w = np.zeros(a)
    with pytest.raises(pkg.IsacError) as ei:
        ctx.check(ctx.lib.isac_eigh(ctx.handle, h.ctypes.data_as(C.c_void_p), C.c_int32(a), w.ctypes.data_as(C.c_void_p), None))
    assert ei.value.name == "HIP" and "eigensolver" in str(ei.value)
