"""Known-answer tests that pin the CPU oracle.

The reference ships no tests or golden vectors and cannot run here (MATLAB), so
these closed-form checks are what anchors the restatement ("parity unpinned")."""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import pytest

import oracle as O
from conftest import make_scene


def _full_rp(n_ants=16, **kw):
    cell = O.default_cell_params(n_ants=n_ants, **kw)
    ci = SimpleNamespace(NRBsDL=273, SubcarrierSpacing=30)
    return O.radar_params(cell, ci, O.nr_ofdm_info(273, 30)), ci


def test_philox_random123_kat():
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, out in kat:
        got = tuple(int(v) for v in O.philox4x32_10(*ctr, *key))
        assert got == out


def test_kat1_cfar_constant():
    assert O.cfar_threshold_factor(24, 1e-9) == pytest.approx(24 * (10 ** (9 / 24) - 1), rel=1e-15)
    assert O.cfar_threshold_factor(24, 1e-9) == pytest.approx(32.91296893587972, rel=1e-14)


def test_kat2_default_scenario_constants():
    rp, _ = _full_rp()
    assert rp.nIFFT == 4096 and rp.nFFT == 256
    assert rp.rRes == pytest.approx(1.2198586344401041, rel=1e-15)
    assert rp.vRes == pytest.approx(4.5621831158455395, rel=1e-14)
    assert rp.Tsri == pytest.approx(3.6669921875e-05, rel=1e-15)
    assert rp.N0 == pytest.approx(1.958675465085905e-12, rel=1e-14)
    cf = O.cfar2d_config(rp)
    assert cf.rowRange == (42, 411) and cf.colRange == (118, 140)
    assert cf.CUTIdx.shape == (2, 8510)
    # rows fastest (cfar2D.m:23-24)
    assert cf.CUTIdx[:, 0].tolist() == [42, 118] and cf.CUTIdx[:, 1].tolist() == [43, 118]
    assert cf.CUTIdx[:, -1].tolist() == [411, 140]


def test_sind_cosd_exact_points_and_mirror():
    assert O.sind(0) == 0 and O.sind(180) == 0 and O.sind(-180) == 0 and O.sind(360) == 0
    assert O.sind(90) == 1 and O.sind(-90) == -1 and O.sind(270) == -1
    assert O.cosd(90) == 0 and O.cosd(270) == 0 and O.cosd(0) == 1 and O.cosd(180) == -1
    ang = np.arange(-180, 181)
    pos = ang[ang >= 0]
    assert np.array_equal(O.sind(pos), O.sind(180 - pos))          # bit-for-bit mirror (A.6)
    neg = ang[ang < 0]
    assert np.array_equal(O.sind(neg), O.sind(-180 - neg))
    assert np.allclose(O.sind(ang), np.sin(np.deg2rad(ang)), atol=1e-15)


def test_kaiser_matches_definition():
    for n in (8, 9, 3276, 4096):
        w = O.kaiser(n, 3.0)
        assert w.shape == (n,) and np.array_equal(w, w[::-1])
        assert np.allclose(w, np.kaiser(n, 3.0), rtol=1e-13, atol=0)


def test_findpeaks_semantics():
    y = np.array([0, 1, 0, 2, 2, 1, 3, 0, 3, 1, 5])
    pk, loc = O.findpeaks(y)
    # plateau -> first sample; endpoints never; stable descending order for the tie 3,3
    assert loc.tolist() == [6, 8, 3, 1] and pk.tolist() == [3, 3, 2, 1]
    pk, loc = O.findpeaks(y, npeaks=2)
    assert loc.tolist() == [6, 8]
    with pytest.raises(ValueError):
        O.findpeaks(y, npeaks=0)
    assert O.findpeaks(np.ones(10))[1].size == 0


def test_unique_stable():
    assert O.unique_stable(np.array([3.0, 1.0, 3.0, 2.0, 1.0])).tolist() == [3.0, 1.0, 2.0]


def test_ofdm_roundtrip_and_cp_structure():
    assert O.cp_lengths(4096, 30, 15).tolist() == [352] + [288] * 13 + [352]
    starts, cps = O.symbol_starts(4096, 30, 28)
    assert starts[14] == 61440 and (starts[-1] + cps[-1] + 4096) == 2 * 61440
    rng = np.random.default_rng(0)
    g = rng.standard_normal((288, 28, 2)) + 1j * rng.standard_normal((288, 28, 2))
    w = O.ofdm_modulate(g, 512, 30)
    assert w.shape[0] == 2 * 7680
    assert np.abs(O.ofdm_demodulate(w, 288, 512, 30) - g).max() < 1e-12
    # a trailing partial symbol is ignored (floor of whole symbols)
    assert O.ofdm_demodulate(w[:-5], 288, 512, 30).shape[1] == 27


def test_kat4_shift_algebra_literal_equals_explicit():
    rng = np.random.default_rng(2)
    for (k, l, a, ni, nf) in [(48, 14, 3, 64, 16), (60, 28, 2, 64, 32), (48, 40, 1, 64, 32)]:
        rx = rng.standard_normal((k, l, a)) + 1j * rng.standard_normal((k, l, a))
        tx = rng.standard_normal((k, l, a)) + 1j * rng.standard_normal((k, l, a))
        assert np.abs(O.rdm_literal(rx, tx, ni, nf) - O.rdm_explicit(rx, tx, ni, nf)).max() == 0.0


def test_kat3_range_bin_of_noiseless_target():
    sc = make_scene(n_ants=2, n_slots=2, targets=((100.0, 0.0, 30.0),), velocity=(0.0,), with_noise=False,
                    zero_s_slots=False)
    assert sc.rp.range[0] == pytest.approx(100.0)
    shift = int(np.ceil(2 * 100.0 * sc.rp.fs / O.LIGHTSPEED))
    assert shift == 82
    rx = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, None)
    rdm = O.rdm_literal(rx, sc.tx_grid, 4096, 256)
    p = np.abs(rdm[:, :, 0]) ** 2
    r, c = np.unravel_index(np.argmax(p), p.shape)
    assert (r + 1, c + 1) == (shift + 1, 129)
    assert (r * sc.rp.rRes) == pytest.approx(100.0284, abs=1e-3)


def test_cfar_brute_force_and_edges():
    rng = np.random.default_rng(3)
    p = rng.exponential(1.0, (40, 30))
    p[20, 15] = 500.0
    p[10, 8] = 90.0
    rows, cols = np.arange(4, 37), np.arange(4, 27)
    cc, rr = np.meshgrid(cols, rows)
    cut = np.stack([rr.ravel(order="F"), cc.ravel(order="F")])
    det = O.ca_cfar2d(p, cut, 1e-6)
    alpha = O.cfar_threshold_factor(24, 1e-6)
    want = []
    for r1, c1 in cut.T:
        r, c = r1 - 1, c1 - 1
        blk = p[r - 3:r + 4, c - 3:c + 4].copy()
        tot = 0.0
        for dc in range(7):
            for dr in range(7):
                if 1 <= dr <= 5 and 1 <= dc <= 5:
                    continue
                tot = tot + blk[dr, dc]
        if p[r, c] > alpha * (tot / 24):
            want.append((r1, c1))
    assert [tuple(x) for x in det.T.tolist()] == want and (21, 16) in want
    with pytest.raises(ValueError):
        O.ca_cfar2d(p, np.array([[3], [10]]), 1e-6)          # training window leaves the matrix
    assert O.ca_cfar2d(np.zeros((20, 20)), np.array([[10], [10]]), 1e-6).shape == (2, 0)   # 0 > 0 is false (B8)


def test_covariance_is_conjugate_transpose_form():
    rng = np.random.default_rng(4)
    g = rng.standard_normal((6, 5, 3)) + 1j * rng.standard_normal((6, 5, 3))
    ra = O.covariance(g)
    x = g.reshape(30, 3, order="F").conj().T          # reshape(rxGrid, nSc*nSym, nAnts)'
    assert np.allclose(ra, x @ x.conj().T / 30, atol=1e-15)
    assert np.array_equal(ra, ra.conj().T)


def test_kat5_music_rank1_and_mirror_ties():
    rp, _ = _full_rp(n_ants=16)
    m = np.arange(16)
    for phi0, want1 in [(20, 20.0), (37, 37.0), (-30, -150.0), (-61, -119.0)]:
        a = np.exp(-2j * np.pi * m * 0.5 * float(O.sind(phi0)))
        ra = np.outer(a, a.conj()) + 1e-3 * np.eye(16)
        l, azi, ele = O.music_doa(1, rp, ra)
        assert l == 1 and azi.tolist() == [want1] and np.isnan(ele).all()
        _, azi2, _ = O.music_doa(2, rp, ra)
        mirror = 180.0 - phi0 if phi0 > 0 else -180.0 - phi0
        assert sorted(azi2.tolist()) == sorted([float(phi0), mirror])
        assert azi2.tolist() == sorted(azi2.tolist())          # equal heights -> index order
    with pytest.raises(ValueError):
        O.music_doa(0, rp, np.eye(16))                         # NPeaks = 0 (B13)
    _, azi, _ = O.music_doa(16, rp, np.eye(16, dtype=complex)) # empty noise subspace -> flat spectrum -> no peaks
    assert azi.size == 0


def test_kat6_determine_num_targets():
    # ascending eigenvalues as eig() returns them (music.m:22 is called before the sort)
    v = np.array([1.0, 1.1, 1.2, 1.3, 50.0, 80.0])
    delta = -np.diff(v)
    half = np.mean(delta[int(np.ceil((delta.size + 1) / 2)) - 1:])
    assert O.determine_num_targets(v) == int(np.argmax(delta - 2 * half)) + 1
    assert O.determine_num_targets(np.array([1.0, 2.0, 3.0])) in (1, 2)


def test_all_nlos_is_an_error():
    sc = make_scene(n_ants=2, n_slots=1, nrb=24, with_noise=False)
    with pytest.raises(ValueError):
        O.basic_radar_channel(sc.tx_wave, sc.rp, np.zeros(1), None)


def test_fft2d_end_to_end_small_and_zero_detection_error():
    sc = make_scene(n_ants=4, n_slots=4, nrb=24, targets=((150.0, 40.0, 1.5),), velocity=(0.0,),
                    num_slots_param=6, zero_s_slots=False)
    rx = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, sc.noise, nfft=sc.wave.Nfft)
    cf = O.cfar2d_config(sc.rp)
    est, dbg = O.fft2d(sc.rp, cf, rx, sc.tx_grid, return_debug=True)
    shift = int(np.ceil(2 * sc.rp.range[0] * sc.rp.fs / O.LIGHTSPEED))
    assert est.rngEst[0] == pytest.approx(shift * sc.rp.rRes)
    assert est.velEst[0] == 0.0
    assert all(d.shape[1] >= 1 for d in dbg.detections)
    # estimates are exact multiples of the resolutions
    assert np.allclose(est.rngEst / sc.rp.rRes, np.round(est.rngEst / sc.rp.rRes), atol=1e-9)
    # a zero rxGrid gives no detections -> findpeaks NPeaks=0 error (-> senResults = NaN in the reference)
    with pytest.raises(ValueError):
        O.fft2d(sc.rp, cf, np.zeros_like(rx), sc.tx_grid)


def test_music2d_small():
    sc = make_scene(n_ants=4, n_slots=2, nrb=24, targets=((150.0, 40.0, 1.5),), velocity=(0.0,),
                    num_slots_param=3, zero_s_slots=False)
    rx = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, sc.noise, nfft=sc.wave.Nfft)
    est, dbg = O.music2d(sc.rp, 30, rx, sc.tx_grid, return_debug=True)
    assert dbg.L >= 1 and est.rngEst.size <= dbg.L
    assert np.array_equal(dbg.Rr, dbg.Rr.conj().T)


def test_toolbox_primitives_against_independent_implementations():
    """The oracle's restatements of toolbox primitives against SciPy / NumPy / pandas implementations of the same
    published definitions (an independent check of the restatement, not of MATLAB itself)."""
    import pandas as pd
    from scipy import linalg, signal
    from scipy import special
    rng = np.random.default_rng(11)
    # kaiser(N, beta): I0(beta sqrt(1 - ((n - (N-1)/2) / ((N-1)/2))^2)) / I0(beta)
    for n in (2, 3, 8, 33, 256, 3276, 4096):
        assert np.abs(O.kaiser(n, 3.0) - signal.windows.kaiser(n, 3.0, sym=True)).max() < 2e-15    # a few ulp of I0
    assert O.kaiser(1, 3.0).tolist() == [1.0]
    # findpeaks(y, 'SortStr', 'descend'): strict interior local maxima, plateaus report their first sample
    for _ in range(20):
        y = np.round(rng.standard_normal(400), 1)            # coarse values: plenty of plateaus and ties
        pks, locs = O.findpeaks(y)
        ref, props = signal.find_peaks(y, plateau_size=1)
        ref_first = props["left_edges"]
        assert sorted(locs.tolist()) == ref_first.tolist()
        assert np.all(np.diff(pks) <= 0)
        for v in np.unique(pks):                             # stable sort: equal peaks keep index order
            assert np.all(np.diff(locs[pks == v]) > 0)
    # unique(x, 'stable')
    x = rng.integers(0, 20, 300).astype(float)
    assert np.array_equal(O.unique_stable(x), pd.unique(x))
    # sind / cosd away from the exact points
    a = rng.uniform(-720, 720, 1000)
    assert np.abs(O.sind(a) - np.sin(np.deg2rad(a))).max() < 2e-15 and np.abs(O.cosd(a) - np.cos(np.deg2rad(a))).max() < 2e-15
    # eig-based pieces: covariance is Hermitian PSD, MUSIC noise projector is idempotent
    g = rng.standard_normal((64, 7, 5)) + 1j * rng.standard_normal((64, 7, 5))
    ra = O.covariance(g)
    assert np.array_equal(ra, ra.conj().T) and linalg.eigvalsh(ra).min() > -1e-12
    # db helpers
    assert O.db2pow(10.0) == pytest.approx(10.0) and O.db2mag(20.0) == pytest.approx(10.0)
    assert O.pow2db(100.0) == pytest.approx(20.0) and O.mag2db(100.0) == pytest.approx(40.0)
    assert special.i0(3.0) == pytest.approx(4.880792585865024)


def test_ofdm_demodulate_against_a_plain_fft_model():
    """nrOFDMDemodulate as restated (CP fraction 0.5 window, phase compensation, fftshift, central K bins) against a
    direct per-symbol DFT of the CP-stripped samples: they must agree on a CP-OFDM waveform (the half-CP window offset is a
    pure per-bin phase ramp that the demodulator undoes)."""
    rng = np.random.default_rng(12)
    nrb, n_slots, nfft = 24, 2, 512
    k, l = 12 * nrb, 14 * n_slots
    grid = (rng.standard_normal((k, l, 2)) + 1j * rng.standard_normal((k, l, 2)))
    wave = O.ofdm_modulate(grid, nfft, 30)
    back = O.ofdm_demodulate(wave, k, nfft, 30)
    assert np.abs(back - grid).max() < 1e-11
    starts, cps = O.symbol_starts(nfft, 30, l)
    for sym in (0, 1, 13, 14, 27):
        seg = wave[starts[sym] + cps[sym]: starts[sym] + cps[sym] + nfft, 0]
        spec = np.fft.fftshift(np.fft.fft(seg))
        mid = spec[nfft // 2 - k // 2: nfft // 2 + k // 2]
        assert np.abs(mid - grid[:, sym, 0]).max() < 1e-10


def test_spectral_philox_noise_statistics():
    """The spectral noise field (one Philox call per element pair, 32-bit Box-Muller uniforms, float32 transform) is white, circular, unit variance:
    moments, real/imag and pair-half cross-correlations, lag correlations along subcarriers / symbols / antennas."""
    w = O.philox_spectral_noise(3276, 28, 8, 0x5EED0002)
    n = w.size
    tol = 5.0 / np.sqrt(n)
    assert abs(w.real.mean()) < tol and abs(w.imag.mean()) < tol
    assert abs(w.real.var() - 1) < 3 * tol and abs(w.imag.var() - 1) < 3 * tol
    assert abs((w.real * w.imag).mean()) < tol
    assert abs((np.abs(w) ** 4).mean() / 8.0 - 1.0) < 0.02               # |z|^2 = 2 Exp(1)  ->  E|z|^4 = 8
    assert abs((w.real ** 4).mean() / 3.0 - 1.0) < 0.02                  # Gaussian kurtosis
    for ax in range(3):
        a = np.moveaxis(w, ax, 0)
        assert abs((a[1:] * np.conj(a[:-1])).mean()) < 2 * tol
    # the two halves of one Philox call (elements k and k + 512) are uncorrelated
    assert abs((w[0:512] * np.conj(w[512:1024])).mean()) < 10 / np.sqrt(512 * 28 * 8)
    assert abs((np.abs(w[0:512]) ** 2 * np.abs(w[512:1024]) ** 2).mean() / 4.0 - 1.0) < 0.05
    # tail: the radius never exceeds the bound of u >= 2^-33, sqrt(2 * 33 ln 2) = 6.76 (float32 evaluation: a few ulps of slack)
    assert np.abs(w).max() <= O.philox.SPECTRAL_NOISE_MAX + 1e-5
    assert np.array_equal(w, O.philox_spectral_noise(3276, 28, 8, 0x5EED0002)) and not np.array_equal(w, O.philox_spectral_noise(3276, 28, 8, 1))
