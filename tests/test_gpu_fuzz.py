"""Seeded differential fuzz of the whole chain: random small scenes (numerology, antennas, slots, targets, Doppler, LoS
pattern, noise) through monoStaticSensing -> fft2D on the GPU against the oracle.  Same tolerances as test_gpu_parity.py:
fields <= 1e-10 relative, CFAR detections / range-velocity bins / integer azimuths exact.  Scenes in which some CUT lies
within 1e-9 (relative) of its CFAR threshold are skipped -- there a rounding-level difference may legitimately flip a
detection.  No scene is skipped for its azimuth list any more: see _check_azimuth (tally printed at the end of the run).  ISAC_FUZZ_N=<n> runs more seeds, ISAC_FUZZ_SEED0=<s> starts every test's window at seed s."""
from __future__ import annotations

import os

import numpy as np
import pytest

import oracle as O
from conftest import load_pkg, make_scene
from test_gpu_parity import RTOL, _guard_band_ok, rel

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("ISAC_FUZZ_N", "16"))
SEED0 = int(os.environ.get("ISAC_FUZZ_SEED0", "0"))       # first seed of every test's window: a campaign can walk fresh windows (profiles/r05_fuzz_campaigns.txt)


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


# antenna counts that reach every covariance / eigensolver kernel: cov_mfma_small (<= 32), cov_mfma_lds<3> (33-48), <4> (49-64: the bench's own),
# cov_mfma_block_pl + eigh_tridiag_dist (65-256); scenes with them stay small (24-51 PRB, 2-4 slots) so that the oracle stays cheap
WIDE_ANTS = (64, 200, 56, 96, 33, 256, 48, 130, 16, 65)


def _scene(seed, wide=None):
    rng = np.random.default_rng(1000 + seed)
    nrb = int(rng.choice([24, 51, 106, 133, 273]))
    n_ants = int(rng.choice([1, 2, 3, 4, 5, 8]))
    n_slots = int(rng.choice([2, 3, 4, 6]))
    if wide is None:
        wide = seed % 4 >= 2                 # half of the seeds are wide-array scenes; the default 16 cover 64, 200, 56, 96, 33, 256, 48, 130 antennas
    if wide:
        nrb, n_slots = int(rng.choice([24, 51])), int(rng.choice([2, 3, 4]))
        n_ants = int(WIDE_ANTS[(2 * (seed // 4) + seed % 4 - 2) % len(WIDE_ANTS)])
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


DEGENERATE_GAP = 1e-12      # relative eigenvalue gap below which MUSIC's signal / noise split is decided by rounding (see _check_azimuth)


def _fold90(a):
    a = np.asarray(a, dtype=np.float64)
    return np.where(a == -90.0, 90.0, a)


def _check_azimuth(pkg, record_property, sc, rp, got, want, ra_dev, ra_ref):
    """MUSIC separates the L = numDets largest eigenvalues from the rest (music.m:21-25).  A perturbation of the size of fp64 rounding (eps w0) rotates the split subspace by
    ~ eps w0 / gap: with a relative gap >= 1e-12 that is <= 2e-4 rad -- invisible on the 1-degree scan -- and the chain's azimuth list is compared as it is ("chain").
    Below that the split sits inside a numerically degenerate cluster: noise-free scenes in which CFAR reports more range bins than there are sources, so that the numDets-th
    and (numDets + 1)-th eigenvalues are both rounding residue (~1e-17 w0) and WHICH vectors count as signal is decided by rounding, in MATLAB too.  Such a scene is no
    longer skipped: every other field has been compared already, and the MUSIC stage itself is then checked on the scene's own covariance at its
    NUMERICAL rank -- doaEstimation.music(L_eff, ., Ra), L_eff = eigenvalues above 1e-9 w0 -- where the split is well defined ("stage_at_rank").  The tally of both kinds is
    printed at the end of the run (conftest.pytest_terminal_summary)."""
    w = np.sort(np.linalg.eigvalsh(ra_ref))[::-1]
    n_sig = min(int(want.rngEst.size), sc.A - 1)
    if n_sig < 1 or (w[n_sig - 1] - w[n_sig]) >= DEGENERATE_GAP * w[0]:
        if not np.array_equal(got.aziEst, want.aziEst):
            # +90 and -90 degrees are ONE steering vector at half-wavelength spacing (exp(-j pi m sind(+-90)) = (-1)^m): the two scan points carry the same spectrum value up
            # to the rounding of their sincos arguments, so which of them findpeaks' descending sort lists first is rounding-defined (seed 1062: [90, -90] vs [-90, 90]).
            # Everything else about the list must agree.
            assert np.array_equal(_fold90(got.aziEst), _fold90(want.aziEst)) and np.array_equal(np.sort(got.aziEst), np.sort(want.aziEst)), (got.aziEst, want.aziEst)
            record_property("azimuth", "chain_pm90_tie")
            return
        record_property("azimuth", "chain")
        return
    l_eff = min(int((w > 1e-9 * w[0]).sum()), sc.A - 1)
    while l_eff >= 1 and (w[l_eff - 1] - w[l_eff]) < 1e-9 * w[0]:         # (two equal LARGE eigenvalues: step below the pair)
        l_eff -= 1
    if l_eff < 1:                                                          # a one-element array, or no separable eigenvalue at all: nothing MUSIC could be asked
        record_property("azimuth", "undefined")
        return
    a = O.music_doa(l_eff, sc.rp, ra_ref)
    b = pkg.sensing.estimation.doaEstimation.music(l_eff, rp, ra_dev)
    if b[0] == a[0] and not np.array_equal(b[1], a[1]):                    # the +-90 degree pair again (one steering vector, see above): order rounding-defined
        assert np.array_equal(_fold90(b[1]), _fold90(a[1])) and np.array_equal(np.sort(b[1]), np.sort(a[1])), (b[1], a[1])
        record_property("azimuth", "stage_at_rank_pm90_tie")
        return
    assert b[0] == a[0] and np.array_equal(b[1], a[1])
    record_property("azimuth", "stage_at_rank")


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + N_CASES))
def test_chain_matches_oracle_on_random_scene(pkg, seed, record_property):
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
    if not rel(gd.power_window, np.abs(dbg.rdm[r0:r0 + nr, c0:c0 + nc, :]) ** 2) < RTOL:
        # self-diagnosis (one such mismatch was seen once in ~1 900 cases under 16 concurrent processes and never reproduced, profiles/r05_fuzz_campaigns.txt):
        # repeat BOTH sides and say which one moved -- a device-side race, a host-side one, or a genuine disagreement
        got2, gd2 = pkg.sensing.estimation.fft2D(rp, cf, ref_echo, sc.tx_grid, return_debug=True)
        _, dbg2 = O.fft2d(sc.rp, ocf, ref_echo, sc.tx_grid, return_debug=True)
        ref1, ref2 = (np.abs(d.rdm[r0:r0 + nr, c0:c0 + nc, :]) ** 2 for d in (dbg, dbg2))
        bad = np.argwhere(np.abs(gd.power_window - ref1) > 1e-8 * ref1.max())
        pytest.fail(f"power window: device call 1 vs oracle call 1 {rel(gd.power_window, ref1):.3e}; device call 2 vs oracle call 1 {rel(gd2.power_window, ref1):.3e}; "
                    f"device 1 vs device 2 {rel(gd.power_window, gd2.power_window):.3e}; oracle 1 vs oracle 2 {rel(ref1, ref2):.3e}; {len(bad)} entries off in call 1, "
                    f"rows {sorted(set(bad[:, 0].tolist()))[:8]} cols {sorted(set(bad[:, 1].tolist()))[:8]} antennas {sorted(set(bad[:, 2].tolist()))}")
    assert rel(gd.Ra, dbg.Ra) < RTOL
    for a in range(sc.A):
        assert np.array_equal(gd.detections[a], dbg.detections[a]), f"antenna {a}"
    assert np.array_equal(got.rngEst, want.rngEst) and np.array_equal(got.velEst, want.velEst)
    _check_azimuth(pkg, record_property, sc, rp, got, want, gd.Ra, dbg.Ra)


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + max(6, N_CASES // 4)))
def test_fused_and_philox_paths_on_random_scene(pkg, seed):
    """Full-size numerology (Nfft = nIFFT = 4096): (a) the fused monoStaticSensing+range path is bit-identical to the
    unfused call sequence for injected, Philox and no noise and 1..6 targets (the compile-time target-count kernels
    1..4 and the run-time one); (b) Philox-mode echo grids match the oracle fed with the restated generator."""
    from oracle.philox import philox_normal_pairs
    rng = np.random.default_rng(7000 + seed)
    q = int(rng.integers(1, 7))
    n_ants = int(rng.choice([1, 2, 3, 5]))
    n_slots = int(rng.choice([2, 3]))
    r = rng.uniform(60.0, 400.0, q)
    az = np.deg2rad(rng.uniform(-70.0, 70.0, q))
    targets = tuple((float(r[i] * np.cos(az[i])), float(r[i] * np.sin(az[i])), 1.5) for i in range(q))
    sc = make_scene(n_ants=n_ants, n_slots=n_slots, nrb=273, targets=targets, velocity=tuple(float(v) for v in rng.integers(-10, 11, q)),
                    seed=seed, zero_s_slots=False)
    ctx = pkg.default_context()
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    d_wave, d_txg = ctx.to_device(sc.tx_wave), ctx.to_device(sc.tx_grid)
    mode = seed % 3
    kw = [dict(noise=ctx.to_device(sc.noise)), dict(seed=0xABCD + seed), dict()][mode]
    e0 = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, **kw)
    e1 = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, fuse_fft2d=(rp, cf, d_txg), **kw)
    g0, g1 = e0.numpy(), e1.numpy()
    assert np.array_equal(g0, g1)
    if mode == 1:                                         # on-device generator vs its NumPy restatement
        t, a = sc.tx_wave.shape
        nz = philox_normal_pairs(np.arange(t * a, dtype=np.uint64), 0xABCD + seed, 0).reshape(a, t).T
        ref = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, nz, nfft=4096)
        assert rel(g0, ref) < RTOL
    try:
        est0, dbg0 = pkg.sensing.estimation.fft2D(rp, cf, e0, d_txg, return_debug=True)
    except pkg.IsacError as e:
        assert e.name == "NO_DETECTION"
        with pytest.raises(pkg.IsacError):
            pkg.sensing.estimation.fft2D(rp, cf, e1, d_txg)
        return
    est1, dbg1 = pkg.sensing.estimation.fft2D(rp, cf, e1, d_txg, return_debug=True)     # consumes the cached range rows
    assert np.array_equal(dbg0.power_window, dbg1.power_window)
    assert all(np.array_equal(x, y) for x, y in zip(dbg0.detections, dbg1.detections))
    assert np.array_equal(est0.rngEst, est1.rngEst) and np.array_equal(est0.velEst, est1.velEst) and np.array_equal(est0.aziEst, est1.aziEst)


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + max(8, N_CASES // 4)))
def test_cdl_apply_on_random_configuration(pkg, seed):
    import oracle.cdl as OC
    rng = np.random.default_rng(9000 + seed)
    profile = ["CDL-A", "CDL-D"][int(rng.integers(0, 2))]
    tx_size = (int(rng.choice([1, 2, 4])), int(rng.choice([1, 2, 3, 4, 8])), 2, 1, 1)
    fs = float(rng.choice([15.36e6, 30.72e6, 122.88e6]))
    t_len = int(rng.integers(200, 9000))
    t0 = float(rng.uniform(0.0, 0.05))
    cfg = OC.cdl_config(profile, 3.5e9, tx_size, (1, 1, 2, 1, 1), fs)
    ch = pkg.communication.channelModels.CDLChannel(profile, 300e-9, 3.5e9, tx_size, (1, 1, 2, 1, 1), fs)
    ch.time = t0
    nt = int(np.prod(tx_size))
    x = np.asfortranarray(rng.standard_normal((t_len, nt)) + 1j * rng.standard_normal((t_len, nt)))
    got = pkg.communication.channelModels.applyCDL(ch, x)
    assert got.shape == (t_len, 2) and rel(got, OC.apply_cdl(cfg, x, t0)) < RTOL


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + max(8, N_CASES // 4)))
def test_cdl_batch_on_random_configuration(pkg, seed):
    """applyCDLBatch (isac_cdl_apply_batch_dev + device path gains) fuzzed in BOTH directions: downlink (gNB array -> 2) and uplink (2 -> gNB array: the filter-first
    kernels), 1-4 UEs x 1-3 consecutive slots in one call (a channel that appears several times advances its time from job to job), UEs with the reference's shared seed or
    their own, downlink slots on one shared waveform or every job on its own, start times drawn next to a path-gain refresh so that jobs hold two gain blocks; sampling
    rates up to 122.88 MHz (delays of several 128-row tiles in the fused kernels).  Every output against the oracle's apply at that job's channel time."""
    import oracle.cdl as OC
    CM = pkg.communication.channelModels
    ctx = pkg.default_context()
    rng = np.random.default_rng(9700 + seed)
    profile = ["CDL-A", "CDL-D"][int(rng.integers(0, 2))]
    uplink = bool(rng.integers(0, 2))
    gnb = (int(rng.choice([1, 2, 4])), int(rng.choice([1, 2, 3, 4, 8])), 2, 1, 1)
    ue = (1, 1, 2, 1, 1)
    tx, rx = (ue, gnb) if uplink else (gnb, ue)
    nt, nr = int(np.prod(tx)), int(np.prod(rx))
    fs = float(rng.choice([15.36e6, 30.72e6, 122.88e6]))
    t_len = int(rng.integers(300, 6000))
    n_ue, n_slots = int(rng.integers(1, 5)), int(rng.integers(1, 4))
    seeds = [73 if rng.integers(0, 2) else 60 + u for u in range(n_ue)]
    refresh = 1.0 / 640
    t_start = []
    for u in range(n_ue):
        kind = int(rng.integers(0, 3))
        if kind == 0:
            t_start.append(float(rng.uniform(0.0, 0.05)))
        else:                                                             # a refresh falls inside slot `hit` of this UE
            hit = int(rng.integers(0, n_slots))
            t_start.append(float(int(rng.integers(1, 30)) * refresh - (hit * t_len + int(rng.integers(1, t_len))) / fs))
    t_start = [t if t >= 0 else t + 40 * refresh for t in t_start]
    chans = [CM.CDLChannel(profile, 300e-9, 3.5e9, tx, rx, fs, Seed=sd) for sd in seeds]
    for ch, t in zip(chans, t_start):
        ch.time = t
    shared = (not uplink) and bool(rng.integers(0, 2))
    jobs = [(u, s_) for s_ in range(n_slots) for u in range(n_ue)]
    n_wave = n_slots if shared else len(jobs)
    xs = [np.asfortranarray(rng.standard_normal((t_len, nt)) + 1j * rng.standard_normal((t_len, nt))) for _ in range(n_wave)]
    d_xs = [ctx.to_device(x) for x in xs]
    which = [(s_ if shared else i) for i, (u, s_) in enumerate(jobs)]
    outs = CM.applyCDLBatch([chans[u] for u, _ in jobs], [d_xs[w] for w in which], ctx=ctx)
    for (u, s_), w, o in zip(jobs, which, outs):
        cfg = OC.cdl_config(profile, 3.5e9, tx, rx, fs, seed=seeds[u])
        want = OC.apply_cdl(cfg, xs[w], t_start[u] + s_ * t_len / fs)
        got = o.numpy()
        assert got.shape == (t_len, nr) and rel(got, want) < RTOL, (seed, profile, uplink, gnb, fs, t_len, u, s_)


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + max(8, N_CASES // 4)))
def test_sinr_cqi_on_random_configuration(pkg, seed):
    import oracle.cqi as OQ
    rng = np.random.default_rng(9500 + seed)
    nl = int(rng.integers(1, 9))
    p = int(rng.choice([q for q in (2, 4, 8, 16, 32) if q >= nl]))
    nr = int(rng.choice([r for r in (1, 2, 4, 8, 16) if r >= 1]))
    n_re = int(rng.integers(1, 3000))
    h = np.asfortranarray((rng.standard_normal((n_re, nr, p)) + 1j * rng.standard_normal((n_re, nr, p))) * rng.uniform(0.1, 10.0))
    w, _ = np.linalg.qr(rng.standard_normal((p, nl)) + 1j * rng.standard_normal((p, nl)))
    w = w / np.sqrt(nl)
    sigma = float(rng.uniform(0.05, 3.0))
    want = OQ.precoded_sinr_batch(h, sigma, w)
    got = pkg.communication.phyLayer.precodedSINR(h, sigma, w)
    assert got.shape == (n_re,) and np.abs(got - want).max() <= 1e-9 * np.abs(want).max()
    cqi, mean = pkg.communication.phyLayer.cqiFromChannel(h, sigma, w, OQ.DOWNLINK_SINR90PC)
    assert mean == pytest.approx(want.mean(), rel=1e-11)
    edges = 10.0 ** (np.asarray(OQ.DOWNLINK_SINR90PC) / 10.0)
    if np.min(np.abs(want.mean() / edges - 1.0)) > 1e-9:      # away from a table edge the integer CQI is exact
        assert cqi == OQ.get_cqi(want.mean(), OQ.DOWNLINK_SINR90PC)


N_SPECTRAL = max(8, N_CASES // 2)


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + N_SPECTRAL))
def test_spectral_fused_path_on_random_scene(pkg, seed, record_property):
    """The route bench.py times, fuzzed: per-target demodulation -> fused synthesis + range kernel with the AWGN on the demodulated grid ->
    cached fft2D, for random antenna counts (incl. the 33..64 range), 1..6 targets (compile-time kernels 1..4 and the run-time one),
    zero-filled 'S' slots or not.  (a) injected spectral field W: echo grid, CFAR lists and estimates against the oracle's TIME-domain chain fed
    the equivalent time-domain noise; (b) Philox spectral mode: echo grid against the oracle fed the restated generator's field;
    (c) fused == unfused bit for bit."""
    from conftest import spectral_to_time_noise
    rng = np.random.default_rng(9000 + seed)
    q = int(rng.integers(1, 7))
    n_ants = int(rng.choice([1, 2, 3, 5, 8, 17, 40, 56, 64, 72]))    # (56 / 64: cov_mfma_lds_kernel<4>, the bench's own instantiation; 72: the block kernel + the distributed tridiagonalisation)
    n_slots = int(rng.choice([2, 4] if n_ants > 8 else [2, 3, 4])) if n_ants <= 40 else 2
    r = rng.uniform(60.0, 300.0, q)
    az = np.deg2rad(rng.uniform(-70.0, 70.0, q))
    targets = tuple((float(r[i] * np.cos(az[i])), float(r[i] * np.sin(az[i])), 1.5) for i in range(q))
    sc = make_scene(n_ants=n_ants, n_slots=n_slots, nrb=273, targets=targets, velocity=tuple(float(v) for v in rng.integers(-10, 11, q)),
                    seed=seed, zero_s_slots=bool(rng.integers(0, 2)), with_noise=False)
    ctx = pkg.default_context()
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    d_wave, d_txg = ctx.to_device(sc.tx_wave), ctx.to_device(sc.tx_grid)
    w = np.asfortranarray(rng.standard_normal(sc.tx_grid.shape) + 1j * rng.standard_normal(sc.tx_grid.shape))
    d_w = ctx.to_device(w)
    e_f = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, spectral_noise=d_w, fuse_fft2d=(rp, cf, d_txg))
    ref_echo = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los,
                                     spectral_to_time_noise(w, sc.T, 4096, 30, sc.rp.fc, sc.rp.fs), nfft=4096)
    g_f = e_f.numpy()
    assert rel(g_f, ref_echo) < RTOL
    e_u = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, spectral_noise=d_w)
    assert np.array_equal(g_f, e_u.numpy())
    # Philox spectral mode against the restated generator
    e_p = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, seed=0x77 + seed, noise_domain="spectral",
                                        fuse_fft2d=(rp, cf, d_txg))
    wp = O.philox_spectral_noise(sc.K, sc.L, sc.A, 0x77 + seed)
    ref_p = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los,
                                  spectral_to_time_noise(wp, sc.T, 4096, 30, sc.rp.fc, sc.rp.fs), nfft=4096)
    # float32 hardware Box-Muller on the device vs its float32 restatement: a float32 bound on the unit samples (oracle/philox.py)
    sig_w = np.sqrt(sc.rp.N0 / 2.0) * 64.0
    assert np.abs(e_p.numpy() - ref_p).max() < sig_w * O.philox.SPECTRAL_NOISE_ATOL + RTOL * np.abs(ref_p).max()
    # detections / estimates of the injected-field run against the oracle
    ocf = O.cfar2d_config(sc.rp)
    e_f = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, spectral_noise=d_w, fuse_fft2d=(rp, cf, d_txg))
    try:
        want, dbg = O.fft2d(sc.rp, ocf, ref_echo, sc.tx_grid, return_debug=True, rdm_fn=O.rdm_explicit)
    except ValueError:
        with pytest.raises(pkg.IsacError) as ei:
            pkg.sensing.estimation.fft2D(rp, cf, e_f, d_txg, reuse_range=True)
        assert ei.value.name == "NO_DETECTION"
        return
    if not all(_guard_band_ok(np.abs(dbg.rdm[:, :, a]) ** 2, ocf.CUTIdx, ocf.Pfa) for a in range(sc.A)):
        pytest.skip("a CUT sits within 1e-9 of its CFAR threshold: rounding-defined scene")
    got, gd = pkg.sensing.estimation.fft2D(rp, cf, e_f, d_txg, return_debug=True, reuse_range=True)
    assert all(np.array_equal(x, y) for x, y in zip(gd.detections, dbg.detections))
    assert np.array_equal(got.rngEst, want.rngEst) and np.array_equal(got.velEst, want.velEst)
    _check_azimuth(pkg, record_property, sc, rp, got, want, gd.Ra, dbg.Ra)      # same rule as the time-domain fuzz above


N_EIG = max(12, N_CASES)


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + N_EIG))
def test_eigh_top_on_random_spectrum(pkg, seed):
    """isac_eigh_top fuzzed over 65 <= n <= 256 (eigh_tridiag_dist_kernel -> bisection -> subspace kernel; every third seed 17 <= n <= 64: the one-workgroup
    reduction) on Hermitian matrices with a PRESCRIBED spectrum: k leading eigenvalues separated by random relative gaps between 1e-9 and 1 (tight pairs,
    well-separated ones, random dynamic range up to 1e12), a noise floor with its own spread, a random unitary basis.  Eigenvalues to 1e-13 ||H||; the returned
    vectors span an invariant subspace (residual 1e-11 ||H||, orthonormal to 1e-12) -- inside a tight cluster any rotation of the basis is right.  music.m:19-29."""
    ctx = pkg.default_context()
    rng = np.random.default_rng(31000 + seed)
    n = int(rng.integers(17, 65)) if seed % 3 == 2 else int(rng.integers(65, 257))
    lmax = max(1, min(32 if n <= 128 else 16, 122880 // (32 * n)))
    k = int(rng.integers(1, min(lmax, 8) + 1))
    top = [10.0 ** rng.uniform(0, 6)]
    for _ in range(k - 1):
        top.append(top[-1] * (1.0 - 10.0 ** rng.uniform(-9, -0.05)))
    floor_hi = top[-1] * 10.0 ** rng.uniform(-6, -0.3)
    floor = np.sort(floor_hi * 10.0 ** (-rng.uniform(0, rng.choice([0.0, 0.5, 6.0]), n - k)))[::-1]
    lam = np.concatenate([top, floor])
    q_, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    h = (q_ * lam) @ q_.conj().T
    h = np.asfortranarray(0.5 * (h + h.conj().T))
    wr = np.linalg.eigvalsh(h)
    scale = np.abs(wr).max()
    for n_top in sorted({k, min(k + 1, n - 1), 1}):
        w, u = ctx.eigh_top(h, n_top)
        assert np.abs(w - wr).max() < 1e-13 * scale, (seed, n, k, n_top)
        assert np.abs(u.conj().T @ u - np.eye(n_top)).max() < 1e-12, (seed, n, k, n_top)
        # invariant subspace: H U = U (U^H H U)
        assert np.abs(h @ u - u @ (u.conj().T @ h @ u)).max() < 1e-11 * scale, (seed, n, k, n_top)
