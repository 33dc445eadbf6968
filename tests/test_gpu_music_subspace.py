"""MUSIC's default eigensolver route on the device (csrc/music.hip, "eigensolver III"): Householder tridiagonalisation -> all eigenvalues by
Sturm-count bisection -> the L = numDets signal eigenvectors by block inverse iteration -> a' Uan Uan' a = ||a - Us Us' a||^2
(music.m:19-29,82-91), against (1) SciPy's eigh for the operator itself (isac_eigh_top), (2) the full-eigendecomposition route of the same
library (isac_ctx_set_option(ISAC_OPT_MUSIC_ROUTE, 1)) and (3) the oracle's music_doa / fft2d for the estimates.  Tolerances: eigenvalues <= 1e-13 ||H||,
orthonormality <= 1e-12, invariant-subspace residual <= 1e-11 ||H||; azimuth estimates exact."""
from __future__ import annotations

import numpy as np
import pytest
from scipy import linalg

import oracle as O
from oracle import subspace_music as SM
from conftest import load_pkg, make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


@pytest.fixture(scope="module")
def ctx(pkg):
    return pkg.Context()


def _lmax(a):
    return max(1, min(32 if a <= 128 else 16, 122880 // (32 * a)))


def _sample_cov(rng, a, q, n, snr):
    m = np.arange(a)
    ang = rng.uniform(-70, 70, q)
    sg = np.stack([np.exp(-2j * np.pi * m * 0.5 * np.sin(np.deg2rad(x))) for x in ang], 1)
    s = (rng.standard_normal((q, n)) + 1j * rng.standard_normal((q, n))) * np.sqrt(snr)
    x = sg @ s + (rng.standard_normal((a, n)) + 1j * rng.standard_normal((a, n)))
    ra = x @ x.conj().T / n
    return 0.5 * (ra + ra.conj().T)


@pytest.mark.parametrize("a", [3, 4, 5, 16, 33, 64, 65, 100, 128, 129, 200, 256])
def test_eigh_top_random_hermitian(ctx, a):
    rng = np.random.default_rng(a)
    m = rng.standard_normal((a, a)) + 1j * rng.standard_normal((a, a))
    h = m @ m.conj().T / a + np.diag(rng.uniform(0, 3, a))
    wr, vr = linalg.eigh(h)
    scale = np.abs(wr).max()
    for n_top in sorted({0, 1, 2, min(4, a), min(_lmax(a), a - 1), min(_lmax(a) + 1, a), a}):
        w, u = ctx.eigh_top(h, n_top)
        assert np.abs(w - wr).max() < 1e-13 * scale, (a, n_top)
        if n_top == 0:
            continue
        assert np.abs(u.conj().T @ u - np.eye(n_top)).max() < 1e-12, (a, n_top)
        lam = wr[::-1][:n_top]
        assert np.abs(h @ u - u * lam).max() < 1e-11 * scale, (a, n_top)


@pytest.mark.parametrize("kind,a", [("identity", 100), ("rank2", 130), ("diag_repeated", 96), ("tiny", 72), ("huge", 65), ("clustered", 200),
                                    ("tridiag_zero_blocks", 128), ("identity", 12), ("rank2", 40), ("tiny", 33), ("huge", 64),
                                    ("clustered", 48), ("zero", 20), ("zero", 80), ("sample_cov", 64), ("sample_cov", 256), ("rank1_plus_floor", 16)])
def test_eigh_top_degenerate_spectra(ctx, kind, a):
    """Exact splits, zero and repeated eigenvalues, scales near the fp64 range limits, tight clusters: eigenvalues to eps ||H||; the
    vectors are an orthonormal basis of an invariant subspace (inside a repeated eigenvalue any basis is right)."""
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
    elif kind == "sample_cov":
        h = _sample_cov(rng, a, 3, 4000, 1e4)
    elif kind == "rank1_plus_floor":
        v = np.exp(-2j * np.pi * np.arange(a) * 0.5 * float(O.sind(37)))
        h = np.outer(v, v.conj()) + 1e-3 * np.eye(a)
    else:
        h = np.zeros((a, a), dtype=np.complex128)
        for b0 in range(0, a, 16):                   # decoupled 16 x 16 Hermitian blocks
            m = rng.standard_normal((16, 16)) + 1j * rng.standard_normal((16, 16))
            h[b0:b0 + 16, b0:b0 + 16] = m + m.conj().T
    h = (h + h.conj().T) / 2
    wr = linalg.eigvalsh(h)
    scale = max(np.abs(wr).max(), 1e-300)
    for n_top in (1, 2, 3, min(8, a - 1)):
        w, u = ctx.eigh_top(h, n_top)
        assert np.all(np.isfinite(w)) and np.all(np.isfinite(u))
        assert np.abs(w - wr).max() < max(1e-13 * scale, 1e-300), (kind, a, n_top)      # (zero matrix: the bracket is a few pivmin wide)
        assert np.abs(u.conj().T @ u - np.eye(n_top)).max() < 1e-12, (kind, a, n_top)
        # invariant subspace: H U = U (U' H U) whenever the n_top-th and (n_top+1)-th eigenvalues are separated or the
        # whole cluster they share is flat to rounding
        wd = wr[::-1]
        if wd[n_top - 1] - wd[n_top] > 1e-6 * scale or np.ptp(wd[n_top - 1:]) < 1e-11 * scale or kind in ("identity", "zero", "diag_repeated"):
            assert np.abs(h @ u - u @ (u.conj().T @ h @ u)).max() < 1e-10 * scale, (kind, a, n_top)


def test_music_routes_agree_with_oracle(pkg, ctx):
    """doaEstimation.music through route 0 (subspace), route 1 (full eig) and the oracle: KATs with mirror ties, a model-order case,
    sample covariances of several sizes / SNRs / L (incl. L above the subspace kernel's capacity and L >= nAnts)."""
    sc = make_scene(n_ants=16, n_slots=1, nrb=24, with_noise=False)
    m16 = np.arange(16)
    cases = []
    for phi0 in (20, 37, -30, -61, 45):
        a = np.exp(-2j * np.pi * m16 * 0.5 * float(O.sind(phi0)))
        cases += [(np.outer(a, a.conj()) + 1e-3 * np.eye(16), l) for l in (1, 2)]
    a1 = np.exp(-2j * np.pi * m16 * 0.5 * float(O.sind(15)))
    a2 = np.exp(-2j * np.pi * m16 * 0.5 * float(O.sind(-40)))
    two = 4 * np.outer(a1, a1.conj()) + np.outer(a2, a2.conj()) + np.diag(np.random.default_rng(3).uniform(0.01, 0.03, 16))
    cases += [(two, l) for l in (None, 1, 2, 3, 5, 8, 15, 16, 20)]
    rng = np.random.default_rng(0)
    for a in (3, 4, 8, 16, 32, 64, 128, 256):
        for _ in range(3):
            q = int(rng.integers(1, 4))
            ra = _sample_cov(rng, a, q, int(rng.choice([500, 20000])), 10 ** rng.uniform(-1, 5))
            cases += [(ra, l) for l in sorted({1, q, min(q + 2, a - 1), min(a - 1, 6), min(_lmax(a) + 1, a - 1)}) if l >= 1]
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)      # (only the scan parameters matter: the array size is Ra's)
    for ra, l in cases:
        n = ra.shape[0]
        want = O.music_doa(l, sc.rp, ra)
        w = np.sort(np.linalg.eigvalsh(ra))[::-1]
        ctx.set_music_route(0)
        got0 = pkg.sensing.estimation.doaEstimation.music(l, rp, ra, ctx=ctx)
        ctx.set_music_route(1)
        got1 = pkg.sensing.estimation.doaEstimation.music(l, rp, ra, ctx=ctx)
        ctx.set_music_route(0)
        assert got0[0] == got1[0] == want[0], (n, l)
        ls = want[0]
        if 1 <= ls < n and (w[ls - 1] - w[ls]) < 1e-9 * w[0] and not np.array_equal(got0[1], want[1]):
            continue                                     # split inside a rounding-degenerate cluster: peaks undefined (see test_gpu_fuzz)
        assert np.array_equal(got0[1], want[1]), (n, l, got0[1], want[1])
        assert np.array_equal(got1[1], want[1]), (n, l)
        # the NumPy restatement of the route takes the same decisions
        assert np.array_equal(SM.music_doa_subspace(l, sc.rp, ra)[1], want[1]) or n > 64


def test_fft2d_chain_same_estimates_on_both_routes(pkg):
    sc = make_scene(n_ants=8, n_slots=4, nrb=24, targets=((150.0, 40.0, 1.5), (-90.0, 70.0, 5.0)), velocity=(0.0, 6.0), num_slots_param=6, seed=21)
    echo = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, sc.noise, nfft=sc.wave.Nfft)
    want = O.fft2d(sc.rp, O.cfar2d_config(sc.rp), echo, sc.tx_grid)
    import ctypes as C
    out, spec = [], []
    for route in (0, 1):
        c = pkg.Context()
        c.set_music_route(route)
        rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
        cf = pkg.sensing.detection.cfar2D(rp)
        d_e, d_t = c.to_device(echo), c.to_device(sc.tx_grid)
        out.append(pkg.sensing.estimation.fft2D(rp, cf, d_e, d_t))
        p = np.zeros(512)
        n = C.c_int32(0)
        c.check(c.lib.isac_fft2d_get_music_spectrum(c.handle, p.ctypes.data_as(C.c_void_p), C.c_int32(512), C.byref(n)))
        spec.append(p[: n.value].copy())
    for est in out:
        assert np.array_equal(est.aziEst, want.aziEst) and np.array_equal(est.rngEst, want.rngEst) and np.array_equal(est.velEst, want.velEst)
    # the two routes evaluate the same quadratic form a' Uan Uan' a (music.m:90): the dB spectra agree far below the peak-picking resolution
    assert spec[0].size == spec[1].size == 361 and np.abs(spec[0] - spec[1]).max() < 1e-6


def test_nan_covariance_is_an_error_not_a_hang(pkg, ctx):
    sc = make_scene(n_ants=16, n_slots=1, nrb=24, with_noise=False)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    ra = np.full((16, 16), np.nan, dtype=np.complex128)
    with pytest.raises(pkg.IsacError) as ei:
        pkg.sensing.estimation.doaEstimation.music(2, rp, ra, ctx=ctx)
    assert ei.value.name in ("HIP", "NO_DETECTION")
