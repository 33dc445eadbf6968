"""oracle/subspace_music.py (the NumPy restatement of the device's signal-subspace MUSIC route: zhetd2 -> bisection -> block inverse
iteration -> back-transformation -> ||a - Us Us' a||^2) against the line-by-line oracle of music.m (full eig, explicit noise projector):
identical azimuth estimates, eigenvalues to eps ||Ra||.  CPU only; the device kernels are checked against both in
tests/test_gpu_music_subspace.py."""
from __future__ import annotations

import numpy as np
import pytest
from scipy import linalg

import oracle as O
from oracle import subspace_music as SM
from conftest import make_scene


def _cases():
    m = np.arange(16)
    for phi0 in (20, -30, 45):
        a = np.exp(-2j * np.pi * m * 0.5 * float(O.sind(phi0)))
        for l in (1, 2):
            yield np.outer(a, a.conj()) + 1e-3 * np.eye(16), l          # exactly degenerate noise floor + mirror ties
    a1 = np.exp(-2j * np.pi * m * 0.5 * float(O.sind(15)))
    a2 = np.exp(-2j * np.pi * m * 0.5 * float(O.sind(-40)))
    two = 4 * np.outer(a1, a1.conj()) + np.outer(a2, a2.conj()) + np.diag(np.random.default_rng(3).uniform(0.01, 0.03, 16))
    for l in (None, 2, 5, 16, 20):
        yield two, l
    rng = np.random.default_rng(0)
    for a in (3, 5, 8, 12):
        q = int(rng.integers(1, 3))
        mm = np.arange(a)
        sg = np.stack([np.exp(-2j * np.pi * mm * 0.5 * np.sin(np.deg2rad(x))) for x in rng.uniform(-60, 60, q)], 1)
        s = (rng.standard_normal((q, 2000)) + 1j * rng.standard_normal((q, 2000))) * 30.0
        x = sg @ s + (rng.standard_normal((a, 2000)) + 1j * rng.standard_normal((a, 2000)))
        ra = x @ x.conj().T / 2000
        for l in (1, q, min(q + 2, a - 1)):
            yield 0.5 * (ra + ra.conj().T), l


def test_subspace_route_restatement_matches_full_eig_oracle():
    sc = make_scene(n_ants=16, n_slots=1, nrb=24, with_noise=False)
    n = 0
    for ra, l in _cases():
        want = O.music_doa(l, sc.rp, ra)
        got = SM.music_doa_subspace(l, sc.rp, ra)
        wr = linalg.eigvalsh(ra)
        assert np.abs(got[4] - wr).max() < 1e-13 * np.abs(wr).max()
        assert got[0] == want[0] and np.array_equal(got[1], want[1]), (ra.shape, l, got[1], want[1])
        n += 1
    assert n >= 20


def test_restatement_building_blocks():
    rng = np.random.default_rng(7)
    m = rng.standard_normal((10, 10)) + 1j * rng.standard_normal((10, 10))
    h = m @ m.conj().T
    d, e, v_all, tau = SM.householder_tridiag(h)
    t = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    q = SM.back_transform(np.eye(10), v_all, tau)
    assert np.abs(q.conj().T @ q - np.eye(10)).max() < 1e-13
    assert np.abs(q @ t @ q.conj().T - h).max() < 1e-12 * np.abs(h).max()          # A = Q T Q^H
    w = SM.bisect_all(d, e)
    assert np.abs(w - linalg.eigvalsh(h)).max() < 1e-13 * np.abs(h).max()
    z = SM.signal_vectors_tridiag(d, e, w, 3)
    assert np.abs(z.T @ z - np.eye(3)).max() < 1e-13
    assert np.abs(t @ z - z * w[::-1][:3]).max() < 1e-11 * np.abs(h).max()


@pytest.mark.parametrize("n", [3, 8, 33, 65, 130])
def test_one_pass_householder_equals_zhetd2(n):
    """The deferred-update walk of eigh_tridiag_fused_kernel (restated in oracle.subspace_music.householder_tridiag_one_pass) is zhetd2:
    same d, e, tau and reflectors (the device kernels agree bit for bit -- tools/_tridiag_ab.py; the NumPy forms order their sums differently)."""
    from oracle.subspace_music import householder_tridiag, householder_tridiag_one_pass
    rng = np.random.default_rng(n)
    m = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    h = m @ m.conj().T / n + np.diag(rng.uniform(0, 3, n))
    if n == 8:                                        # a column that needs no reflector (tau = 0) in the middle of the walk
        h[4:, 2] = 0.0; h[2, 4:] = 0.0
        h[3, 2] = h[2, 3] = 0.7
    d0, e0, v0, t0 = householder_tridiag(h)
    d1, e1, v1, t1 = householder_tridiag_one_pass(h)
    s = np.abs(h).max()
    assert np.abs(d0 - d1).max() < 1e-13 * s and np.abs(e0 - e1).max() < 1e-13 * s
    assert np.abs(t0 - t1).max() < 1e-13 and np.abs(v0 - v1).max() < 1e-12


@pytest.mark.parametrize("n", [3, 8, 33, 65, 130, 256])
def test_distributed_householder_equals_zhetd2(n):
    """The column-owner walk of eigh_tridiag_dist_kernel (restated in oracle.subspace_music.householder_tridiag_distributed: p from the owners' columns
    only, one exchange per reflector, the next column updated and the next reflector derived by everybody, two reciprocals in zlarfg) is zhetd2:
    same d, e, tau and reflectors to rounding.  On the device the placement modes agree bit for bit and with numpy.linalg to 1e-12
    (tests/test_gpu_parity.py::test_distributed_tridiagonalisation_modes_agree)."""
    from oracle.subspace_music import householder_tridiag, householder_tridiag_distributed
    rng = np.random.default_rng(n)
    m = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    h = m @ m.conj().T / n + np.diag(rng.uniform(0, 3, n))
    if n == 8:                                        # a column that needs no reflector (tau = 0) in the middle of the walk
        h[4:, 2] = 0.0; h[2, 4:] = 0.0
        h[3, 2] = h[2, 3] = 0.7
    d0, e0, v0, t0 = householder_tridiag(h)
    d1, e1, v1, t1 = householder_tridiag_distributed(h, cols_per_owner=4 if n > 8 else 3)
    s = np.abs(h).max()
    assert np.abs(d0 - d1).max() < 1e-12 * s and np.abs(e0 - e1).max() < 1e-12 * s
    assert np.abs(t0 - t1).max() < 1e-12 and np.abs(v0 - v1).max() < 1e-11
