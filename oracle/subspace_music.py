"""NumPy restatement of the SIGNAL-SUBSPACE route the HIP library takes for doaEstimation.music on a ULA
(csrc/music.hip: eigh_tridiag_* -> eigh_bisect_kernel -> music_subspace_kernel -> music_scan_kernel mode 3).

music.m:19-29 forms the noise projector Uan*Uan' from a full eig(Ra).  MUSIC only needs
    a' Uan Uan' a  =  || a - Us Us' a ||^2 ,   Us = the L eigenvectors of the L largest eigenvalues,
so the device computes just those L vectors: Householder tridiagonalisation (zhetd2), all eigenvalues by Sturm-count
bisection, block inverse iteration on the real tridiagonal (pivoted LU, modified Gram-Schmidt between the rounds), and
the back-transformation through the reflectors.  This file states the same steps in plain NumPy so that the numerics of
the route (peak positions against the full-eig oracle in oracle/music.py) can be checked on the CPU.

TEST INFRASTRUCTURE ONLY -- never imported by the product.
"""
from __future__ import annotations

import numpy as np

from .matlab_compat import EPS, sind, mag2db, findpeaks
from .music import ula_scan_angles

ROUNDS = 2          # inverse-iteration rounds of the device kernel (music_subspace_kernel)


def householder_tridiag(h):
    """zhetd2 (lower): A = Q T Q^H, Q = H_0 ... H_{n-2}, H_k = I - tau_k v_k v_k^H.  Returns d [n], e [n-1] (real), V (v_k in
    column k, rows k+1.., v_k[k+1] = 1), tau [n-1]."""
    a = np.array(h, dtype=np.complex128)
    n = a.shape[0]
    d = np.zeros(n)
    e = np.zeros(max(n - 1, 0))
    tau = np.zeros(max(n - 1, 0), dtype=np.complex128)
    v_all = np.zeros((n, n), dtype=np.complex128)
    for k in range(n - 1):
        alpha = a[k + 1, k]
        x = a[k + 2:, k]
        xnorm2 = float(np.sum(x.real ** 2 + x.imag ** 2))
        v = np.zeros(n, dtype=np.complex128)
        v[k + 1] = 1.0
        if xnorm2 != 0.0 or alpha.imag != 0.0:                         # zlarfg
            beta = -np.copysign(np.sqrt(alpha.real ** 2 + alpha.imag ** 2 + xnorm2), alpha.real)
            t = complex((beta - alpha.real) / beta, -alpha.imag / beta)
            v[k + 2:] = x / (alpha - beta)
        else:
            beta, t = alpha.real, 0.0
        d[k] = a[k, k].real
        e[k] = beta
        tau[k] = t
        v_all[:, k] = v
        if t != 0:
            a22 = a[k + 1:, k + 1:]
            vv = v[k + 1:]
            p = t * (a22 @ vv)
            w = p - 0.5 * t * np.vdot(p, vv) * vv
            a[k + 1:, k + 1:] = a22 - np.outer(vv, w.conj()) - np.outer(w, vv.conj())
    d[n - 1] = a[n - 1, n - 1].real
    return d, e, v_all, tau


def householder_tridiag_one_pass(h):
    """The same reduction the way eigh_tridiag_fused_kernel (csrc/music.hip, n > 64) walks the matrix: the rank-2 update of reflector k - 1 is
    carried as a pending pair (v, w) and applied element by element while the matrix-vector product of reflector k is accumulated -- column
    k first (it defines the reflector), then one pass over the trailing rows / columns.  Returns what householder_tridiag returns."""
    a = np.array(h, dtype=np.complex128)
    n = a.shape[0]
    d = np.zeros(n)
    e = np.zeros(max(n - 1, 0))
    tau = np.zeros(max(n - 1, 0), dtype=np.complex128)
    v_all = np.zeros((n, n), dtype=np.complex128)
    pv = np.zeros(n, dtype=np.complex128)                             # pending reflector / w (zero: nothing pending)
    pw = np.zeros(n, dtype=np.complex128)
    for k in range(n - 1):
        a[k:, k] = a[k:, k] - pv[k:] * np.conj(pw[k]) - pw[k:] * np.conj(pv[k])          # (a) column k
        alpha = a[k + 1, k]
        x = a[k + 2:, k]
        xnorm2 = float(np.sum(x.real ** 2 + x.imag ** 2))
        v = np.zeros(n, dtype=np.complex128)
        v[k + 1] = 1.0
        if xnorm2 != 0.0 or alpha.imag != 0.0:                         # zlarfg
            beta = -np.copysign(np.sqrt(alpha.real ** 2 + alpha.imag ** 2 + xnorm2), alpha.real)
            t = complex((beta - alpha.real) / beta, -alpha.imag / beta)
            v[k + 2:] = x / (alpha - beta)
        else:
            beta, t = alpha.real, 0.0
        d[k] = a[k, k].real
        e[k] = beta
        tau[k] = t
        v_all[:, k] = v
        r = slice(k + 1, n)                                            # (b) one pass: update, store, accumulate
        a[r, r] = a[r, r] - np.outer(pv[r], pw[r].conj()) - np.outer(pw[r], pv[r].conj())
        acc = a[r, r] @ v[r]
        pv[:] = 0.0
        pw[:] = 0.0
        if t != 0:                                                     # (c) the new pending pair
            p = t * acc
            pw[r] = p - 0.5 * t * np.vdot(p, v[r]) * v[r]
            pv[r] = v[r]
    d[n - 1] = (a[n - 1, n - 1] - pv[n - 1] * np.conj(pw[n - 1]) - pw[n - 1] * np.conj(pv[n - 1])).real
    return d, e, v_all, tau


def householder_tridiag_distributed(h, cols_per_owner=4):
    """The same reduction the way eigh_tridiag_dist_kernel (csrc/music.hip, 64 < n <= 256) distributes it: the matrix is held COLUMN-wise by owners of
    `cols_per_owner` columns each (full columns, both triangles); per reflector every owner forms its entries of p from its own columns only
    (p_j = tau sum_i conj(a_ij) v_i -- the matrix is Hermitian, no row access), ONE exchange makes p and the next column (as the owner holds it, the
    current update not yet applied) known to everybody, and everybody derives w, the updated next column and the next reflector redundantly before
    updating its own columns.  zlarfg uses two reciprocals (1 / beta, 1 / |alpha - beta|^2) instead of four divisions.  Returns what
    householder_tridiag returns."""
    a = np.array(h, dtype=np.complex128)
    n = a.shape[0]
    d = np.zeros(n)
    e = np.zeros(max(n - 1, 0))
    tau = np.zeros(max(n - 1, 0), dtype=np.complex128)
    v_all = np.zeros((n, n), dtype=np.complex128)
    owners = [range(c0, min(n, c0 + cols_per_owner)) for c0 in range(0, n, cols_per_owner)]

    def derive(k, col):                                                # the reflector of step k from column k (rows >= k valid)
        alpha = col[k + 1]
        x = col[k + 2:]
        xnorm2 = float(np.sum(x.real ** 2 + x.imag ** 2))
        v = np.zeros(n, dtype=np.complex128)
        v[k + 1] = 1.0
        beta, t = alpha.real, 0.0
        if xnorm2 != 0.0 or alpha.imag != 0.0:
            beta = -np.copysign(np.sqrt(alpha.real ** 2 + alpha.imag ** 2 + xnorm2), alpha.real)
            ib = 1.0 / beta
            t = complex((beta - alpha.real) * ib, -alpha.imag * ib)
            dl = alpha - beta
            idn = 1.0 / (dl.real ** 2 + dl.imag ** 2)
            v[k + 2:] = x * complex(dl.real * idn, -dl.imag * idn)
        d[k] = col[k].real
        e[k] = beta
        tau[k] = t
        v_all[:, k] = v
        return v, t

    if n < 2:
        d[:] = a.diagonal().real
        return d, e, v_all, tau
    v, t = derive(0, a[:, 0].copy())
    for k in range(n - 1):
        p = np.zeros(n, dtype=np.complex128)                           # the exchange: p (each owner its columns) + column k + 1 as its owner holds it
        for own in owners:
            for j in own:
                if j > k:
                    p[j] = t * np.vdot(a[:, j], v)                     # sum_i conj(a_ij) v_i  (v is zero above row k + 1)
        col = a[:, k + 1].copy()
        a2 = -0.5 * t * np.vdot(p, v)                                  # everybody: w, the next column brought up to date
        w = np.where(np.arange(n) > k, p + a2 * v, 0.0)
        wk1 = p[k + 1] + a2
        cn = col - v * np.conj(wk1) - w
        if k + 1 == n - 1:
            d[n - 1] = (col[k + 1] - 2.0 * wk1.real).real
            break
        v_next, t_next = derive(k + 1, cn)
        for own in owners:                                             # rank-2 update of the owners' columns
            for j in own:
                if j > k:
                    a[:, j] = a[:, j] - v * np.conj(w[j]) - w * np.conj(v[j])
        v, t = v_next, t_next
    return d, e, v_all, tau


def sturm_count(d, e2, x, pivmin):
    """Number of eigenvalues of tridiag(d, e) below x (negative pivots of T - x I, dstebz-style pivmin clamp)."""
    cnt = 0
    q = d[0] - x
    if abs(q) < pivmin:
        q = -pivmin
    cnt += q < 0
    for i in range(1, d.size):
        q = (d[i] - x) - e2[i - 1] / q
        if abs(q) < pivmin:
            q = -pivmin
        cnt += q < 0
    return int(cnt)


def bisect_all(d, e, lanes=64):
    """All eigenvalues, ascending, to an absolute accuracy of ~eps ||T|| by `lanes`-section of the Gershgorin interval."""
    n = d.size
    e2 = e * e
    ea = np.concatenate([[0.0], np.abs(e), [0.0]])
    gl = float(np.min(d - ea[:-1] - ea[1:]))
    gu = float(np.max(d + ea[:-1] + ea[1:]))
    bnorm = max(abs(gl), abs(gu))
    pivmin = np.finfo(float).tiny * max(1.0, float(e2.max()) if e2.size else 1.0)
    gl -= 2.0 * bnorm * EPS * n + 2.0 * pivmin
    gu += 2.0 * bnorm * EPS * n + 2.0 * pivmin
    tol = 2.0 * EPS * bnorm + 2.0 * pivmin
    w = np.empty(n)
    for i in range(n):
        lo, hi = gl, gu
        for _ in range(64):
            if hi - lo <= tol:
                break
            xs = lo + (hi - lo) * (np.arange(1, lanes + 1) / (lanes + 1.0))
            cs = np.array([sturm_count(d, e2, x, pivmin) for x in xs])
            below = xs[cs <= i]
            above = xs[cs > i]
            nlo = below.max() if below.size else lo
            nhi = above.min() if above.size else hi
            if nlo > nhi:
                nlo = nhi = 0.5 * (nlo + nhi)
            if nlo == lo and nhi == hi:
                break
            lo, hi = nlo, nhi
        w[i] = 0.5 * (lo + hi)
    return w


def _lu_solve_tridiag(d, e, lam, x, tiny):
    """(T - lam I) y = x by Gaussian elimination with partial pivoting (dlagtf / dlagts); zero pivots are replaced by `tiny`."""
    n = d.size
    y = x.copy()
    u0 = np.empty(n); u1 = np.zeros(n); u2 = np.zeros(n)
    a = d[0] - lam
    b = e[0] if n > 1 else 0.0
    for i in range(n - 1):
        c = e[i]
        dn = d[i + 1] - lam
        en = e[i + 1] if i + 1 < n - 1 else 0.0
        if abs(a) >= abs(c):
            if abs(a) < tiny:
                a = -tiny if a < 0.0 else tiny
            m = c / a
            u0[i], u1[i], u2[i] = a, b, 0.0
            a, b = dn - m * b, en
            y[i + 1] -= m * y[i]
        else:
            if abs(c) < tiny:
                c = -tiny if c < 0.0 else tiny
            m = a / c
            u0[i], u1[i], u2[i] = c, dn, en
            a, b = b - m * dn, -m * en
            y[i], y[i + 1] = y[i + 1], y[i] - m * y[i + 1]
    if abs(a) < tiny:
        a = np.copysign(tiny, a) if a != 0.0 else tiny
    u0[n - 1] = a
    for i in range(n - 1, -1, -1):
        s = y[i]
        if i + 1 < n:
            s -= u1[i] * y[i + 1]
        if i + 2 < n:
            s -= u2[i] * y[i + 2]
        y[i] = s / u0[i]
    return y


def start_vector(n, j):
    """Deterministic pseudo-random start vector of column j (the device uses the same LCG)."""
    s = (0x9E3779B9 * (j + 1) + 0x7F4A7C15) & 0xFFFFFFFF
    out = np.empty(n)
    for i in range(n):
        s = (1664525 * s + 1013904223) & 0xFFFFFFFF
        out[i] = (s >> 8) * (1.0 / 16777216.0) - 0.5
    return out


def signal_vectors_tridiag(d, e, w, n_sig, rounds=ROUNDS):
    """Orthonormal basis Z [n x n_sig] (real) of the invariant subspace of the n_sig largest eigenvalues of tridiag(d, e)."""
    n = d.size
    lam = w[::-1][:n_sig]                                     # descending
    tnorm = max(np.abs(d).max(), np.abs(e).max() if e.size else 0.0, 2.0 ** -400)   # (the device works on the safe-scaled matrix)
    tiny = EPS * tnorm
    z = np.stack([start_vector(n, j) for j in range(n_sig)], axis=1)
    for r in range(rounds):
        for j in range(n_sig):
            y = _lu_solve_tridiag(d, e, lam[j], z[:, j], tiny)
            z[:, j] = y / np.abs(y).max()
        for _ in range(2 if r == rounds - 1 else 1):          # modified Gram-Schmidt in descending-eigenvalue order
            for j in range(n_sig):
                for i in range(j):
                    z[:, j] -= (z[:, i] @ z[:, j]) * z[:, i]
                z[:, j] /= np.sqrt(z[:, j] @ z[:, j])
    return z


def back_transform(z, v_all, tau):
    """U = Q Z with Q = H_0 ... H_{n-2}."""
    u = z.astype(np.complex128)
    n = u.shape[0]
    for k in range(n - 2, -1, -1):
        if tau[k] != 0:
            v = v_all[:, k]
            u -= np.outer(tau[k] * v, v.conj() @ u)
    return u


def music_spectrum_subspace(us, n_ants, rp):
    """music.m:82-96 with a' Uan Uan' a evaluated as || a - Us Us' a ||^2."""
    nn = np.arange(n_ants, dtype=np.float64)
    angles = ula_scan_angles(rp)
    p = np.empty(angles.size)
    for i, ang in enumerate(angles):
        aa = np.exp(-2j * np.pi * nn * 0.5 * float(sind(ang)))
        r = aa - us @ (us.conj().T @ aa) if us.shape[1] else aa
        x = float(np.sum(r.real ** 2 + r.imag ** 2)) if us.shape[1] < n_ants else 0.0
        p[i] = abs(1.0 / (x + EPS))
    with np.errstate(divide="ignore"):
        return mag2db(p / p.max())


def music_doa_subspace(num_dets, rp, ra):
    """Same contract as oracle.music.music_doa, through the subspace route."""
    from .music import determine_num_targets
    ra = np.asarray(ra, dtype=np.complex128)
    n = ra.shape[0]
    d, e, v_all, tau = householder_tridiag(ra)
    w = bisect_all(d, e)
    n_sig = determine_num_targets(w) if num_dets is None else int(num_dets)
    k = min(n_sig, n)
    z = signal_vectors_tridiag(d, e, w, k) if k < n else np.eye(n)
    us = back_transform(z, v_all, tau) if k < n else np.eye(n, dtype=np.complex128)
    pdb = music_spectrum_subspace(us, n, rp)
    _, locs = findpeaks(pdb, npeaks=n_sig, sort_descend=True)
    azi = locs * rp.azimuthScanGranularity - rp.azimuthScanScale / 2.0
    return n_sig, azi.astype(np.float64), np.full(azi.shape, np.nan), pdb, w
