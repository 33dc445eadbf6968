"""Counter-based AWGN generator used in the library's *performance* noise mode, restated
for the oracle (the reference's ``randn`` at basicRadarChannel.m:68 is MATLAB's
mt19937ar+ziggurat stream and is not reproducible outside MATLAB -- SURVEY.md A.7).

Philox4x32-10 (Salmon et al., SC'11; Random123 constants), one call per complex
sample:  counter = (e_lo, e_hi, stream, 0) with e = t + T*a the column-major element
index of rxWaveform(t, a);  key = (seed_lo, seed_hi).  Box-Muller on two 53-bit
uniforms gives independent N(0,1) real and imaginary parts.

``philox_spectral_noise`` restates the library's *spectral* noise mode (ISAC_NOISE_PHILOX_SPECTRAL,
csrc/echo_dev.hpp): the AWGN is drawn directly on the demodulated grid, one Philox call per PAIR of grid
elements, 32-bit Box-Muller uniforms transformed in SINGLE precision (the device uses the hardware's
v_log_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32, ~1 ulp approximations: device and restatement agree to a
float32 bound -- ``SPECTRAL_NOISE_ATOL`` on unit-variance samples -- not bit for bit; the mode is statistical
parity with the reference's randn by construction).
TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)

# |device - restatement| on unit-variance samples: measured 9.8e-7 max on 367 k samples (profiles/r04_generator_accuracy.txt): a few float32 ulps of a value <= 6.76
SPECTRAL_NOISE_ATOL = 5e-6
SPECTRAL_NOISE_MAX = float(np.sqrt(2.0 * 33.0 * np.log(2.0)))


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All inputs broadcastable uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint32).copy()
    c1 = np.asarray(c1, dtype=np.uint32).copy()
    c2 = np.asarray(c2, dtype=np.uint32).copy()
    c3 = np.asarray(c3, dtype=np.uint32).copy()
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for r in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def philox_normal_pairs(elem_index, seed: int, stream: int = 0) -> np.ndarray:
    """Complex N(0,1)+jN(0,1) sample for each 64-bit element index."""
    e = np.asarray(elem_index, dtype=np.uint64)
    x0, x1, x2, x3 = philox4x32_10((e & _MASK).astype(np.uint32), (e >> np.uint64(32)).astype(np.uint32),
                                   np.uint32(stream), np.uint32(0),
                                   np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
    w0 = x0.astype(np.uint64) | (x1.astype(np.uint64) << np.uint64(32))
    w1 = x2.astype(np.uint64) | (x3.astype(np.uint64) << np.uint64(32))
    u1 = ((w0 >> np.uint64(11)).astype(np.float64) + 1.0) * 2.0 ** -53      # (0, 1]
    u2 = (w1 >> np.uint64(11)).astype(np.float64) * 2.0 ** -53              # [0, 1)
    r = np.sqrt(-2.0 * np.log(u1))
    ang = 2.0 * np.pi * u2
    return r * np.cos(ang) + 1j * (r * np.sin(ang))


def philox_spectral_noise(n_sc: int, n_sym: int, n_ants: int, seed: int, columns=None) -> np.ndarray:
    """Unit complex normals W [n_sc x n_sym x n_ants] of the spectral noise mode (`columns`: only these columns
    l + n_sym * a, returned as [n_sc x len(columns)] -- for spot checks at the full benchmark size).

    Element (k, l, a), column = l + n_sym * a:  slot = (k mod 512) + 512 * ((k div 512) div 2),
    half = (k div 512) mod 2  (elements k and k + 512 share a call);  Philox4x32-10 counter = (slot + 2048 * column as 64 bits,
    stream word 2, 0), key = seed;
    outputs (o[2 half], o[2 half + 1]) -> u = fl32(o) 2^-32 + 2^-33 in (0, 1], theta = 2 pi fl32(o') 2^-32;
    W = sqrt(-2 ln u) (cos theta + j sin theta), every operation in float32, widened to float64 at the end."""
    k = np.arange(n_sc, dtype=np.uint64)
    slot = (k % np.uint64(512)) + np.uint64(512) * ((k // np.uint64(512)) // np.uint64(2))
    half = ((k // np.uint64(512)) % np.uint64(2)).astype(np.int64)
    col = np.arange(n_sym * n_ants, dtype=np.uint64) if columns is None else np.asarray(columns, dtype=np.uint64)
    ctr = slot[:, None] + np.uint64(2048) * col[None, :]
    x = philox4x32_10((ctr & _MASK).astype(np.uint32), (ctr >> np.uint64(32)).astype(np.uint32), np.uint32(2), np.uint32(0),
                      np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
    h = half[:, None]
    f32 = np.float32
    ur = np.where(h == 0, x[0], x[2]).astype(f32)                       # uint32 -> float32, round to nearest even (v_cvt_f32_u32)
    ua = np.where(h == 0, x[1], x[3]).astype(f32)
    u = ur * f32(2.0 ** -32) + f32(2.0 ** -33)                          # (0, 1]; the product is exact, so this equals the device's fma
    rad = np.sqrt(np.log2(u) * f32(-1.3862943611198906))                # sqrt(-2 ln u) in float32
    turns = (ua * f32(2.0 ** -32)).astype(np.float64)                   # [0, 1]; sin / cos of the float32 argument, rounded to float32
    cs = np.cos(2.0 * np.pi * turns).astype(f32)
    sn = np.sin(2.0 * np.pi * turns).astype(f32)
    w = (rad * cs).astype(np.float64) + 1j * (rad * sn).astype(np.float64)
    if columns is not None:
        return np.asfortranarray(w)
    return np.asfortranarray(w.reshape(n_sc, n_sym, n_ants, order="F"))
