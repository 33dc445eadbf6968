"""Philox4x32-10 (Random123) on the host: the seeded draws of the CDL channel (ray phases / couplings).
The sensing kernels carry their own device implementation (csrc/echo.hip)."""
from __future__ import annotations

import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_LO = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0.astype(np.uint64)
        p1 = _M1 * c2.astype(np.uint64)
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ np.uint32(k0), (p1 & _LO).astype(np.uint32),
                          (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ np.uint32(k1), (p0 & _LO).astype(np.uint32))
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def uniform(seed: int, stream: int, n: int, tag: int = 0xCD1) -> np.ndarray:
    """n doubles in [0,1): counter (i, 0, stream, tag), key = seed."""
    i = np.arange(n, dtype=np.uint64)
    x0, x1, _, _ = philox4x32_10((i & _LO).astype(np.uint32), np.uint32(0), np.uint32(stream), np.uint32(tag),
                                 seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    w = x0.astype(np.uint64) | (x1.astype(np.uint64) << np.uint64(32))
    return (w >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
