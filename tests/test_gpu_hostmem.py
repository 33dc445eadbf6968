"""Host memory handling of round 6 (include/isac.h "device memory + copies"): caller arrays reach the device through the context's pinned bounce buffer in 8 MB chunks
(copy_h2d / copy_d2h) and caller-visible device memory comes from a per-device pool of parked blocks.  Exact round trips at sizes around the chunk boundaries, pool reuse,
and many allocate / upload / compute / free cycles with changing sizes (the pattern of the fuzz case that read a zero channel estimate, profiles/r06_fuzz_campaigns.txt)."""
from __future__ import annotations

import numpy as np
import pytest

from conftest import load_pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


@pytest.mark.parametrize("n_bytes", [1, 15, 4096, (8 << 20) - 8, 8 << 20, (8 << 20) + 8, (16 << 20) + 24, (40 << 20) + 136])
def test_round_trip_through_the_bounce_buffer(pkg, n_bytes):
    ctx = pkg.default_context()
    rng = np.random.default_rng(n_bytes)
    a = rng.integers(0, 256, n_bytes, dtype=np.uint8)
    d = ctx.to_device(a)
    back = d.numpy()
    assert back.dtype == np.uint8 and np.array_equal(back, a)
    # an unaligned window of the device array back into an unaligned host view
    if n_bytes > 64:
        import ctypes as C
        out = np.zeros(n_bytes + 7, dtype=np.uint8)
        ctx.check(ctx.lib.isac_memcpy_d2h(ctx.handle, C.c_void_p(out.ctypes.data + 3), C.c_void_p(d.ptr + 5), C.c_size_t(n_bytes - 9)))
        assert np.array_equal(out[3:3 + n_bytes - 9], a[5:n_bytes - 4]) and not out[:3].any() and not out[n_bytes - 6:].any()


def test_pool_hands_a_parked_block_out_again(pkg):
    ctx = pkg.default_context()
    n = 77_777                                          # (an element count no other test uses: the pool serves the smallest parked block that fits, oldest first)
    a = ctx.empty((n,), np.complex128)                  # 1.24 MB
    p = a.ptr
    a.free()
    b = ctx.empty((n,), np.complex128)
    assert b.ptr == p                                   # the parked block, not a new allocation
    c = ctx.empty((n,), np.complex128)
    assert c.ptr != p                                   # b still holds it
    big = ctx.empty((n * 64,), np.complex128)           # 80 MB: the parked 1.24 MB block cannot serve it
    assert big.ptr not in (b.ptr, c.ptr)
    b.free()
    again = ctx.empty((n - 1000,), np.complex128)       # within the 25 % + 64 KB slack of the parked block
    assert again.ptr == p
    import ctypes as C
    again.free()
    assert ctx.lib.isac_dev_free(ctx.handle, C.c_void_p(p)) != 0     # freeing a parked block again is refused (ISAC_ERR_INVALID_ARG), not queued twice


def test_allocate_upload_compute_free_cycles(pkg):
    """precodedSINR / cqiFromChannel on freshly uploaded estimates of changing size, 300 times: the mean of the second call equals the mean of the first call's per-RE values."""
    PL = pkg.communication.phyLayer
    rng = np.random.default_rng(7)
    for it in range(300):
        n_re, nr, p, nl = int(rng.integers(1, 3000)), int(rng.choice([1, 2, 4, 16])), int(rng.choice([4, 8, 32])), int(rng.integers(1, 5))
        h = np.asfortranarray((rng.standard_normal((n_re, nr, p)) + 1j * rng.standard_normal((n_re, nr, p))) * rng.uniform(0.1, 10.0))
        w, _ = np.linalg.qr(rng.standard_normal((p, nl)) + 1j * rng.standard_normal((p, nl)))
        w = w / np.sqrt(nl)
        sigma = float(rng.uniform(0.05, 3.0))
        per = PL.precodedSINR(h, sigma, w)
        assert per.min() > 0.0                          # a zero would be an all-zero channel row
        _, mean = PL.cqiFromChannel(h, sigma, w)
        assert mean == pytest.approx(per.mean(), rel=1e-12), it
