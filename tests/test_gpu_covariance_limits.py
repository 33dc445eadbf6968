"""Covariance (fft2D.m:106-107) at the edges of its 32-bit buffer addressing: the kernels address a 16-antenna block through one raw-buffer
descriptor, so N x 16 antennas x 16 B must stay below 4 GB (A <= 64: N < 2^24 samples per antenna; the LDS-staged form N < 2^23) and, for A > 64, N x 32
antennas x 16 B below 2 GB for the pipelined block kernel (N < 2^22) / N x 16 x 16 B below 2 GB for the burst form (N < 2^23).  Every switch-over is crossed
here with the result checked against a torch fp64 product of the same device data; one sample more than the last supported count is a clean
ISAC_ERR_UNSUPPORTED, not a wrapped offset.  (A CPI of the reference is 16 slots = 0.98 M samples; 2^24 samples are 273 slots.)"""
import ctypes as C

import numpy as np
import pytest

from conftest import load_pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


@pytest.mark.parametrize("n,a", [((1 << 24) - 1, 2),           # one antenna block, last supported sample count
                                 ((1 << 23) - 1, 40),          # three blocks: LDS-staged kernel at its limit
                                 ((1 << 23) + 3, 40),          # ... just above it: the register-operand kernel
                                 ((1 << 22) - 1, 65),          # A > 64: pipelined block kernel at its limit
                                 ((1 << 22) + 5, 65),          # ... just above it: the burst-form block kernel
                                 ((1 << 23) - 1, 65)])         # ... at ITS limit
def test_covariance_at_the_addressing_limits(pkg, n, a):
    import torch
    ctx = pkg._lib.Context(0)
    gen = torch.Generator(device="cuda").manual_seed(n % 1000 + a)
    g = torch.randn((a, n, 2), dtype=torch.float64, device="cuda", generator=gen)      # [antenna][sample][re, im] = column-major [n x a] complex
    g[0] *= 3.0
    gc = torch.view_as_complex(g)
    want = (gc.conj() @ gc.T / n).cpu().numpy()                                       # want[i, j] = sum_n conj(G[n, i]) G[n, j] / N
    torch.cuda.synchronize()
    d_ra = ctx.empty((a, a))
    ctx.check(ctx.lib.isac_covariance_dev(ctx.handle, C.c_void_p(g.data_ptr()), C.c_int64(n), C.c_int32(a), C.c_void_p(d_ra.ptr)))
    ra = d_ra.numpy()
    assert np.abs(ra - want).max() < 1e-11 * np.abs(want).max() and np.array_equal(ra, ra.conj().T)
    del g, gc
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n,a", [(1 << 24, 2), (1 << 24, 64), (1 << 23, 65), (1 << 23, 256)])
def test_covariance_beyond_the_limit_is_refused(pkg, n, a):
    ctx = pkg._lib.Context(0)
    d_any = ctx.empty((16, 16))                                # (the check precedes every memory access)
    with pytest.raises(pkg.IsacError) as ei:
        ctx.check(ctx.lib.isac_covariance_dev(ctx.handle, C.c_void_p(d_any.ptr), C.c_int64(n), C.c_int32(a), C.c_void_p(d_any.ptr)))
    assert ei.value.name == "UNSUPPORTED" and "samples per antenna" in str(ei.value)
