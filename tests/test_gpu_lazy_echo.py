"""The LAZY echo grid (include/isac.h 'LAZY echo grid'): isac_mono_static_sensing_fused_dev with d_echo_grid == NULL keeps echoGrid inside the
context -- on the spectral Philox route with 49..64 antennas and one or two LoS targets as a descriptor that the covariance kernel re-forms tile by tile (cov_lazy_kernel),
everywhere else in a context-owned buffer -- and isac_fft2d_submit_cached_dev with d_rx_grid == NULL consumes it.  Against the MATERIALISING sequence of the same calls:
|rdm|^2 window, every antenna's CFAR list, range / velocity / azimuth estimates identical; Ra <= 1e-13 (same terms, another summation order; bit-equal where the grid is
in memory); the materialised grid bit for bit the array the non-lazy call stores; Ra also against a NumPy covariance of that array (monoStaticSensing.m:1-23,
fft2D.m:37-46,59-115)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

from conftest import load_pkg, make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300))


def _targets(q):
    return (((100.0, 20.0, 1.5),), ((120.0, 60.0, 1.5), (-250.0, 80.0, 1.5)), ((120.0, 60.0, 1.5), (-250.0, 80.0, 1.5), (60.0, -30.0, 1.5)))[q - 1], ((7.0,), (10.0, -6.0), (10.0, -6.0, 3.0))[q - 1]


def _fft2d(pkg, rp, cf, grid, d_txg, ctx):
    """(est | None, debug): a scene without a CFAR detection raises NO_DETECTION in both forms (findpeaks NPeaks = 0, music.m:102) -- the |rdm|^2 window, the (empty) lists
    and Ra are still there to compare."""
    from importlib import import_module
    dbg_fn = import_module(pkg.__name__ + ".sensing.estimation.fft2D").fft2D_debug
    try:
        return pkg.sensing.estimation.fft2D(rp, cf, grid, d_txg, return_debug=True, reuse_range=True)
    except pkg.IsacError as e:
        assert e.name == "NO_DETECTION"
        return None, dbg_fn(ctx, grid.shape[2])


def _both(pkg, ctx, sc, rp, cf, d_wave, d_txg, shape, **kw):
    arr = pkg.sensing.monoStaticSensing(d_wave, shape, sc.carrier, rp, sc.los, nfft=4096, fuse_fft2d=(rp, cf, d_txg), ctx=ctx, **kw)
    est_a, dbg_a = _fft2d(pkg, rp, cf, arr, d_txg, ctx)
    lz = pkg.sensing.monoStaticSensing(d_wave, shape, sc.carrier, rp, sc.los, nfft=4096, fuse_fft2d=(rp, cf, d_txg), ctx=ctx, lazy=True, **kw)
    assert lz.shape == tuple(arr.shape)
    est_l, dbg_l = _fft2d(pkg, rp, cf, lz, d_txg, ctx)
    return arr, (est_a, dbg_a), lz, (est_l, dbg_l)


def _same_estimates(a, b):
    (est_a, dbg_a), (est_l, dbg_l) = a, b
    assert np.array_equal(dbg_a.power_window, dbg_l.power_window)
    assert all(np.array_equal(x, y) for x, y in zip(dbg_a.detections, dbg_l.detections))
    assert (est_a is None) == (est_l is None)
    if est_a is not None:
        assert np.array_equal(est_a.rngEst, est_l.rngEst) and np.array_equal(est_a.velEst, est_l.velEst) and np.array_equal(est_a.aziEst, est_l.aziEst)


@pytest.mark.parametrize("n_ants,q,n_slots,zero_s,seed", [(64, 1, 4, True, 11), (64, 2, 2, False, 12), (56, 1, 2, False, 13), (49, 2, 3, True, 14), (63, 1, 2, True, 15)])
def test_native_lazy_grid_matches_materialising_sequence(pkg, n_ants, q, n_slots, zero_s, seed):
    """The regenerating form (echo_range_sl_kernel<., 1, false> + cov_lazy_kernel): 49 / 56 / 63 / 64 antennas (padding antennas of the fourth 16-block masked), one and two
    LoS targets, zero-filled 'S' slots, K = 3276 (the last 1024-subcarrier block pair is partial: its partner half is masked)."""
    tg, vel = _targets(q)
    sc = make_scene(n_ants=n_ants, n_slots=n_slots, nrb=273, targets=tg, velocity=vel, seed=seed, zero_s_slots=zero_s, with_noise=False)
    ctx = pkg.Context()
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    d_wave, d_txg = ctx.to_device(sc.tx_wave), ctx.to_device(sc.tx_grid)
    arr, ra, lz, rl = _both(pkg, ctx, sc, rp, cf, d_wave, d_txg, sc.tx_grid.shape, seed=0xA5 + seed, noise_domain="spectral")
    _same_estimates(ra, rl)
    assert rel(rl[1].Ra, ra[1].Ra) < 1e-13 and np.array_equal(rl[1].Ra, rl[1].Ra.conj().T)
    g = arr.numpy()
    assert np.array_equal(lz.numpy(), g)                                   # the materialised descriptor == the array the non-lazy call stored
    x = g.reshape(-1, n_ants, order="F")
    assert rel(rl[1].Ra, (x.conj().T @ x) / x.shape[0]) < 1e-12                          # Ra[a, b] = sum_n conj(G[n, a]) G[n, b] / N  (fft2D.m:106-107)
    ctx.close()


def test_native_lazy_grid_with_padded_symbol_dimension(pkg):
    """txDimension(2) beyond the waveform's whole symbols (monoStaticSensing.m:19-21 zero-pads): the lazy covariance sums the L_whole synthesised columns and divides by K L_out."""
    sc = make_scene(n_ants=64, n_slots=2, nrb=273, targets=((100.0, 20.0, 1.5),), velocity=(7.0,), seed=21, zero_s_slots=False, with_noise=False)
    ctx = pkg.Context()
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    t_cut = sc.T - 3000                                                     # the last symbol is incomplete: 27 whole symbols, txDimension says 28
    d_wave = ctx.to_device(np.asfortranarray(sc.tx_wave[:t_cut]))
    d_txg = ctx.to_device(sc.tx_grid)
    arr, ra, lz, rl = _both(pkg, ctx, sc, rp, cf, d_wave, d_txg, sc.tx_grid.shape, seed=0x51, noise_domain="spectral")
    g = arr.numpy()
    assert g.shape[1] == 28 and not g[:, 27, :].any() and g[:, 26, :].any()
    _same_estimates(ra, rl)
    assert rel(rl[1].Ra, ra[1].Ra) < 1e-13
    assert np.array_equal(lz.numpy(), g)
    ctx.close()


@pytest.mark.parametrize("n_ants,q,kw", [(8, 1, dict(seed=3, noise_domain="spectral")), (64, 3, dict(seed=4, noise_domain="spectral")), (40, 1, dict(seed=5, noise_domain="spectral")),
                                         (64, 1, dict(seed=6)), (16, 2, "injected"), (72, 1, dict(seed=7, noise_domain="spectral"))])
def test_lazy_grid_in_the_context_owned_buffer(pkg, n_ants, q, kw):
    """Shapes / noise modes the regenerating kernels do not cover (8, 16, 40, 72 antennas; three LoS targets; per-sample Philox noise; injected time-domain noise): the grid
    lives in the context's own buffer, the kernels are the array form's -- everything bit for bit, Ra included."""
    tg, vel = _targets(q)
    sc = make_scene(n_ants=n_ants, n_slots=2, nrb=273, targets=tg, velocity=vel, seed=30 + n_ants, zero_s_slots=False, with_noise=(kw == "injected"))
    ctx = pkg.Context()
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    d_wave, d_txg = ctx.to_device(sc.tx_wave), ctx.to_device(sc.tx_grid)
    if kw == "injected":
        kw = dict(noise=ctx.to_device(sc.noise))
    arr, ra, lz, rl = _both(pkg, ctx, sc, rp, cf, d_wave, d_txg, sc.tx_grid.shape, **kw)
    _same_estimates(ra, rl)
    assert np.array_equal(rl[1].Ra, ra[1].Ra)
    assert np.array_equal(lz.numpy(), arr.numpy())
    ctx.close()


def test_lazy_grid_errors(pkg):
    """No lazy grid on the context -> isac_fft2d_submit_cached_dev(rx = NULL) and isac_echo_grid_materialize_dev fail loudly; a later echo call replaces the descriptor."""
    sc = make_scene(n_ants=64, n_slots=2, nrb=273, seed=41, zero_s_slots=False, with_noise=False)
    ctx = pkg.Context()
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    d_wave, d_txg = ctx.to_device(sc.tx_wave), ctx.to_device(sc.tx_grid)
    dims = (C.c_int32 * 3)()
    assert ctx.lib.isac_echo_grid_materialize_dev(ctx.handle, None, dims) == 1            # ISAC_ERR_INVALID_ARG
    lz = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, fuse_fft2d=(rp, cf, d_txg), ctx=ctx, lazy=True, seed=9, noise_domain="spectral")
    assert ctx.lib.isac_echo_grid_materialize_dev(ctx.handle, None, dims) == 0 and tuple(dims) == sc.tx_grid.shape
    with pytest.raises(ValueError):
        pkg.sensing.estimation.fft2D(rp, cf, lz, d_txg)                                   # a lazy grid has no array for the un-cached range stage
    pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, nfft=4096, ctx=ctx, seed=9, noise_domain="spectral")      # a plain echo call: descriptor gone
    with pytest.raises(pkg.IsacError) as ei:
        pkg.sensing.estimation.fft2D(rp, cf, lz, d_txg, reuse_range=True)
    assert ei.value.name == "INVALID_ARG"
    with pytest.raises(pkg.IsacError):
        lz.materialize()
    ctx.close()


def test_lazy_grid_at_the_bench_shape(pkg):
    """BASELINE configs[1] at its stated size (K = 3276, L = 224, A = 64, T = 983 040): the lazy sequence bench.py times against the materialising one -- |rdm|^2 window, all 64
    CFAR lists, every estimate identical; Ra <= 1e-13."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import FULL
    kw = dict(FULL["config2_a64"]); kw["with_noise"] = False
    sc = make_scene(**kw)
    ctx = pkg.Context()
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    d_wave, d_txg = ctx.to_device(sc.tx_wave), ctx.to_device(sc.tx_grid)
    arr, ra, lz, rl = _both(pkg, ctx, sc, rp, cf, d_wave, d_txg, sc.tx_grid.shape, seed=0x5EED0001, noise_domain="spectral")
    assert sc.tx_grid.shape == (3276, 224, 64)
    _same_estimates(ra, rl)
    assert rl[0] is not None and rl[0].rngEst.size >= 1
    assert rel(rl[1].Ra, ra[1].Ra) < 1e-13
    m = lz.materialize()
    assert np.array_equal(m.numpy()[::7, ::5, ::3], arr.numpy()[::7, ::5, ::3])
    ctx.close()
