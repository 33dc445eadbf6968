"""Scheduling choices of the pipelined host API must not change a single result: ISAC_OPT_WIDE_ORDER (the covariance on the main stream,
every narrow kernel on the second) and isac_ctx_share_streams (several contexts -- one per CPI in flight -- on one pair of streams, collected
per CPI through an event, not a stream synchronisation) against the default two-streams-per-context order and against the oracle."""
from __future__ import annotations

import numpy as np
import pytest

import oracle as O
from conftest import load_pkg, make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


def _scene(seed, n_ants=4, **kw):
    return make_scene(n_ants=n_ants, n_slots=4, nrb=273, targets=((100.0 + 7.0 * seed, 20.0, 1.5), (250.0, -120.0 + 5.0 * seed, 1.5)),
                      velocity=(7.0, -4.0), seed=seed, **kw)


def _oracle(sc):
    rx = O.mono_static_sensing(sc.tx_wave, sc.tx_grid.shape, sc.carrier, sc.rp, sc.los, sc.noise, nfft=sc.wave.Nfft)
    return rx, O.fft2d(sc.rp, O.cfar2d_config(sc.rp), rx, sc.tx_grid)


def _same(e, want):
    return np.array_equal(e.rngEst, want.rngEst) and np.array_equal(e.velEst, want.velEst) and np.array_equal(e.aziEst, want.aziEst)


@pytest.mark.parametrize("n_ants", [4, 64, 80])
def test_wide_order_gives_the_same_call(pkg, n_ants):
    sc = _scene(3, n_ants=n_ants)
    rx, want = _oracle(sc)
    out = []
    for wide in (False, True):
        c = pkg.Context()
        c.set_wide_order(wide)
        rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
        cf = pkg.sensing.detection.cfar2D(rp)
        e, d = pkg.sensing.estimation.fft2D(rp, cf, c.to_device(rx), c.to_device(sc.tx_grid), return_debug=True)
        assert _same(e, want)
        out.append(d)
    d0, d1 = out
    assert np.array_equal(d0.power_window, d1.power_window) and np.array_equal(d0.Ra, d1.Ra) and np.array_equal(d0.spectrum_db, d1.spectrum_db)
    for a in range(n_ants):
        assert np.array_equal(d0.detections[a], d1.detections[a])


@pytest.mark.parametrize("fused", [True, False])
def test_contexts_on_shared_streams_pipeline(pkg, fused):
    """Six different CPIs in flight on three contexts that share one pair of streams, twice around the ring; every collect returns its own
    CPI's estimates (the oracle's), in submission order, while later CPIs are still queued behind it on the same streams."""
    n_ctx = 3
    ctxs = [pkg.Context() for _ in range(n_ctx)]
    for c in ctxs:
        c.set_wide_order(True)
    for c in ctxs[1:]:
        c.share_streams(ctxs[0])
    scenes = [_scene(s) for s in range(6)]
    wants = [_oracle(sc) for sc in scenes]
    dev = []
    for k, sc in enumerate(scenes):
        c = ctxs[k % n_ctx]
        rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
        dev.append((c, rp, pkg.sensing.detection.cfar2D(rp), c.to_device(sc.tx_wave), c.to_device(sc.tx_grid), c.to_device(sc.noise), c.to_device(wants[k][0])))
    pending = [None] * n_ctx
    got = [None] * len(scenes)

    def collect(slot):
        if pending[slot] is not None:
            got[pending[slot]] = pkg.sensing.estimation.fft2D_collect(ctxs[slot])
            pending[slot] = None

    for k, (c, rp, cf, d_wave, d_tx, d_noise, d_rx) in enumerate(dev):
        slot = k % n_ctx
        collect(slot)
        if fused:
            # echo synthesis (injected AWGN: the parity mode) + range stage on the device, then the cached submit: the whole hot path on the shared streams
            sc = scenes[k]
            echo = pkg.sensing.monoStaticSensing(d_wave, sc.tx_grid.shape, sc.carrier, rp, sc.los, noise=d_noise, nfft=sc.wave.Nfft, ctx=c,
                                                 fuse_fft2d=(rp, cf, d_tx))
            pkg.sensing.estimation.fft2D_submit(rp, cf, echo, d_tx, ctx=c, reuse_range=True)
        else:
            pkg.sensing.estimation.fft2D_submit(rp, cf, d_rx, d_tx, ctx=c)
        pending[slot] = k
    for s in range(n_ctx):
        collect((len(dev) + s) % n_ctx)
    for k in range(len(scenes)):
        assert _same(got[k], wants[k][1]), k
    # back to private streams: the contexts keep working on their own
    for c in ctxs[1:]:
        c.share_streams(None)
    c, rp, cf, _, d_tx, _, d_rx = dev[1]
    assert _same(pkg.sensing.estimation.fft2D(rp, cf, d_rx, d_tx), wants[1][1])


def test_share_streams_refuses_a_pending_context(pkg):
    a, b = pkg.Context(), pkg.Context()
    sc = _scene(1)
    rx, _ = _oracle(sc)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    pkg.sensing.estimation.fft2D_submit(rp, cf, b.to_device(rx), b.to_device(sc.tx_grid), ctx=b)
    with pytest.raises(pkg.IsacError):
        b.share_streams(a)
    pkg.sensing.estimation.fft2D_collect(b)
    b.share_streams(a)
    b.share_streams(None)


def test_csi_report_between_submit_and_collect_keeps_the_pending_cpi(pkg):
    """isac_csi_report[_batch]_dev on a context with a submitted, not yet collected CPI used to copy its results into the very pinned buffer
    the submit's D2H copy fills and isac_fft2d_collect parses.  The CSI path has its own result buffer now: the collect that follows returns the CPI's
    own estimates, and the report equals the one computed on an idle context."""
    from types import SimpleNamespace
    import oracle.cqi as OQ
    c = pkg.Context()
    sc = _scene(2)
    rx, want = _oracle(sc)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    rng = np.random.default_rng(5)
    nrb = 52
    k = np.concatenate([[12 * r + 1, 12 * r + 2] for r in range(nrb)])
    l = np.ones_like(k)
    carrier = SimpleNamespace(NSizeGrid=nrb, NStartGrid=0, SymbolsPerSlot=14)
    rep = SimpleNamespace(NSizeBWP=nrb, NStartBWP=0, PanelDimensions=(2, 1), CodebookMode=1, PMIMode="Subband", CQIMode="Subband", SubbandSize=8)
    hs = [c.to_device(np.asfortranarray((rng.standard_normal((k.size, 2, 4)) + 1j * rng.standard_normal((k.size, 2, 4))) * 2.0)) for _ in range(12)]
    nv = [0.05 * (1 + u) for u in range(12)]
    idle = pkg.communication.phyLayer.cqiSelectBatch(carrier, SimpleNamespace(k=k, l=l), rep, 1, hs, nv, OQ.DOWNLINK_SINR90PC, ctx=c)
    d_rx, d_tx = c.to_device(rx), c.to_device(sc.tx_grid)
    for _ in range(3):
        pkg.sensing.estimation.fft2D_submit(rp, cf, d_rx, d_tx, ctx=c)
        busy = pkg.communication.phyLayer.cqiSelectBatch(carrier, SimpleNamespace(k=k, l=l), rep, 1, hs, nv, OQ.DOWNLINK_SINR90PC, ctx=c)
        assert _same(pkg.sensing.estimation.fft2D_collect(c), want)
        for a, b in zip(idle, busy):
            assert np.array_equal(np.nan_to_num(a[0], nan=-1.0), np.nan_to_num(b[0], nan=-1.0))


def test_reserve_prepares_the_context_and_leaves_no_state_behind(pkg):
    """isac_ctx_reserve: dry runs of monoStaticSensing -> fft2D at the caller's shape on grids the library makes itself.  Afterwards the real
    chain gives the oracle's estimates, nothing of the dry run is cached (a cached fft2D without a preceding fused call is still an error), and a second
    reserve with a wall-clock budget keeps running dry CPIs for about that long."""
    c = pkg.Context()
    sc = _scene(4)
    rx, want = _oracle(sc)
    rp = pkg.sensing.radarParams(sc.cell, sc.carrier, sc.wave)
    cf = pkg.sensing.detection.cfar2D(rp)
    ms = pkg.sensing.reserve(sc.T, sc.tx_grid.shape, sc.carrier, rp, cf, nfft=sc.wave.Nfft, ctx=c)
    assert ms > 0.0
    d_rx, d_tx = c.to_device(rx), c.to_device(sc.tx_grid)
    with pytest.raises(pkg.IsacError):
        pkg.sensing.estimation.fft2D(rp, cf, d_rx, d_tx, ctx=c, reuse_range=True)
    assert _same(pkg.sensing.estimation.fft2D(rp, cf, d_rx, d_tx, ctx=c), want)
    echo = pkg.sensing.monoStaticSensing(c.to_device(sc.tx_wave), sc.tx_grid.shape, sc.carrier, rp, sc.los, noise=c.to_device(sc.noise), nfft=sc.wave.Nfft, ctx=c)
    assert _same(pkg.sensing.estimation.fft2D(rp, cf, echo, d_tx, ctx=c), want)
    ms2 = pkg.sensing.reserve(sc.T, sc.tx_grid.shape, sc.carrier, rp, cf, nfft=sc.wave.Nfft, warm_ms=60.0, ctx=c)
    assert 60.0 <= ms2 < 2000.0
