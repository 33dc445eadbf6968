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
