"""PDSCH precoding (prgPrecode.m:53-144) and the device-side CSI channel estimate of a CDL channel: the two config-5 inputs that were host-made until round 5."""
from __future__ import annotations

import numpy as np
import pytest

from conftest import load_pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


@pytest.mark.parametrize("nrb,n_sym,nu,P,nprg,nstart", [(52, 14, 1, 4, 1, 0), (52, 14, 2, 8, 26, 0), (273, 14, 4, 64, 137, 0), (24, 7, 3, 16, 5, 3), (106, 14, 2, 32, 53, 1)])
def test_prg_precode_matches_oracle(pkg, nrb, n_sym, nu, P, nprg, nstart):
    """isac_prg_precode_dev through the reference's index-list signature against the loop-for-loop restatement (oracle/precode.py): PDSCH-like RE sets (whole
    symbols minus a DM-RS comb), 1-4 layers, wideband and per-PRG precoders, a carrier that does not start at CRB 0 -- antsym <= 1e-12, antind exact."""
    import oracle.precode as OPR
    rng = np.random.default_rng(nrb + nu + P)
    K = 12 * nrb
    k = np.arange(K)
    re = np.concatenate([(k if l % 4 else k[k % 2 == 1]) + K * l for l in range(1, n_sym)])          # symbol 0 empty, every 4th symbol a comb
    portind = (re[:, None] + K * n_sym * np.arange(nu)[None, :] + 1).astype(np.int64)
    portsym = (rng.choice([-1.0, 1.0], portind.shape) + 1j * rng.choice([-1.0, 1.0], portind.shape)) / np.sqrt(2.0)
    F = (rng.standard_normal((nu, P, nprg)) + 1j * rng.standard_normal((nu, P, nprg))) / np.sqrt(P)
    want_sym, want_ind = OPR.prg_precode((K, n_sym), nstart, portsym, portind, F)
    got_sym, got_ind = pkg.communication.phyLayer.prgPrecode((K, n_sym), nstart, portsym, portind, F)
    assert np.array_equal(got_ind, want_ind)
    assert np.abs(got_sym - want_sym).max() <= 1e-12 * np.abs(want_sym).max()
    # dense device-resident form == scattering the index-list result
    ctx = pkg.default_context()
    layers = np.zeros(K * n_sym * nu, dtype=np.complex128)
    layers[portind.reshape(-1) - 1] = portsym.reshape(-1)
    grid = pkg.communication.phyLayer.prgPrecodeGrid(ctx.to_device(layers.reshape((K, n_sym, nu), order="F")), F, nstart).numpy()
    dense = np.zeros(K * n_sym * P, dtype=np.complex128)
    dense[want_ind.reshape(-1) - 1] = want_sym.reshape(-1)
    assert np.abs(grid.reshape(-1, order="F") - dense).max() <= 1e-12 * np.abs(want_sym).max()


@pytest.mark.parametrize("profile,tx_size,t0", [("CDL-A", (4, 8, 2, 1, 1), 0.0), ("CDL-D", (4, 8, 2, 1, 1), 0.0123), ("CDL-A", (1, 2, 2, 1, 1), 1.0 / 640 + 1e-7)])
def test_cdl_freq_response_on_device_matches_oracle(pkg, profile, tx_size, t0):
    """CDLChannel.freq_response_device (isac_cdl_path_gains_dev + isac_cdl_freq_response_dev): the CSI-RS channel estimate of a channel time formed entirely on the
    device, against the oracle's evaluation of the same sample-and-hold gain block (<= 1e-10)."""
    import oracle.cdl as OC
    ctx = pkg.default_context()
    fs, nrb = 122.88e6, 273
    cfg = OC.cdl_config(profile, 3.5e9, tx_size, (1, 1, 2, 1, 1), fs)
    ch = pkg.communication.channelModels.CDLChannel(profile, 300e-9, 3.5e9, tx_size, (1, 1, 2, 1, 1), fs)
    ch.time = t0
    k = np.concatenate([[12 * r + 1, 12 * r + 2] for r in range(nrb)])
    got = ch.freq_response_device(k, 12 * nrb, 30e3, 4, ctx).numpy()
    want = OC.freq_response(cfg, t0, k, 12 * nrb, 30e3, 4)
    assert got.shape == want.shape == (k.size, 2, 4)
    assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max()
    # a second call at a later time (other gain block) reuses the cached tables
    ch.time = t0 + 0.01
    got2 = ch.freq_response_device(k, 12 * nrb, 30e3, 4, ctx).numpy()
    assert np.abs(got2 - OC.freq_response(cfg, t0 + 0.01, k, 12 * nrb, 30e3, 4)).max() <= 1e-10 * np.abs(want).max()


def test_csi_estimate_batch_equals_single_calls_and_oracle(pkg):
    """isac_cdl_csi_estimate_batch_dev: six UEs of one delay profile (different seeds, channel times in different gain blocks) in one launch -- each estimate
    equals the two-step single-UE evaluation (path gains, then frequency response) to rounding and the oracle to 1e-10."""
    import oracle.cdl as OC
    ctx = pkg.default_context()
    CM = pkg.communication.channelModels
    fs, nrb = 122.88e6, 273
    k = np.concatenate([[12 * r + 1, 12 * r + 2] for r in range(nrb)])
    for profile in ("CDL-A", "CDL-D"):
        chans = [CM.CDLChannel(profile, 300e-9, 3.5e9, (4, 8, 2, 1, 1), (1, 1, 2, 1, 1), fs, Seed=70 + u) for u in range(6)]
        times = [0.0, 0.0031, 0.2, 1.0 / 640 + 1e-7, 0.05, 0.9]
        outs = CM.csiEstimateBatch(chans, k, 12 * nrb, 30e3, 4, ctx=ctx, times=times)
        for u, (ch, t) in enumerate(zip(chans, times)):
            single = ch.freq_response_device(k, 12 * nrb, 30e3, 4, ctx, t=ch.snap_time(t)).numpy()
            got = outs[u].numpy()
            assert np.abs(got - single).max() <= 1e-13 * np.abs(single).max()
            cfg = OC.cdl_config(profile, 3.5e9, (4, 8, 2, 1, 1), (1, 1, 2, 1, 1), fs, seed=70 + u)
            want = OC.freq_response(cfg, t, k, 12 * nrb, 30e3, 4)
            assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max(), (profile, u)
    with pytest.raises(ValueError):
        CM.csiEstimateBatch([chans[0], CM.CDLChannel("CDL-A", 300e-9, 3.5e9, (4, 8, 2, 1, 1), (1, 1, 2, 1, 1), fs)], k, 12 * nrb, 30e3, 4, ctx=ctx)
