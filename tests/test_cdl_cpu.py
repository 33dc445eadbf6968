"""CDL channel: self-checks of the restated TR 38.901 tables/model (no reference vectors exist) and
agreement of the product's host-side parameter prep with the oracle."""
from __future__ import annotations

import numpy as np
import pytest

import oracle.cdl as OC
from conftest import load_pkg


@pytest.mark.parametrize("profile,n_paths,k_db", [("CDL-A", 23, None), ("CDL-D", 13, 13.3)])
def test_tables_self_consistency(profile, n_paths, k_db):
    cfg = OC.cdl_config(profile, 3.5e9, (1, 4, 2, 1, 1), (1, 1, 2, 1, 1), 15.36e6)
    rays = OC.draw_rays(cfg)
    assert rays.aod.shape == (n_paths, 20)
    assert rays.power.sum() == pytest.approx(1.0, rel=1e-14)              # NormalizePathGains
    if k_db is not None:                                                  # TR 38.901 Table 7.7.1-4: K = 13.3 dB
        assert 10 * np.log10(rays.power[0] / rays.power[1]) == pytest.approx(k_db, abs=1e-9)
    d = OC.path_delays(cfg)
    assert d.shape == (n_paths,) and d[0] == 0 and np.all(d >= 0)
    assert d.max() == pytest.approx(300e-9 * (9.6586 if profile == "CDL-A" else 12.525))
    info = OC.channel_info(cfg)
    assert info.MaxChannelDelay == int(np.ceil(d.max() * 15.36e6)) + 7
    # ray offsets: zero mean, unit rms (Table 7.5-3 is normalised to 1 deg rms)
    assert abs(OC.RAY_OFFSETS.mean()) < 1e-12 and np.sqrt((OC.RAY_OFFSETS ** 2).mean()) == pytest.approx(1.0, abs=2e-3)
    # couplings are permutations of the 20 rays
    assert np.allclose(np.sort((rays.aoa - rays.aoa.mean(axis=1, keepdims=True)), axis=1),
                       np.sort((rays.aod - rays.aod.mean(axis=1, keepdims=True)) * (11.0 if profile == "CDL-A" else 8.0) / 5.0, axis=1))
    g, shift = OC.filter_taps(cfg)
    assert g.shape == (n_paths, 16) and np.allclose(g.sum(axis=1), 1.0, atol=2e-3)    # unit DC gain fractional-delay filters
    assert np.all(shift == np.floor(d * 15.36e6))


def test_element_pattern_and_positions():
    ft, fp = OC.field_pattern(90.0, 0.0, "38.901", 45.0)                   # boresight: 8 dBi, +45 deg slant
    assert ft == pytest.approx(10 ** (8 / 20) * np.cos(np.pi / 4)) and fp == pytest.approx(10 ** (8 / 20) * np.sin(np.pi / 4))
    ft, _ = OC.field_pattern(90.0, 180.0, "38.901", 0.0)                   # back lobe: -30 dB + 8 dBi
    assert ft == pytest.approx(10 ** ((8 - 30) / 20))
    pos, pol = OC.element_positions((1, 4, 2, 1, 1))
    assert pos.shape == (8, 3) and pol.tolist() == [0, 0, 0, 0, 1, 1, 1, 1]
    assert pos[:4, 1].tolist() == [0.0, 0.5, 1.0, 1.5] and np.all(pos[:, 2] == 0)


def test_average_channel_power_is_normalised():
    """Ensemble check over seeds: isotropic Rx, E|H[s,u]|^2 summed over paths == Tx element gain averaged over the
    departure angles weighted by path power (unit total power)."""
    acc = 0.0
    n_seed = 12
    for seed in range(n_seed):
        cfg = OC.cdl_config("CDL-A", 3.5e9, (1, 1, 1, 1, 1), (1, 1, 1, 1, 1), 15.36e6, seed=seed)
        cfg.TxElement = "isotropic"
        cfg.TxPolAngles = (0.0,)
        cfg.RxPolAngles = (0.0,)
        h = OC.path_gains(cfg, 0.0)
        acc += np.sum(np.abs(h[:, 0, 0]) ** 2)
    # co-polar (theta-theta) term has unit mean power, cross terms 1/kappa are filtered out by the polarisation choice
    assert acc / n_seed == pytest.approx(1.0, rel=0.25)


@pytest.mark.parametrize("profile", ["CDL-A", "CDL-D"])
def test_host_mirror_matches_oracle(profile):
    pkg = load_pkg()
    cfg = OC.cdl_config(profile, 3.5e9, (1, 4, 2, 1, 1), (1, 1, 2, 1, 1), 15.36e6)
    ch = pkg.communication.channelModels.CDLChannel(profile, 300e-9, 3.5e9, (1, 4, 2, 1, 1), (1, 1, 2, 1, 1), 15.36e6)
    for t in (0.0, 0.0123):
        ho, hp = OC.path_gains(cfg, t), ch.path_gains(t)
        assert np.abs(ho - hp).max() <= 1e-13 * np.abs(ho).max()
    go, so = OC.filter_taps(cfg)
    gp, sp = ch.filter_taps()
    assert np.array_equal(go, gp) and np.array_equal(so, sp)
    assert ch.info().MaxChannelDelay == OC.channel_info(cfg).MaxChannelDelay
    with pytest.raises(ValueError):
        pkg.communication.channelModels.CDLChannel("CDL-B")


def test_block_plan_of_a_static_channel_and_shared_taps():
    """Host mirror: MaximumDopplerShift = 0 (a valid nrCDLChannel configuration) is ONE gain block at the current channel time, not a division by zero;
    the filter taps shared between channels of one delay profile are read-only."""
    pkg = load_pkg()
    CM = pkg.communication.channelModels
    ch = CM.CDLChannel("CDL-A", 300e-9, 3.5e9, (1, 4, 2, 1, 1), (1, 1, 2, 1, 1), 15.36e6, MaximumDopplerShift=0.0)
    ch.time = 0.37
    assert ch.block_plan(7680) == ([0.37], [0])
    mv = CM.CDLChannel("CDL-A", 300e-9, 3.5e9, (1, 4, 2, 1, 1), (1, 1, 2, 1, 1), 15.36e6)
    t_snap, starts = mv.block_plan(7680)
    assert starts[0] == 0 and len(t_snap) == len(starts) >= 1
    g, sh = mv.filter_taps()
    with pytest.raises(ValueError):
        g[0, 0] = 1.0
    with pytest.raises(ValueError):
        sh[0] = 1


def test_apply_cdl_is_linear_and_causal():
    cfg = OC.cdl_config("CDL-D", 3.5e9, (1, 2, 2, 1, 1), (1, 1, 2, 1, 1), 15.36e6)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((600, 4)) + 1j * rng.standard_normal((600, 4))
    y = OC.apply_cdl(cfg, x)
    assert y.shape == (600, 2)
    assert np.allclose(OC.apply_cdl(cfg, 2 * x), 2 * y)
    x2 = x.copy(); x2[300:] = 0
    assert np.allclose(OC.apply_cdl(cfg, x2)[:300], y[:300])             # causal: later inputs do not change earlier outputs
    imp = np.zeros((200, 4), complex); imp[0, 0] = 1
    yi = OC.apply_cdl(cfg, imp)
    assert np.abs(yi[:1]).max() < np.abs(yi[7]).max()                   # energy arrives after the 7-sample filter delay


@pytest.mark.parametrize("profile,tx,rx,t_len,t0", [
    ("CDL-A", (1, 1, 2, 1, 1), (1, 4, 2, 1, 1), 1500, 1.0 / 640 - 700 / 15.36e6),     # uplink shape, across a path-gain refresh
    ("CDL-D", (1, 4, 2, 1, 1), (1, 1, 2, 1, 1), 1200, 0.01),                           # downlink shape
    ("CDL-A", (1, 1, 2, 1, 1), (1, 1, 2, 1, 1), 900, 0.0)])
def test_apply_orders_agree(profile, tx, rx, t_len, t0):
    """apply_cdl's two evaluation orders (contract-then-filter: the literal formula; filter-then-contract: what makes the 2 -> 64 uplink case of config 5
    affordable for the oracle) are the same sum re-associated: equal to rounding, ragged waveform, a gain refresh inside."""
    cfg = OC.cdl_config(profile, 3.5e9, tx, rx, 15.36e6)
    rng = np.random.default_rng(t_len)
    nt = int(np.prod(tx))
    x = rng.standard_normal((t_len, nt)) + 1j * rng.standard_normal((t_len, nt))
    a = OC.apply_cdl(cfg, x, t0, order="contract_first")
    b = OC.apply_cdl(cfg, x, t0, order="filter_first")
    assert np.abs(a - b).max() <= 1e-13 * np.abs(a).max()
    c = OC.apply_cdl(cfg, x, t0)
    assert np.array_equal(c, b if int(np.prod(rx)) > nt else a)
