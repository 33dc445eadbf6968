"""Uplink channel quality from SRS on the device (isac_srs_pmi_select_batch_dev) and rank selection (ri_total_sinr of isac_csi_report_batch_dev) against the oracle
restatements of pmiSelect.m:28-65 / sinrPerSubband.m:12-36 / gNBPhy.m:1033-1058 (oracle/srs.py) and riSelect.m:207-292 (oracle/pmi.py::ri_select):
TPMI per subband, per-RB CQI and the rank exact (integers), subband SINRs / totalSINR <= 1e-10 relative."""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import pytest

import oracle.cqi as OQ
import oracle.pmi as OP
import oracle.srs as OS
from conftest import load_pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


def ul_channel(rng, nrb, r, p, n_sym, taps=5, gain=1.0):
    k = 12 * nrb
    g = (rng.standard_normal((taps, r, p)) + 1j * rng.standard_normal((taps, r, p))) / np.sqrt(2 * taps)
    ph = np.exp(-2j * np.pi * np.outer(np.arange(k), rng.uniform(0, 60, taps)) / 4096)
    h = gain * np.einsum("kt,trp->krp", ph, g)
    return np.ascontiguousarray(np.broadcast_to(h[:, None], (k, n_sym, r, p)))


def srs_mask(nrb, n_sym, comb, rb_lo, rb_hi):
    m = np.zeros((12 * nrb, n_sym), dtype=bool)
    m[12 * rb_lo:12 * rb_hi:comb, :] = True
    return m


@pytest.mark.parametrize("nrb,band,r,p,layers,comb,span,n_sym", [
    (273, 16, 64, 2, 1, 2, (0, 272), 1),      # the reference's carrier and gNB array; SRS over 272 of 273 RBs: fractional last band
    (273, 16, 64, 2, 2, 4, (0, 272), 2),
    (52, 4, 8, 2, 1, 2, (8, 44), 1),          # SRS on part of the carrier: leading / trailing bands take the mean PMI (gNBPhy.m:1035-1040)
    (51, 4, 4, 2, 2, 1, (0, 51), 1),          # 12.75 bands
    (24, 4, 2, 1, 1, 4, (0, 24), 4),          # single SRS port
    (106, 8, 16, 2, 1, 2, (0, 106), 1),
])
def test_srs_report_matches_oracle(pkg, nrb, band, r, p, layers, comb, span, n_sym):
    rng = np.random.default_rng(nrb + 31 * r + layers)
    n_ue = 3
    mask = srs_mask(nrb, n_sym, comb, *span)
    ll, kk = np.nonzero(mask.T)
    ctx = pkg._lib.default_context()
    hs, d_hs, nvars = [], [], []
    for u in range(n_ue):
        h = ul_channel(rng, nrb, r, p, n_sym, gain=[0.05, 0.3, 2.0][u]) * mask[:, :, None, None]
        hs.append(h)
        d_hs.append(ctx.to_device(np.asfortranarray(h[kk, ll, :, :])))
        nvars.append([0.02, 0.1, 0.5][u])
    got = pkg.communication.phyLayer.srsReportBatch(layers, d_hs, kk, nvars, band, nrb, OQ.UPLINK_SINR90PC, ctx=ctx)
    for u in range(n_ue):
        want_pmi, want_sel, want_cqi = OS.srs_report(layers, hs[u], nvars[u], band, nrb, OQ.UPLINK_SINR90PC)
        g_pmi, g_sel, g_cqi = got[u]
        assert np.abs(g_sel - want_sel).max() <= 1e-10 * np.abs(want_sel).max()
        # a subband whose two best TPMIs tie within rounding is order-of-addition-defined: none expected on these channels, so exact
        assert np.array_equal(g_pmi, want_pmi), (u, g_pmi, want_pmi)
        assert np.array_equal(g_cqi, want_cqi), (u, g_cqi, want_cqi)
        assert g_cqi.min() >= 1 and g_cqi.shape == (nrb,)
    assert len({tuple(g[2]) for g in got}) > 1                               # the three UEs do differ


def test_pmi_select_function_signature(pkg):
    """pmiSelect(nlayers, hest, noiseest, bandSize) on a [K x L x R x P] array: NaN where the reference's function leaves NaN, (NaN, NaN) for noiseest == 0."""
    rng = np.random.default_rng(5)
    nrb = 24
    mask = srs_mask(nrb, 2, 2, 4, 16)
    h = ul_channel(rng, nrb, 4, 2, 2) * mask[:, :, None, None]
    want, want_sinr, _ = OS.pmi_select(1, h, 0.05, 4)
    got, got_sel = pkg.communication.phyLayer.pmiSelect(1, h, 0.05, 4)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])
    ok = ~np.isnan(want)
    assert np.allclose(got_sel[ok], want_sinr[ok, want[ok].astype(int)], rtol=1e-10, atol=0)
    assert np.isnan(pkg.communication.phyLayer.pmiSelect(1, h, 0.0, 4)[0])


def test_pusch_codebook_matches_oracle(pkg):
    for layers, ports in ((1, 1), (1, 2), (2, 2)):
        w = pkg.communication.phyLayer.puschCodebook(layers, ports)
        assert w.shape[2] == OS.max_pusch_tpmi(layers, ports) + 1
        for t in range(w.shape[2]):
            assert np.abs(w[:, :, t] - OS.pusch_codebook(layers, ports, t)).max() < 2.3e-16
    with pytest.raises(Exception):
        pkg.communication.phyLayer.puschCodebook(1, 4)


@pytest.mark.parametrize("nrb,ports,panel,sb,pmimode", [(273, 4, (2, 1), 16, "Subband"), (52, 8, (2, 2), 8, "Subband"), (24, 2, (1, 1), 4, "Wideband"), (52, 4, (2, 1), 4, "Wideband")])
def test_rank_selection_matches_oracle(pkg, nrb, ports, panel, sb, pmimode):
    rng = np.random.default_rng(nrb + ports)
    carrier = SimpleNamespace(NSizeGrid=nrb, NStartGrid=0, SymbolsPerSlot=14)
    rep = SimpleNamespace(NSizeBWP=nrb, NStartBWP=0, PanelDimensions=panel, CodebookMode=1, PMIMode=pmimode, CQIMode="Subband", SubbandSize=sb)
    k = np.concatenate([[12 * rb + 1, 12 * rb + 2] for rb in range(nrb)])
    l = np.ones_like(k)
    csirs = SimpleNamespace(k=k, l=l)
    ctx = pkg._lib.default_context()
    hs, nvars = [], []
    for u in range(6):
        k_all = 12 * nrb
        if u % 3 == 0:                                                     # rank one: the second layer is empty
            a = rng.standard_normal((2, 1)) + 1j * rng.standard_normal((2, 1))
            b = rng.standard_normal((1, ports)) + 1j * rng.standard_normal((1, ports))
            ph = np.exp(-2j * np.pi * np.arange(k_all) * rng.uniform(0, 30) / 4096)
            h = ph[:, None, None] * (a @ b)[None]
        elif u % 3 == 2:                                                   # matched to a two-layer codeword: both layers strong
            w2 = OP.type1_single_panel_codebook(panel, 1, 2, ports)
            e = w2[:, :, u % w2.shape[2], w2.shape[3] // 2, 0, 0]
            h0 = 4.0 * e.conj().T + 0.05 * (rng.standard_normal((2, ports)) + 1j * rng.standard_normal((2, ports)))
            ph = np.exp(-2j * np.pi * np.arange(k_all) * rng.uniform(0, 30) / 4096)
            h = ph[:, None, None] * h0[None]
        else:
            g = (rng.standard_normal((4, 2, ports)) + 1j * rng.standard_normal((4, 2, ports))) / np.sqrt(8)
            ph = np.exp(-2j * np.pi * np.outer(np.arange(k_all), rng.uniform(0, 40, 4)) / 4096)
            h = np.einsum("kt,trp->krp", ph, g) * [1.0, 3.0, 10.0][u % 3]
        hs.append(np.ascontiguousarray(np.broadcast_to(h[:, None], (k_all, 14, 2, ports))))
        nvars.append([0.5, 0.05, 0.002][u % 3])
    d_hs = [ctx.to_device(np.asfortranarray(h[k - 1, l - 1, :, :])) for h in hs]
    got = pkg.communication.phyLayer.riSelectBatch(carrier, csirs, rep, d_hs, nvars, OQ.DOWNLINK_SINR90PC, ctx=ctx)
    ranks = []
    for u in range(6):
        ri, pmi, total = OP.ri_select(rep, hs[u], k, l, nvars[u])
        # each rank's total on the device against the oracle's
        for rank in (1, 2):
            _, _, _, pinfo = pkg.communication.phyLayer.cqiSelect(carrier, csirs, rep, rank, d_hs[u], nvars[u], OQ.DOWNLINK_SINR90PC, ctx=ctx)
            assert abs(pinfo.RITotalSINR - total[rank - 1]) <= 1e-10 * max(1.0, abs(total[rank - 1])), (u, rank, pinfo.RITotalSINR, total)
        if abs(abs(total[1] - total[0]) - 0.1) < 1e-9:
            continue                                                        # on the 0.1 threshold: rounding-defined
        assert got[u][0] == ri, (u, got[u][0], ri, total)
        assert np.array_equal(got[u][2].i1, pmi.i1) and np.array_equal(got[u][2].i2, pmi.i2)
        want_cqi = OP.cqi_select(rep, int(ri), hs[u], k, l, nvars[u], OQ.DOWNLINK_SINR90PC)[0]
        assert np.array_equal(got[u][1], want_cqi, equal_nan=True)
        ranks.append(ri)
    assert 1 in ranks and 2 in ranks
    # the single-UE signature
    ri1, pmi1 = pkg.communication.phyLayer.riSelect(carrier, csirs, rep, hs[1], nvars[1])
    assert ri1 == got[1][0] and np.array_equal(pmi1.i1, got[1][2].i1)


@pytest.mark.parametrize("profile", ["CDL-A", "CDL-D"])
def test_uplink_channel_estimate_and_srs_report_at_config5_shape(pkg, profile):
    """The gNB's measurement as bench.py's config 5 steps it: the perfect UL estimate of a 2 -> 64 channel at every subcarrier (isac_cdl_csi_estimate_batch_dev with the
    69 KB LDS shape) against the oracle's frequency response, then the SRS report on it (273 PRBs, 16-PRB subbands, 64 receive elements) against oracle/srs.py."""
    import oracle.cdl as OC
    ctx = pkg.default_context()
    CM = pkg.communication.channelModels
    fs, nrb, rx_size = 122.88e6, 273, (4, 8, 2, 1, 1)
    k1 = np.arange(1, 12 * nrb + 1)
    chans = [CM.CDLChannel(profile, 300e-9, 3.5e9, (1, 1, 2, 1, 1), rx_size, fs, Seed=90 + u) for u in range(3)]
    times = [0.0, 0.0123, 0.31]
    outs = CM.csiEstimateBatch(chans, k1, 12 * nrb, 30e3, 2, ctx=ctx, times=times)
    nvars = [3e-3, 0.05, 1.0]
    got = pkg.communication.phyLayer.srsReportBatch(1, outs, k1 - 1, nvars, 16, nrb, OQ.UPLINK_SINR90PC, ctx=ctx)
    for u in range(3):
        cfg = OC.cdl_config(profile, 3.5e9, (1, 1, 2, 1, 1), rx_size, fs, seed=90 + u)
        want_h = OC.freq_response(cfg, times[u], k1, 12 * nrb, 30e3, 2)
        h = outs[u].numpy()
        assert h.shape == want_h.shape == (12 * nrb, 64, 2)
        assert np.abs(h - want_h).max() <= 1e-10 * np.abs(want_h).max()
        want_pmi, want_sel, want_cqi = OS.srs_report(1, want_h[:, None, :, :], nvars[u], 16, nrb, OQ.UPLINK_SINR90PC)
        assert np.abs(got[u][1] - want_sel).max() <= 1e-9 * np.abs(want_sel).max()
        assert np.array_equal(got[u][0], want_pmi) and np.array_equal(got[u][2], want_cqi), (u, got[u][0], want_pmi)
