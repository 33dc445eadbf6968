"""CSI report on the device (isac_csi_report_dev: exhaustive Type-I PMI search + subband SINR -> CQI) against the oracle restatement of
dlPMISelect.m:385-500 / cqiSelect.m:500-687 (oracle/pmi.py): PMI indices, CQI indices and differential values exact (integers),
subband SINRs <= 1e-10.  Exact ties after round(., 4) (dlPMISelect.m:449) resolve to the first entry on both sides; a channel whose total
lands within 1e-10 of a rounding boundary is skipped as rounding-defined, not tolerated."""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import pytest

import oracle.cqi as OQ
import oracle.pmi as OP
from conftest import load_pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


def channel(rng, nrb, nr, p, taps=4):
    """Frequency-selective [K x 14 x nRx x P] channel: a few random taps -> smooth variation over the subcarriers."""
    k = 12 * nrb
    g = (rng.standard_normal((taps, nr, p)) + 1j * rng.standard_normal((taps, nr, p))) / np.sqrt(2 * taps)
    ph = np.exp(-2j * np.pi * np.outer(np.arange(k), rng.uniform(0, 40, taps)) / 4096)
    h = np.einsum("kt,trp->krp", ph, g)
    return np.ascontiguousarray(np.broadcast_to(h[:, None], (k, 14, nr, p)))


def same(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


@pytest.mark.parametrize("nrb,ports,panel,layers,mode,sbsize,pmimode,cqimode,nstart", [
    (273, 4, (2, 1), 1, 1, 16, "Subband", "Subband", 0),        # the reference's own configuration (setupCSIRS.m:5-23)
    (273, 4, (2, 1), 2, 1, 32, "Subband", "Subband", 0),
    (52, 4, (2, 1), 2, 2, 8, "Subband", "Subband", 5),          # unaligned BWP start: short first / last subband
    (52, 8, (2, 2), 1, 1, 4, "Wideband", "Subband", 0),
    (52, 8, (4, 1), 2, 1, 8, "Subband", "Wideband", 0),
    (24, 2, (1, 1), 1, 1, 4, "Wideband", "Wideband", 0),
    (24, 2, (1, 1), 2, 1, 4, "Subband", "Subband", 0),
    (20, 16, (4, 2), 2, 1, 4, "Subband", "Subband", 0),         # < 24 PRBs: one subband whatever the mode
])
def test_csi_report_matches_oracle(pkg, nrb, ports, panel, layers, mode, sbsize, pmimode, cqimode, nstart):
    rng = np.random.default_rng(nrb * 7 + ports + layers)
    carrier = SimpleNamespace(NSizeGrid=nrb, NStartGrid=0, SymbolsPerSlot=14)
    rep = SimpleNamespace(NSizeBWP=nrb, NStartBWP=nstart, PanelDimensions=panel, CodebookMode=mode, PMIMode=pmimode, CQIMode=cqimode, SubbandSize=sbsize)
    h = channel(rng, nrb, 2, ports) * 3.0
    k = np.concatenate([[12 * r + 1, 12 * r + 2] for r in range(nrb)])          # row-5-like: two REs of the first port per RB, symbol 1
    l = np.ones_like(k)
    if nrb == 52 and ports == 4:                                                    # CSI-RS absent from the third subband: i2 / CQI NaN there
        sb = OP.subband_info(pmimode, nstart, nrb, sbsize)
        lo = sum(sb.SubbandSizes[:2]) * 12
        keep = (k <= lo) | (k > lo + sb.SubbandSizes[2] * 12)
        k, l = k[keep], l[keep]
    nvar = 0.02
    csirs = SimpleNamespace(k=k, l=l)
    want_cqi, want_pmi, want_ci, want_pi = OP.cqi_select(rep, layers, h, k, l, nvar, OQ.DOWNLINK_SINR90PC)
    tot = np.nansum(want_pi.SINRPerRE, axis=(0, 1, 2)).reshape(-1, order="F")
    got_cqi, got_pmi, got_ci, got_pi = pkg.communication.phyLayer.cqiSelect(carrier, csirs, rep, layers, h, nvar, OQ.DOWNLINK_SINR90PC)
    got_tot = got_pi.TotalSINR.reshape(-1, order="F")
    assert np.abs(got_tot - tot).max() <= 1e-10 * np.abs(tot).max()
    if not np.array_equal(OP.matlab_round4(got_tot), OP.matlab_round4(tot)):      # a total within 1e-10 of a round(., 4) boundary
        pytest.skip("a total SINR sits on a rounding boundary of round(., 4): the PMI is rounding-defined")
    assert same(got_pmi.i1, want_pmi.i1) and same(got_pmi.i2, want_pmi.i2), (got_pmi, want_pmi)
    assert same(got_cqi, want_cqi), (got_cqi, want_cqi)
    assert same(got_ci.SubbandCQI, want_ci.SubbandCQI)
    a, b = got_ci.SINRPerSubbandPerCW, want_ci.SINRPerSubbandPerCW
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.abs(a[~np.isnan(a)] - b[~np.isnan(b)]).max() <= 1e-10 * np.nanmax(np.abs(b))
    assert not np.all(np.isnan(got_cqi)) and got_cqi[0] >= 1


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_csi_report_irregular_re_sets(pkg, seed):
    """The subband reduction walks host-built per-subband RE lists: REs in SHUFFLED order, on three different symbols with unequal counts per (subband, symbol),
    a ragged subset of the RBs -- PMI / CQI integers exact, subband SINRs <= 1e-10 against the oracle (mean over symbols of the per-symbol means)."""
    rng = np.random.default_rng(100 + seed)
    nrb, ports, layers = 52, 4, 1 + seed % 2
    carrier = SimpleNamespace(NSizeGrid=nrb, NStartGrid=0, SymbolsPerSlot=14)
    rep = SimpleNamespace(NSizeBWP=nrb, NStartBWP=3, PanelDimensions=(2, 1), CodebookMode=1, PMIMode="Subband", CQIMode="Subband", SubbandSize=8)
    k_all = 12 * rng.choice(nrb, size=nrb - 7, replace=False) + rng.integers(1, 12, nrb - 7)       # one RE in most RBs, seven RBs without CSI-RS
    k = np.concatenate([k_all, k_all[: nrb // 2] + 1, k_all[: nrb // 4]])                             # unequal counts per symbol
    l = np.concatenate([np.full(k_all.size, 2), np.full(nrb // 2, 6), np.full(nrb // 4, 11)])
    perm = rng.permutation(k.size)
    k, l = k[perm], l[perm]
    h = np.ascontiguousarray(channel(rng, nrb, 2, ports) * (1.0 + 0.3 * rng.standard_normal((1, 14, 1, 1))))   # symbol-dependent gain: the symbol means differ
    nvar = 0.05
    want_cqi, want_pmi, want_ci, want_pi = OP.cqi_select(rep, layers, h, k, l, nvar, OQ.DOWNLINK_SINR90PC)
    tot = np.nansum(want_pi.SINRPerRE, axis=(0, 1, 2)).reshape(-1, order="F")
    got_cqi, got_pmi, got_ci, got_pi = pkg.communication.phyLayer.cqiSelect(carrier, SimpleNamespace(k=k, l=l), rep, layers, h, nvar, OQ.DOWNLINK_SINR90PC)
    got_tot = got_pi.TotalSINR.reshape(-1, order="F")
    assert np.abs(got_tot - tot).max() <= 1e-10 * np.abs(tot).max()
    if not np.array_equal(OP.matlab_round4(got_tot), OP.matlab_round4(tot)):
        pytest.skip("a total SINR sits on a rounding boundary of round(., 4): the PMI is rounding-defined")
    assert same(got_pmi.i1, want_pmi.i1) and same(got_pmi.i2, want_pmi.i2), (got_pmi, want_pmi)
    assert same(got_cqi, want_cqi), (got_cqi, want_cqi)
    a, b = got_ci.SINRPerSubbandPerCW, want_ci.SINRPerSubbandPerCW
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.abs(a[~np.isnan(a)] - b[~np.isnan(b)]).max() <= 1e-10 * np.nanmax(np.abs(b))


def test_csi_report_batch_equals_single_reports(pkg):
    """isac_csi_report_batch_dev: six UEs of one cell (different channels and noise variances, one of them with an all-NaN-prone tiny channel) in one
    call -- every field of every UE's report equals the single-UE call's (same kernels, same host half, one synchronisation instead of six)."""
    from conftest import load_pkg as _lp
    rng = np.random.default_rng(20)
    nrb, ports, layers = 273, 4, 1
    carrier = SimpleNamespace(NSizeGrid=nrb, NStartGrid=0, SymbolsPerSlot=14)
    rep = SimpleNamespace(NSizeBWP=nrb, NStartBWP=0, PanelDimensions=(2, 1), CodebookMode=1, PMIMode="Subband", CQIMode="Subband", SubbandSize=16)
    k = np.concatenate([[12 * r + 1, 12 * r + 2] for r in range(nrb)])
    l = np.ones_like(k)
    csirs = SimpleNamespace(k=k, l=l)
    ctx = pkg.default_context()
    hs, nvars = [], []
    for u in range(6):
        h = channel(rng, nrb, 2, ports) * (3.0 if u != 4 else 1e-3)
        hs.append(ctx.to_device(np.asfortranarray(h[k - 1, l - 1, :, :])))
        nvars.append(0.02 * (1 + u))
    batch = pkg.communication.phyLayer.cqiSelectBatch(carrier, csirs, rep, layers, hs, nvars, OQ.DOWNLINK_SINR90PC, ctx=ctx)
    assert len(batch) == 6
    for u in range(6):
        cqi1, pmi1, ci1, _ = pkg.communication.phyLayer.cqiSelect(carrier, csirs, rep, layers, hs[u], nvars[u], OQ.DOWNLINK_SINR90PC, ctx=ctx)
        cqi, pmi, ci = batch[u]
        assert same(cqi, cqi1) and same(pmi.i1, pmi1.i1) and same(pmi.i2, pmi1.i2) and same(ci.SubbandCQI, ci1.SubbandCQI)
        assert same(ci.SINRPerSubbandPerCW, ci1.SINRPerSubbandPerCW)
    assert len({tuple(np.nan_to_num(b[0], nan=-1.0)) for b in batch}) > 1          # the UEs do report different things


def test_csi_report_without_csirs_is_all_nan(pkg):
    rep = SimpleNamespace(NSizeBWP=52, NStartBWP=0, PanelDimensions=(2, 1), CodebookMode=1, PMIMode="Subband", CQIMode="Subband", SubbandSize=8)
    carrier = SimpleNamespace(NSizeGrid=52, NStartGrid=0, SymbolsPerSlot=14)
    h = np.zeros((624, 14, 2, 4), dtype=np.complex128)
    cqi, pmi, ci, _ = pkg.communication.phyLayer.cqiSelect(carrier, SimpleNamespace(k=np.zeros(0, int), l=np.zeros(0, int)), rep, 1, h, 0.1, OQ.DOWNLINK_SINR90PC)
    assert cqi.size == 8 and np.all(np.isnan(cqi)) and np.all(np.isnan(pmi.i1)) and np.all(np.isnan(pmi.i2)) and pmi.i2.size == 7
