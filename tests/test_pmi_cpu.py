"""Type-I single-panel codebook: the product's host-side builder (isac_type1sp_codebook, no GPU needed) against the oracle's
loop-for-loop restatement of getPMIType1SinglePanelCodebook (dlPMISelect.m:853-1083) and TS 38.214 properties; the oracle's own
PMI / CQI selection on hand-made channels (known answers)."""
from types import SimpleNamespace

import numpy as np
import pytest

import oracle.cqi as OQ
import oracle.pmi as OP
from conftest import load_pkg


@pytest.mark.parametrize("ports,panel", [(2, (1, 1)), (4, (2, 1)), (8, (2, 2)), (8, (4, 1)), (12, (3, 2)), (16, (4, 2)), (16, (8, 1)), (32, (4, 4)), (32, (16, 1))])
@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("layers", [1, 2])
def test_codebook_matches_reference_restatement(ports, panel, mode, layers):
    pkg = load_pkg()
    rc = SimpleNamespace(PanelDimensions=panel, CodebookMode=mode)
    want = OP.type1_single_panel_codebook(panel, mode, layers, ports)
    got = pkg.communication.phyLayer.type1SinglePanelCodebook(rc, layers, ports)
    assert got.shape == want.shape and np.abs(got - want).max() < 1e-15
    e = got.reshape(ports, layers, -1)
    for i in range(e.shape[2]):                      # TS 38.214: unit total power, orthogonal layers
        g = e[:, :, i].conj().T @ e[:, :, i]
        assert np.abs(g - np.eye(layers) / layers).max() < 1e-14


def test_unsupported_rank_is_an_error():
    pkg = load_pkg()
    with pytest.raises(pkg.IsacError):
        pkg.communication.phyLayer.type1SinglePanelCodebook(SimpleNamespace(PanelDimensions=(2, 1), CodebookMode=1), 3, 4)


def test_oracle_pmi_known_answers():
    """A rank-1 channel along codebook beam (i11 = 3, i2 = 2) must be found; subband info follows TS 38.214 Table 5.2.1.4-2 arithmetic."""
    sb = OP.subband_info("Subband", 0, 52, 8)
    assert sb.NumSubbands == 7 and sb.SubbandSizes == [8, 8, 8, 8, 8, 8, 4]
    sb = OP.subband_info("Subband", 5, 52, 8)
    assert sb.SubbandSizes[0] == 3 and sb.SubbandSizes[-1] == 1 and sum(sb.SubbandSizes) == 52
    assert OP.subband_info("Subband", 0, 20, 4).NumSubbands == 1 and OP.subband_info("Wideband", 0, 273, 16).SubbandSizes == [273]
    assert OP.matlab_round4(np.array([1.23455, -1.23455, 2.00005])).tolist() == [1.2346, -1.2346, 2.0001]
    w = OP.type1_single_panel_codebook((2, 1), 1, 1, 4)
    target = w[:, 0, 2, 3, 0, 0]
    rng = np.random.default_rng(1)
    nrb = 24
    rx = rng.standard_normal(2) + 1j * rng.standard_normal(2)
    h = np.tile((rx[:, None] * target.conj()[None, :])[None, None], (12 * nrb, 14, 1, 1))
    rep = SimpleNamespace(NSizeBWP=nrb, NStartBWP=0, PanelDimensions=(2, 1), CodebookMode=1, PMIMode="Subband", CQIMode="Subband", SubbandSize=4)
    k = np.concatenate([[12 * r + 1, 12 * r + 2] for r in range(nrb)])
    cqi, pmi, ci, _ = OP.cqi_select(rep, 1, h, k, np.ones_like(k), 0.01, OQ.DOWNLINK_SINR90PC)
    assert pmi.i1.tolist() == [4, 1, 1] and np.all(pmi.i2 == 3) and cqi.size == 7 and np.all(cqi[1:] == 0)
    # no CSI-RS in the BWP -> everything NaN
    cqi, pmi, _, _ = OP.cqi_select(rep, 1, h, np.zeros(0, int), np.zeros(0, int), 0.01, OQ.DOWNLINK_SINR90PC)
    assert np.all(np.isnan(cqi)) and np.all(np.isnan(pmi.i1)) and np.all(np.isnan(pmi.i2))


def test_prg_precode_restatement_equals_dense_formula():
    """oracle/precode.py (prgPrecode.m:53-144 loop for loop) against the closed form grid[k, l, :] = layers[k, l, :] F(:, :, prg(k)), prg from getPRGSet."""
    import oracle.precode as OPR
    rng = np.random.default_rng(3)
    for nrb, L, nu, P, nprg, nstart in ((24, 4, 2, 8, 5, 3), (52, 3, 1, 4, 1, 0), (10, 2, 4, 16, 10, 0)):
        K = 12 * nrb
        re = np.sort(rng.choice(K * L, size=K * L // 2, replace=False))
        portind = re[:, None] + K * L * np.arange(nu)[None, :] + 1
        portsym = rng.standard_normal(portind.shape) + 1j * rng.standard_normal(portind.shape)
        F = rng.standard_normal((nu, P, nprg)) + 1j * rng.standard_normal((nu, P, nprg))
        sym, ind = OPR.prg_precode((K, L), nstart, portsym, portind, F)
        prg = OPR.get_prg_set(nrb, nstart, nprg)
        assert prg.min() >= 1 and prg.max() <= nprg and np.all(np.diff(prg) >= 0)
        want = np.stack([portsym[i] @ F[:, :, prg[(re[i] % K) // 12] - 1] for i in range(re.size)])
        assert np.abs(sym - want).max() <= 1e-13 * np.abs(want).max()
        assert np.array_equal(ind, re[:, None] + K * L * np.arange(P)[None, :] + 1)
