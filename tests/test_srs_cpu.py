"""Known answers for the oracle restatement of the SRS measurement (oracle/srs.py: pmiSelect.m / sinrPerSubband.m / precodedSINR.m / gNBPhy.m:1033-1058) and of
riSelect.m (oracle/pmi.py::ri_select) -- values that follow from the definitions by hand, so that the GPU parity tests compare against a checked checker."""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import pytest

import oracle.cqi as OQ
import oracle.pmi as OP
import oracle.srs as OS


def test_max_tpmi_table():
    assert [OS.max_pusch_tpmi(*a) for a in ((1, 1), (1, 2), (1, 4), (2, 2), (2, 4), (3, 4), (4, 4))] == [0, 5, 27, 2, 21, 6, 4]
    with pytest.raises(ValueError):
        OS.max_pusch_tpmi(2, 1)
    with pytest.raises(ValueError):
        OS.max_pusch_tpmi(1, 3)


def test_pusch_codebook_tables():
    for t in range(6):                                                    # Table 6.3.1.5-1: total power 1/2 (antenna selection) or 1
        w = OS.pusch_codebook(1, 2, t)
        assert w.shape == (2, 1) and np.isclose(np.vdot(w, w).real, 0.5 if t < 2 else 1.0)
    assert np.allclose(OS.pusch_codebook(1, 2, 4)[:, 0] * np.sqrt(2), [1, 1j])
    for t in range(3):                                                    # Table 6.3.1.5-4: orthogonal columns of norm^2 1/2
        w = OS.pusch_codebook(2, 2, t)
        assert np.allclose(w.conj().T @ w, 0.5 * np.eye(2))


def test_precoded_sinr_identity_channel():
    # H = I, W = I / sqrt 2: W'H'HW = I / 2 -> per layer 1 / den - 1 = 0.5 / sigma^2, two layers -> 1 / sigma^2
    assert np.isclose(OS.precoded_sinr(np.eye(2), 0.1, OS.pusch_codebook(2, 2, 0)), 1 / 0.01)
    # one layer, h = [1; 1] on one receive antenna, w = [1; 1] / sqrt 2: |h w|^2 / sigma^2 = 2 / sigma^2; w = [1; -1] / sqrt 2 -> 0
    h = np.array([[1.0, 1.0]])
    assert np.isclose(OS.precoded_sinr(h, 0.5, OS.pusch_codebook(1, 2, 2)), 2 / 0.25)
    assert np.isclose(OS.precoded_sinr(h, 0.5, OS.pusch_codebook(1, 2, 3)), 0.0)


def test_sinr_per_subband_by_hand():
    # 5 RBs, bands of 2 RBs -> subbands of 24, 24, 12 subcarriers (the fractional last band takes the rest)
    s = np.zeros((60, 2, 3))
    s[0:24:2, 0, :] = [1.0, 2.0, 3.0]                                     # comb 2 on one symbol: 12 REs in band 1
    s[48:60, 1, :] = [4.0, 0.0, 1.0]                                      # band 3: 12 REs; band 2: none
    out, idx = OS.sinr_per_subband(s, 2)
    assert idx.tolist() == [[1, 24], [25, 48], [49, 60]]
    assert np.allclose(out[0], [1, 2, 3]) and np.all(np.isnan(out[1])) and np.allclose(out[2], [4, 0, 1])


def test_pmi_select_prefers_the_matched_precoder_and_reports_nan_bands():
    k = 60
    h = np.zeros((k, 1, 1, 2), dtype=np.complex128)
    h[0:24, 0, 0, :] = [1, 1j]                                            # matched filter [1; -j]: TPMI 5
    h[48:60, 0, 0, :] = [2, -1]                                           # TPMI 3 ([1, -1] itself sums to zero over the ports: "no estimate" by pmiSelect.m:36)
    pmi, sinr, _ = OS.pmi_select(1, h, 0.1, 2)
    assert pmi[0] == 5 and np.isnan(pmi[1]) and pmi[2] == 3
    assert np.isclose(sinr[0, 5], 2 / 0.1) and np.isclose(sinr[0, 4], 0.0)
    assert OS.pmi_select(1, h, 0.0, 2)[0] is np.nan or np.isnan(OS.pmi_select(1, h, 0.0, 2)[0])
    hq = h.copy(); hq[48:60, 0, 0, :] = [1, -1]
    assert np.isnan(OS.pmi_select(1, hq, 0.1, 2)[0][2])                   # the reference's availability test is sum(hest, 3:4) ~= 0
    # the gNB's report: band 2 takes floor(mean(5, 3)) = 4 and the mean SINR row; bands 1..n-1 go through the table, the last band copies its neighbour
    pmi_f, sel, cqi = OS.srs_report(1, h, 0.1, 2, 5, OQ.UPLINK_SINR90PC)
    assert pmi_f.tolist() == [5, 4, 3]
    assert np.isclose(sel[0], 20.0) and np.isclose(sel[1], 0.5 * (sinr[0, 4] + sinr[2, 4]))
    want0 = np.count_nonzero(OQ.UPLINK_SINR90PC <= 10 * np.log10(20.0)) - 1
    want1 = max(np.count_nonzero(OQ.UPLINK_SINR90PC <= 10 * np.log10(sel[1])) - 1, 1)
    assert cqi.tolist() == [want0, want0, want1, want1, want1]


def _report(nrb=24):
    return SimpleNamespace(NSizeBWP=nrb, NStartBWP=0, PanelDimensions=(2, 1), CodebookMode=1, PMIMode="Subband", SubbandSize=4)


def test_ri_select_rank_follows_the_channel():
    rng = np.random.default_rng(3)
    nrb, p = 24, 4
    k = np.arange(1, 12 * nrb + 1, 6)
    l = np.ones_like(k)
    u = rng.standard_normal((2, 1)) + 1j * rng.standard_normal((2, 1))
    v = rng.standard_normal((1, p)) + 1j * rng.standard_normal((1, p))
    h1 = np.broadcast_to((u @ v)[None, None], (12 * nrb, 14, 2, p)).copy()    # rank-one channel: the second layer carries nothing
    ri, pmi, total = OP.ri_select(_report(nrb), h1, k, l, 0.01)
    assert ri == 1 and total[0] > total[1] - 0.1 and not np.any(np.isnan(pmi.i1))
    q, _ = np.linalg.qr(rng.standard_normal((p, p)) + 1j * rng.standard_normal((p, p)))
    h2 = np.broadcast_to((3.0 * q[:2, :])[None, None], (12 * nrb, 14, 2, p)).copy()   # two orthogonal strong rows: two layers at high SNR
    ri2, _, total2 = OP.ri_select(_report(nrb), h2, k, l, 0.001)
    assert ri2 == 2 and total2[1] > total2[0] + 0.1
    # restriction to rank 1 only; no CSI-RS -> NaN
    assert OP.ri_select(_report(nrb), h2, k, l, 0.001, ri_restriction=[1, 0, 0, 0, 0, 0, 0, 0])[0] == 1
    assert np.isnan(OP.ri_select(_report(nrb), h2, k[:0], l[:0], 0.001)[0])
