"""Smoke test of the config-5 assembly (examples/config5.py): LoS -> sensing CPI -> per-UE CDL apply -> SINR->CQI on a
small drop, one GPU.  The individual seams are checked against the oracle in test_gpu_parity.py / test_gpu_los.py; here
only the plumbing and the reference's NaN-on-blocked-target convention are exercised."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config5_example_runs(capsys, monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import config5
    monkeypatch.setattr(sys, "argv", ["config5.py", "--cells", "3", "--ues", "2", "--ants", "16", "--slots", "16"])
    config5.main()
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert len(out["cells"]) == 3 and len(out["rank0_ues"]) == 3
    for cell, ues in zip(out["cells"], out["rank0_ues"]):
        assert len(ues["cqi"]) == 2 and all(q is None or 0 <= q <= 15 for q in ues["cqi"]) and len(ues["pmi_i1"]) == 2
        assert 0 <= ues["ue_los"] <= 2 and ues["n_walls"] > 0
        if cell["nRng"] is not None:                       # a detected LoS target: estimates are bin multiples
            assert cell["nRng"] >= 1 and 0.0 < cell["rngEst0"] < 600.0 and abs(cell["velEst0"]) <= 60.0
    assert any(c["nRng"] is not None for c in out["cells"])


def test_config5_at_size_with_oracle_probe(capsys, monkeypatch):
    """BASELINE configs[4] at its stated size on one GPU -- 21 cells x 10 UEs, 64-antenna gNB, 16-slot CPIs -- with one UE's seams re-computed
    by the oracle: the CDL-D/A channel apply (<= 1e-10) and the Type-I PMI search + subband CQI report (integers exact)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import config5
    import oracle.cdl as OC
    import oracle.cqi as OQ
    import oracle.pmi as OP
    config5.PROBE = {"cell": 4, "ue": 3}
    monkeypatch.setattr(sys, "argv", ["config5.py", "--cells", "21", "--ues", "10", "--ants", "64", "--slots", "16"])
    try:
        config5.main()
    finally:
        cap = config5.PROBE.get("capture")
        config5.PROBE = None
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert len(out["cells"]) == 21 and [c["cell"] for c in out["cells"]] == list(range(21))
    assert sum(c["nRng"] is not None for c in out["cells"]) >= 10                   # most cells see their target (the rest: blocked -> NaN)
    assert all(len(u["cqi"]) == 10 for u in out["rank0_ues"])
    assert cap is not None
    # ---- CDL apply of the probed UE against the oracle's TR 38.901 restatement
    fs = 122.88e6
    cfg = OC.cdl_config(cap["profile"], 3.5e9, cap["tx_size"], (1, 1, 2, 1, 1), fs)
    want_rx = OC.apply_cdl(cfg, cap["wave"], cap["t0"])
    assert cap["rx"].shape == want_rx.shape == (61440, 2)
    assert np.abs(cap["rx"] - want_rx).max() <= 1e-10 * np.abs(want_rx).max()
    # ---- CSI report of the probed UE against the oracle's dlPMISelect / cqiSelect restatement
    k, l = cap["csi_k"], cap["csi_l"]
    h = np.zeros((3276, 14, 2, 4), dtype=np.complex128)
    h[k - 1, l - 1] = cap["hf"]
    cqi, pmi, ci, pi = OP.cqi_select(cap["report"], 1, h, k, l, cap["nvar"], OQ.DOWNLINK_SINR90PC)
    same = lambda a, b: np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.asarray(a)[~np.isnan(a)], np.asarray(b)[~np.isnan(b)])
    assert same(cap["pmi"].i1, pmi.i1) and same(cap["pmi"].i2, pmi.i2)
    assert same(cap["cqi"], cqi) and same(cap["subband_cqi"], ci.SubbandCQI) and cqi.size == 19        # wideband + 18 subbands of 16 PRBs
