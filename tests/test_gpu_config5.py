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
        assert len(ues["cqi"]) == 2 and all(-1 <= q <= 15 for q in ues["cqi"])
        assert 0 <= ues["ue_los"] <= 2 and ues["n_walls"] > 0
        if cell["nRng"] is not None:                       # a detected LoS target: estimates are bin multiples
            assert cell["nRng"] >= 1 and 0.0 < cell["rngEst0"] < 600.0 and abs(cell["velEst0"]) <= 60.0
    assert any(c["nRng"] is not None for c in out["cells"])
