"""SINR -> CQI oracle known answers (precodedSINR.m:11-17, cqiSelect.m:697-722, setupSINRtoCQIMappingTable.m:7-11)."""
import numpy as np
import pytest

import oracle.cqi as OQ
from conftest import load_pkg


def test_precoded_sinr_closed_forms():
    # single layer, matched filter bound: sinr = |H w|^2 / sigma^2
    h = np.array([[1 + 1j, 2.0], [0.5, -1j]])
    w = np.array([[1.0], [1j]]) / np.sqrt(2)
    g = h @ w
    assert OQ.precoded_sinr(h, 0.5, w) == pytest.approx(float(np.real(g.conj().T @ g)[0, 0]) / 0.25, rel=1e-13)
    # orthogonal layers on an identity channel: every layer sees 1/sigma^2
    assert OQ.precoded_sinr(np.eye(4), 0.1, np.eye(4)[:, :3]) == pytest.approx(3 * 100.0, rel=1e-12)


def test_get_cqi_table_edges():
    t = OQ.DOWNLINK_SINR90PC
    assert t.size == 15 and OQ.UPLINK_SINR90PC.size == 15 and np.allclose(t - OQ.UPLINK_SINR90PC, 2.0)
    assert OQ.get_cqi(10 ** (-3.46 / 10), t) == 1                 # <= is inclusive
    assert OQ.get_cqi(10 ** (-3.47 / 10), t) == 0                 # below the table -> CQI 0
    assert OQ.get_cqi(10 ** (35.43 / 10) * 1.0001, t) == 15
    assert OQ.get_cqi(10 ** (27.0 / 10), t) == 11
    assert np.isnan(OQ.get_cqi(np.nan, t))
    pkg = load_pkg()
    for s in (1e-3, 0.5, 3.0, 40.0, 900.0, 5000.0):
        assert pkg.communication.phyLayer.getCQI(s, t) == OQ.get_cqi(s, t)
    assert np.array_equal(pkg.communication.phyLayer.DOWNLINK_SINR90PC, OQ.DOWNLINK_SINR90PC)
