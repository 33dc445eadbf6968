"""LoS blockage parity: HIP (isac_los_check_dev / isac_winding_number_dev through the host mirror
networkTopology.blockages) against oracle/los.py on the same seeded layouts.  Decisions: exact (bool/integer);
winding numbers: <= 1e-9 absolute (sum of a few fp64 atan2)."""
from __future__ import annotations

import numpy as np
import pytest

from oracle import los as OL
from conftest import load_pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


def _random_city(rng, n_buildings, span=400.0):
    plans, heights = [], []
    for _ in range(n_buildings):
        cx, cy = rng.uniform(-span, span, 2)
        n = int(rng.integers(3, 9))                                   # 3..8 corners, star-shaped (possibly concave)
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        rad = rng.uniform(8.0, 30.0, n)
        fp = np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)])
        plans.append(np.concatenate([fp, fp[:, :1]], axis=1))         # closed path like the OSM ways
        heights.append(float(rng.uniform(5.0, 40.0)))
    return plans, heights


def test_wall_winding_and_blockage(pkg):
    B = pkg.networkTopology.blockages
    sq = np.array([[0.0, 4.0, 4.0, 0.0], [0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 3.0, 3.0]])
    w = B.wallBlockage(sq, 10)
    n, d = OL.wall_plane(sq)
    assert abs(abs(w.normVec @ n) - 1) < 1e-14 and abs(abs(w.normDist) - abs(d)) < 1e-14
    rng = np.random.default_rng(5)
    pts = np.stack([rng.uniform(-2, 6, 500), np.zeros(500), rng.uniform(-2, 5, 500)])
    pts[:, 0] = [0.0, 0.0, 0.0]                                       # a corner hit -> winding forced to 1
    got = w._tab().winding(pts)[:, 0]
    want = OL.winding_number(sq, n, pts)
    assert np.abs(got - want).max() < 1e-9
    assert got[0] == 1.0
    assert np.array_equal(w.checkIsInside(pts), want > 0.1)
    ant = np.array([[2.0], [-5.0], [1.0]])
    ue = np.stack([rng.uniform(-4, 8, 300), rng.uniform(1, 9, 300), rng.uniform(0, 4, 300)])
    ue[:, 0] = [3.0, -5.0, 2.0]                                       # parallel to the wall plane: NaN -> not blocked
    ue[:, 1] = [2.0, -2.0, 1.0]                                       # infinite-line quirk: blocked
    got_b = w.checkBlockage(ue, ant)
    want_b = OL.wall_check_blockage(sq, n, d, ue, np.repeat(ant, 300, axis=1))
    assert np.array_equal(got_b, want_b)
    assert not got_b[0] and got_b[1]


def test_building_blockage_and_inside(pkg):
    B = pkg.networkTopology.blockages
    fp = np.array([[10.0, 20.0, 20.0, 10.0, 10.0], [-5.0, -5.0, 5.0, 5.0, -5.0]])
    b = B.building(fp, 25.0, 3.0)
    assert b.nWall == 5
    gnb = np.array([[0.0], [0.0], [30.0]])
    ue = np.array([[40.0, 0.0, 1.5], [40.0, 60.0, 1.5], [5.0, 0.0, 26.0]]).T
    assert b.checkBlockage(ue, gnb).tolist() == [True, False, True]
    inside = b.checkIsInside(np.array([[15.0, 0.0, 3.0], [15.0, 0.0, 30.0], [25.0, 0.0, 3.0]]).T)
    assert inside.tolist() == [True, False, False]


@pytest.mark.parametrize("n_buildings,n_links,seed", [(1, 7, 1), (40, 513, 2), (300, 4096, 3)])
def test_city_check_los_matches_oracle(pkg, n_buildings, n_links, seed):
    B = pkg.networkTopology.blockages
    rng = np.random.default_rng(seed)
    plans, heights = _random_city(rng, n_buildings)
    town = B.city.from_floor_plans(plans, heights)
    gnb = np.array([0.0, 0.0, 30.0])
    ue = np.stack([rng.uniform(-450, 450, n_links), rng.uniform(-450, 450, n_links), rng.uniform(1.0, 2.0, n_links)], axis=1)
    got, cnt = town._tab().check_los(ue.T, gnb[:, None], return_counts=True)
    want = OL.check_los(list(zip(plans, heights)), ue, gnb)
    assert np.array_equal(got, want)
    assert np.array_equal(town.checkLoS(ue, gnb), want)
    assert np.array_equal(cnt > 0, ~want)
    assert 0 < want.sum() < n_links or n_buildings == 1               # the drop has both LoS and NLoS links
    # scalar call form of the reference (networkSimulation.m:138)
    assert town.checkLoS(ue[0], gnb) == bool(want[0])
    # per-link antennas (paired form)
    ants = np.stack([rng.uniform(-450, 450, n_links), rng.uniform(-450, 450, n_links), np.full(n_links, 30.0)], axis=1)
    assert np.array_equal(town.checkLoS(ue, ants), OL.check_los(list(zip(plans, heights)), ue, ants))


def test_empty_inputs(pkg):
    B = pkg.networkTopology.blockages
    town = B.city([])
    assert town.checkLoS(np.zeros((4, 3)), np.array([0.0, 0.0, 30.0])).all()          # no walls: everything is LoS
    town2 = B.city.from_floor_plans([np.array([[0.0, 1.0, 1.0, 0.0], [0.0, 0.0, 1.0, 0.0]])], [3.0])
    assert town2.checkLoS(np.zeros((0, 3)), np.array([0.0, 0.0, 30.0])).shape == (0,)
