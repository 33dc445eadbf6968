"""City-level LoS check: networkTopology.blockages.openStreetMapCity.checkLoS
(+networkTopology/+blockages/openStreetMapCity.m:67-93), batched.

The reference builds its building list from an OpenStreetMap HTTP query or a saved file
(openStreetMapCity.m:50-58); that I/O is out of scope here -- a city is constructed from floor plans and heights.
The reference checks one link per call from an interpreted loop (networkSimulation.m:134-160); `checkLoS` takes any
number of links and evaluates all links x all walls in one launch."""
from __future__ import annotations

import numpy as np

from .building import building
from .wallBlockage import WallTable


class city:
    def __init__(self, buildings=(), *, ctx=None):
        self.buildings = list(buildings)
        self._ctx = ctx
        self._table = None

    @classmethod
    def from_floor_plans(cls, floor_plans, heights, wallLoss=10.0, *, ctx=None):
        return cls([building(fp, h, wallLoss) for fp, h in zip(floor_plans, heights)], ctx=ctx)

    def _tab(self):
        if self._table is None:
            self._table = WallTable([w for b in self.buildings for w in b.wallList], self._ctx)
        return self._table

    def checkLoS(self, uePos, antPos):
        """losDecision = checkLoS(uePos, antPos): positions as ROW vectors [n x 3] (the caller's convention,
        networkSimulation.m:138,154); a single antenna row is shared by all UEs.  Returns bool [n], True = LoS;
        a scalar bool for a single link."""
        ue = np.atleast_2d(np.asarray(uePos, dtype=np.float64))
        ant = np.atleast_2d(np.asarray(antPos, dtype=np.float64))
        los = self._tab().check_los(ue.T, ant.T)
        return bool(los[0]) if np.ndim(uePos) == 1 else los
