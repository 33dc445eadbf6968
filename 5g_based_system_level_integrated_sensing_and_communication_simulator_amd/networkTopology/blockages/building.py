"""networkTopology.blockages.building (+networkTopology/+blockages/building.m)."""
from __future__ import annotations

import numpy as np

from .wallBlockage import wallBlockage, WallTable


class building:
    """b = building(floorPlan [2 x n], height, loss, name): one quad wall per floor-plan edge + the ceiling (building.m:36-98)."""

    def __init__(self, floorPlan=None, height=5.0, loss=3.0, name=""):
        if floorPlan is None:                                         # building.m:53-57 defaults
            floorPlan = [[1, 0, 0, 1, 1], [0, 0, 1, 1, 0]]
        fp = np.asarray(floorPlan, dtype=np.float64)
        self.floorPlan, self.height, self.name = fp, float(height), name
        self.xSize = fp[0].max() - fp[0].min()
        self.ySize = fp[1].max() - fp[1].min()
        self.wallList = []
        for i in range(fp.shape[1] - 1):
            ll = [fp[0, i], fp[1, i], 0.0]; lr = [fp[0, i + 1], fp[1, i + 1], 0.0]
            ul = [fp[0, i], fp[1, i], self.height]; ur = [fp[0, i + 1], fp[1, i + 1], self.height]
            self.wallList.append(wallBlockage(np.array([ll, lr, ur, ul]).T, loss))
        self.wallList.append(wallBlockage(np.vstack([fp, self.height * np.ones((1, fp.shape[1]))]), loss))
        self._table = None

    @property
    def nWall(self):
        return len(self.wallList)

    def _tab(self, ctx=None):
        if self._table is None or (ctx is not None and self._table.ctx is not ctx):
            self._table = WallTable(self.wallList, ctx)
        return self._table

    def checkBlockage(self, userPositionList, antennaPositionList, *, ctx=None):
        """[n] bool, True = some wall of the building blocks the link (building.m:113-137)."""
        return ~self._tab(ctx).check_los(userPositionList, antennaPositionList)

    def checkIsInside(self, userPositionList, *, ctx=None):
        """isIndoorDecision (building.m:139-174): below the roof and inside the ceiling polygon."""
        u = np.asarray(userPositionList, dtype=np.float64)
        if u.shape[0] == 2:
            below = np.ones(u.shape[1], dtype=bool)
            u = np.vstack([u, np.zeros((1, u.shape[1]))])
        elif u.shape[0] == 3:
            below = u[2] < self.height
            u = u.copy()
            u[2] = self.height
        else:
            raise ValueError("userPositionList is must be of dimension 3 x nUser.")
        return below & self.wallList[-1].checkIsInside(u, ctx=ctx)
