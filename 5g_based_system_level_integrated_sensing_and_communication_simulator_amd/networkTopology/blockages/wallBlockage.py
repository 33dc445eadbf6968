"""networkTopology.blockages.wallBlockage (+networkTopology/+blockages/wallBlockage.m).

The plane of the wall (normVec, normDist) is host-side scalar prep exactly as in the reference constructor
(wallBlockage.m:57-65); the per-link geometry (projection + winding number) runs on the GPU
(csrc/los.hip via isac_los_check_dev / isac_winding_number_dev).  No CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ... import _lib as L


def _plane(corners: np.ndarray):
    # vectors = repmat(c(:,1),1,n-1) - c(:,2:end); basis = orth(vectors); normal = cross(basis(:,1), basis(:,2))
    vectors = corners[:, :1] - corners[:, 1:]
    u, s, _ = np.linalg.svd(vectors, full_matrices=False)
    tol = max(vectors.shape) * np.spacing(s.max()) if s.size else 0.0
    basis = u[:, s > tol]
    if basis.shape[1] < 2:
        raise ValueError("wall corners are collinear: the wall plane is undefined")
    n = np.cross(basis[:, 0], basis[:, 1])
    n = (1.0 / np.linalg.norm(n)) * n
    return n, float(n @ corners[:, 0])


def pack_walls(walls):
    """Flat wall table of include/isac.h: corners [3 x C] (Fortran order), offsets [W+1], normals [3 x W], normDist [W]."""
    walls = list(walls)
    off = np.zeros(len(walls) + 1, dtype=np.int32)
    for i, w in enumerate(walls):
        off[i + 1] = off[i] + w.cornerList.shape[1]
    corners = np.asfortranarray(np.concatenate([w.cornerList for w in walls], axis=1)) if walls else np.zeros((3, 0), order="F")
    normals = np.asfortranarray(np.stack([w.normVec for w in walls], axis=1)) if walls else np.zeros((3, 0), order="F")
    dist = np.array([w.normDist for w in walls], dtype=np.float64)
    return corners, off, normals, dist


class WallTable:
    """Wall table resident in HBM (upload once per layout, reuse for every batch of links)."""

    def __init__(self, walls, ctx=None):
        self.ctx = ctx or L.default_context()
        corners, off, normals, dist = pack_walls(walls)
        self.n_walls = int(off.size - 1)
        self.corners = self.ctx.to_device(corners) if corners.size else None
        self.offsets = self.ctx.to_device(off)
        self.normals = self.ctx.to_device(normals) if normals.size else None
        self.dist = self.ctx.to_device(dist) if dist.size else None

    def _p(self, d):
        return C.c_void_p(d.ptr if d is not None else 0)

    def check_los(self, ue, ant, return_counts=False):
        """ue, ant [3 x n] paired -> bool [n] (True = line of sight) and optionally the number of blocking walls."""
        ue = np.asfortranarray(np.asarray(ue, dtype=np.float64).reshape(3, -1))
        ant = np.asfortranarray(np.asarray(ant, dtype=np.float64).reshape(3, -1))
        if ant.shape[1] == 1 and ue.shape[1] != 1:
            ant = np.asfortranarray(np.repeat(ant, ue.shape[1], axis=1))
        if ue.shape != ant.shape:
            raise ValueError("ue and ant must both be [3 x n]")
        n = ue.shape[1]
        if n == 0:
            return (np.zeros(0, bool), np.zeros(0, np.int32)) if return_counts else np.zeros(0, bool)
        ctx = self.ctx
        d_ue, d_ant = ctx.to_device(ue), ctx.to_device(ant)
        d_los = ctx.empty((n,), np.uint8)
        d_cnt = ctx.empty((n,), np.int32)
        ctx.check(ctx.lib.isac_los_check_dev(ctx.handle, C.c_void_p(d_ue.ptr), C.c_void_p(d_ant.ptr), C.c_int64(n),
                                             self._p(self.corners), self._p(self.offsets), self._p(self.normals),
                                             self._p(self.dist), C.c_int32(self.n_walls), C.c_void_p(d_los.ptr),
                                             C.c_void_p(d_cnt.ptr)))
        los = d_los.numpy().astype(bool)
        return (los, d_cnt.numpy()) if return_counts else los

    def winding(self, points):
        """points [3 x n] -> winding numbers [n x W]."""
        pts = np.asfortranarray(np.asarray(points, dtype=np.float64).reshape(3, -1))
        n = pts.shape[1]
        if n == 0 or self.n_walls == 0:
            return np.zeros((n, self.n_walls))
        ctx = self.ctx
        d_pts = ctx.to_device(pts)
        out = ctx.empty((n, self.n_walls), np.float64)
        ctx.check(ctx.lib.isac_winding_number_dev(ctx.handle, C.c_void_p(d_pts.ptr), C.c_int64(n), self._p(self.corners),
                                                  self._p(self.offsets), self._p(self.normals), C.c_int32(self.n_walls),
                                                  C.c_void_p(out.ptr)))
        return out.numpy()


class wallBlockage:
    """wall = wallBlockage(cornerList, loss): cornerList [3 x nCorners] (x;y;z), loss in dB."""

    def __init__(self, cornerList=None, loss=10.0):
        if cornerList is None:                                        # wallBlockage.m:36-39 defaults
            cornerList = [[1, 0, 0, 1, 1], [0, 0, 1, 1, 0], [0, 0, 1, 1, 0]]
        c = np.asarray(cornerList, dtype=np.float64)
        if c.ndim != 2 or c.shape[1] < 3:
            raise ValueError("use at least three points to specify a wall")   # wallBlockage.m:41-43
        if c.shape[0] < 3:
            raise ValueError("use 3D points for corners")                      # wallBlockage.m:45-47
        self.cornerList = c
        self.loss = float(loss)
        self.normVec, self.normDist = _plane(c)
        x_size = c[0].max() - c[0].min()
        y_size = c[1].max() - c[1].min()
        self.xCenter, self.yCenter = c[0].min() + x_size / 2, c[1].min() + y_size / 2     # blockage superclass
        self.radius = 0.5 * np.sqrt(x_size ** 2 + y_size ** 2)
        self._table = None

    def _tab(self, ctx=None):
        if self._table is None or (ctx is not None and self._table.ctx is not ctx):
            self._table = WallTable([self], ctx)
        return self._table

    def checkBlockage(self, ue, ant, *, ctx=None):
        """blockageDecision [n] bool, True = collision (wallBlockage.m:88-121)."""
        return ~self._tab(ctx).check_los(ue, ant)

    def checkIsInside(self, ue, *, ctx=None):
        """isInsideDecision [n] bool (wallBlockage.m:70-86)."""
        return self._tab(ctx).winding(ue)[:, 0] > 0.1
