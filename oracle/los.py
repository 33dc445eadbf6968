"""Line-of-sight blockage check: NumPy restatement of the reference's wall / building / city geometry.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the reference ships no vectors; the analytic
known-answer tests live in tests/test_los_cpu.py.

Follows
  +networkTopology/+blockages/wallBlockage.m:24-68    constructor (plane normal from orth() of the corner differences)
  +networkTopology/+blockages/wallBlockage.m:88-134   checkBlockage: project the UE onto the wall plane ALONG the UE-antenna
                                                      line, then winding number > 0.1
  +networkTopology/+blockages/wallBlockage.m:170-216  getWindingNumber (sum of signed angles between consecutive corner
                                                      directions; |sum|; corner hits -> 1)
  +networkTopology/+blockages/building.m:36-98        walls of a building: one quad per floor-plan edge + the ceiling polygon
  +networkTopology/+blockages/building.m:113-137      building blocks if any wall blocks
  +networkTopology/+blockages/openStreetMapCity.m:67-93   LoS = no building blocks

Reference quirk kept on purpose: checkBlockage intersects the *infinite* line through UE and antenna with the wall
(wallBlockage.m:116-121 never tests that the intersection lies between the two end points), so a wall behind the UE or
behind the antenna blocks as well.  A line parallel to the wall plane divides by zero -> NaN winding number -> "not blocked".
"""
from __future__ import annotations

import numpy as np


def wall_plane(corner_list: np.ndarray):
    """(normVec [3], normDist) of wallBlockage.m:57-65.  orth() = left singular vectors of the non-negligible singular
    values; the sign of the normal is arbitrary (LAPACK) and cancels in every use (|winding|, ratio of two projections)."""
    c = np.asarray(corner_list, dtype=np.float64)
    if c.shape[0] < 3:
        raise ValueError("use 3D points for corners")              # wallBlockage.m:45-47
    if c.shape[1] < 3:
        raise ValueError("use at least three points to specify a wall")   # wallBlockage.m:41-43
    vectors = c[:, :1] - c[:, 1:]                                   # repmat(c(:,1)) - c(:,2:end)
    u, s, _ = np.linalg.svd(vectors, full_matrices=False)
    tol = max(vectors.shape) * np.spacing(s.max()) if s.size else 0.0
    basis = u[:, s > tol]
    if basis.shape[1] < 2:
        raise ValueError("wall corners are collinear")              # cross(basis(:,1), basis(:,2)) would index out of range
    n = np.cross(basis[:, 0], basis[:, 1])
    n = (1.0 / np.linalg.norm(n)) * n
    return n, float(n @ c[:, 0])


def winding_number(corner_list: np.ndarray, norm_vec: np.ndarray, point: np.ndarray) -> np.ndarray:
    """getWindingNumber (wallBlockage.m:170-216).  point [3 x n] -> [n]."""
    poly = np.asarray(corner_list, dtype=np.float64)               # [3 x nc]
    pt = np.asarray(point, dtype=np.float64).reshape(3, -1)        # [3 x n]
    with np.errstate(all="ignore"):
        vec = poly[:, None, :] - pt[:, :, None]                    # [3 x n x nc]
        len_vec = np.sqrt(vec[0] ** 2 + vec[1] ** 2 + vec[2] ** 2)  # vecnorm(vec,2,1)
        invalid = len_vec < 1e-10
        vec = vec / len_vec
        shift = np.roll(vec, 1, axis=2)                            # circshift(vec,1,3)
        dotv = shift[0] * vec[0] + shift[1] * vec[1] + shift[2] * vec[2]
        cx = shift[1] * vec[2] - shift[2] * vec[1]
        cy = shift[2] * vec[0] - shift[0] * vec[2]
        cz = shift[0] * vec[1] - shift[1] * vec[0]
        ncross = norm_vec[0] * cx + norm_vec[1] * cy + norm_vec[2] * cz
        diff_angle = np.arctan2(ncross, dotv)
        acc = np.zeros(pt.shape[1])
        for j in range(poly.shape[1]):                             # sum(diffAngle,3): left to right
            acc = acc + diff_angle[:, j]
        w = np.abs(acc)
    w[invalid.sum(axis=1) > 0] = 1.0
    return w


def wall_check_blockage(corner_list, norm_vec, norm_dist, ue, ant) -> np.ndarray:
    """checkBlockage (wallBlockage.m:88-121).  ue, ant [3 x n] (paired) -> bool [n], True = blocked."""
    ue = np.asarray(ue, dtype=np.float64).reshape(3, -1)
    ant = np.asarray(ant, dtype=np.float64).reshape(3, -1)
    with np.errstate(all="ignore"):
        vec = ue - ant
        num = norm_dist - (norm_vec[0] * ue[0] + norm_vec[1] * ue[1] + norm_vec[2] * ue[2])
        den = norm_vec[0] * vec[0] + norm_vec[1] * vec[1] + norm_vec[2] * vec[2]
        proj = ue + vec * (num / den)
    return winding_number(corner_list, norm_vec, proj) > 0.1


def building_walls(floor_plan, height):
    """Corner lists of a building's walls (building.m:82-97): one quad per floor-plan edge, then the ceiling polygon."""
    fp = np.asarray(floor_plan, dtype=np.float64)
    walls = []
    for i in range(fp.shape[1] - 1):
        ll = np.array([fp[0, i], fp[1, i], 0.0]); lr = np.array([fp[0, i + 1], fp[1, i + 1], 0.0])
        ul = np.array([fp[0, i], fp[1, i], height]); ur = np.array([fp[0, i + 1], fp[1, i + 1], height])
        walls.append(np.stack([ll, lr, ur, ul], axis=1))
    walls.append(np.vstack([fp, height * np.ones((1, fp.shape[1]))]))
    return walls


def check_los(buildings, ue_pos, ant_pos) -> np.ndarray:
    """openStreetMapCity.checkLoS (openStreetMapCity.m:67-93) for paired positions.
    buildings: iterable of (floor_plan [2 x n], height); ue_pos, ant_pos [n x 3] (row vectors as the caller passes them,
    networkSimulation.m:138,154).  Returns bool [n], True = line of sight."""
    ue = np.atleast_2d(np.asarray(ue_pos, dtype=np.float64)).T
    ant = np.atleast_2d(np.asarray(ant_pos, dtype=np.float64)).T
    if ant.shape[1] == 1 and ue.shape[1] > 1:
        ant = np.repeat(ant, ue.shape[1], axis=1)
    blocked = np.zeros(ue.shape[1], dtype=np.int64)
    for fp, h in buildings:
        b = np.zeros(ue.shape[1], dtype=np.int64)
        for corners in building_walls(fp, h):
            n, d = wall_plane(corners)
            b = b + wall_check_blockage(corners, n, d, ue, ant)
        blocked = blocked + (b > 0)
    return ~(blocked > 0)
