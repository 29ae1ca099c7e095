"""Road-network scenario on the CPM-lab map (BASELINE configs 1..4).

Restates the parts of the reference's scenario generation that produce optimizer inputs:
    lanelets from the map            RoadDataCommonRoad.get_lanelets (RoadDataCommonRoad.m:48-64)
    lanelet boundaries               RoadDataCommonRoad.get_lanelet_boundary (:259-290): a lanelet with a
                                     same-direction left (else right) neighbour takes that neighbour's outer bound.
                                     The merging/forking extension (:291-700) needs polyshape and is NOT restated —
                                     documented deviation, it only widens a few boundaries near merges.
    reference loops, path ids        scenarios/road_network/get_reference_lanelets_loop.m:1-156
    reference path                   scenarios/road_network/generate_reference_path_loop.m:1-46
    vehicles                         scenarios/road_network/Commonroad.m:5-69 (random draws use numpy, not MATLAB's RandStream)
    predicted lanelets / boundary    hlc/controller/common/get_predicted_lanelets.m:1-63, get_lanelets_boundary.m:1-75
The map itself is a data fixture (data/labmap.npz, extracted by tests/golden/make_labmap_fixture.py).
Tiled copies of the map give the 128- and 512-vehicle configurations.
"""
import math
import os
from typing import List

import numpy as np

from .config import Config
from .mpa import get_mpa
from .scenario import Scenario, Vehicle

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "labmap.npz")

# get_reference_lanelets_loop.m:25-38
_LOOPS = [
    [4, 6, 8, 60, 58, 56, 54, 80, 82, 84, 86, 34, 32, 30, 28, 2],
    [1, 3, 23, 10, 12, 17, 43, 38, 36, 49, 29, 27],
    [64, 62, 75, 55, 53, 79, 81, 101, 88, 90, 95, 69],
    [40, 45, 97, 92, 94, 100, 83, 85, 33, 31, 48, 42],
    [5, 7, 59, 57, 74, 68, 66, 71, 19, 14, 16, 22],
    [41, 39, 20, 63, 61, 57, 55, 67, 65, 98, 37, 35, 31, 29],
    [3, 5, 9, 11, 72, 91, 93, 81, 83, 87, 89, 46, 13, 15],
    [1, 3, 23, 10, 12, 18, 14, 16, 22, 5, 7, 59, 57, 74, 68, 66, 70, 64, 62, 75, 55, 53, 79, 81, 101, 88, 90, 96, 92, 94, 100, 83, 85, 33, 31, 48, 42, 40, 44, 38, 36, 49, 29, 27],
    [1, 3, 5, 9, 11, 26, 52, 37, 35, 31, 29, 27],
    [3, 5, 7, 59, 57, 55, 67, 65, 76, 24, 13, 15],
    [79, 81, 83, 87, 89, 104, 78, 63, 61, 57, 55, 53],
    [33, 31, 29, 41, 39, 50, 102, 91, 93, 81, 83, 85],
]
# get_reference_lanelets_loop.m:40-146: path id -> (loop, starting lanelet)
_PATHS = {
    1: (1, 4), 2: (1, 8), 3: (1, 58), 4: (1, 54), 5: (1, 82), 6: (1, 86), 7: (1, 32), 8: (1, 28),
    9: (2, 1), 10: (2, 10), 11: (2, 17), 12: (2, 38), 13: (2, 49),
    14: (3, 64), 15: (3, 75), 16: (3, 79), 17: (3, 88), 18: (3, 95),
    19: (4, 42), 20: (4, 45), 21: (4, 92), 22: (4, 100), 23: (4, 33),
    24: (5, 22), 25: (5, 59), 26: (5, 68), 27: (5, 19), 28: (5, 14),
    29: (6, 39), 30: (6, 61), 31: (6, 55), 32: (6, 65), 33: (6, 35), 34: (6, 29),
    35: (7, 15), 36: (7, 5), 37: (7, 11), 38: (7, 93), 39: (7, 83), 40: (7, 89), 41: (5, 71),
    51: (8, 18), 52: (8, 70), 53: (8, 96), 54: (8, 44),
    61: (9, 26), 62: (10, 76), 63: (11, 104), 64: (12, 50),
}


def get_reference_lanelets_loop(path_id: int) -> List[int]:
    loop, start = _PATHS[path_id]
    seq = _LOOPS[loop - 1]
    i = seq.index(start)
    return seq[i:] + seq[:i]  # :150-155


class LabMap:
    def __init__(self):
        d = np.load(_DATA)
        self.n = int(d["n_points"].shape[0])
        self.pred = d["pred"]
        self.succ = d["succ"]
        self.adj = d["adj"]
        self.lanelets = []  # (P, 6) rows [rx ry lx ly cx cy]  (LaneletInfo.m:5-10), index = id - 1
        for i in range(self.n):
            P = int(d["n_points"][i])
            left = d["bounds"][i, 0, :P]
            right = d["bounds"][i, 1, :P]
            centre = 0.5 * (left + right)  # RoadDataCommonRoad.m:57-58
            self.lanelets.append(np.column_stack([right, left, centre]))
        # lanelet_boundary{i} = {left (P,2), right (P,2)}  RoadDataCommonRoad.m:259-290 (adjacent-lane rule only)
        self.boundary = []
        for i in range(self.n):
            lan = self.lanelets[i]
            left, right = lan[:, 2:4], lan[:, 0:2]
            la, la_same = self.adj[i, 0]
            ra, ra_same = self.adj[i, 1]
            if la and la_same:
                left = self.lanelets[la - 1][:, 2:4]
            elif ra and ra_same:
                right = self.lanelets[ra - 1][:, 0:2]
            self.boundary.append((left.copy(), right.copy()))

    def is_longitudinal(self, a: int, b: int) -> bool:
        """True if lanelet b follows a or a follows b (LaneletRelationshipType.longitudinal)."""
        return b in self.succ[a - 1] or a in self.succ[b - 1]


_MAP = None


def lab_map() -> LabMap:
    global _MAP
    if _MAP is None:
        _MAP = LabMap()
    return _MAP


def generate_reference_path_loop(lanelets_index, lanelets):
    """generate_reference_path_loop.m:1-46 -> (path (n,2), points_index (1-based last point of each lanelet))."""
    target = [lanelets[i - 1] for i in lanelets_index]
    path = np.vstack([t[:, 4:6] for t in target])
    s = np.diff(path, axis=0).sum(axis=1)
    tol = 1e-4 * max(np.max(np.abs(s)), 0.0)  # ismembertol(sum(diff(path,1),2), 0, 1e-4): tolerance scaled by max |data|
    redundant = np.concatenate(([False], np.abs(s) <= tol))
    reduced = path[~redundant]
    lengths = np.array([t.shape[0] for t in target])
    cum_len = np.cumsum(lengths)
    cum_red = np.cumsum(redundant.astype(np.int64))
    points_index = cum_len - cum_red[cum_len - 1]
    return reduced, points_index


def calculate_yaw_first(path):
    return math.atan2(path[1, 1] - path[0, 1], path[1, 0] - path[0, 0])  # utility/calculate_yaw.m:18-21


def get_predicted_lanelets(n_points_total, points_index_of_lanelets, lanelets_index, ref_points_index, current_point_index):
    """get_predicted_lanelets.m:25-62 (all indices 1-based)."""
    rpi = list(ref_points_index)
    index_add = rpi[-1] + 4
    if index_add > n_points_total:
        index_add -= n_points_total
    rpi.append(index_add)
    idx = [int(np.sum(p > points_index_of_lanelets)) + 1 for p in rpi]
    seen = []
    for q in idx:  # unique(..., 'stable')
        if q not in seen:
            seen.append(q)
    if len(seen) == 1:
        nxt = seen[0] + 1
        if nxt > len(lanelets_index):
            nxt = 1
        seen.append(nxt)
    current_idx = int(np.sum(current_point_index > points_index_of_lanelets)) + 1
    current_idx = min(current_idx, len(lanelets_index))
    seen = [min(q, len(lanelets_index)) for q in seen]
    return [lanelets_index[q - 1] for q in seen], lanelets_index[current_idx - 1]


def get_lanelets_boundary(predicted_lanelets, boundaries, lanelets_index, is_loop):
    """get_lanelets_boundary.m:18-68 -> (left (2,P), right (2,P))."""
    pb = [boundaries[i - 1] for i in predicted_lanelets]
    left = np.vstack([b[0][:-1] for b in pb] + [pb[-1][0][-1:]])  # :26-28
    right = np.vstack([b[1][:-1] for b in pb] + [pb[-1][1][-1:]])  # :30-32
    pos = lanelets_index.index(predicted_lanelets[0])  # :39
    if pos != 0:
        pred = lanelets_index[pos - 1]
    elif is_loop:
        pred = lanelets_index[-1]
    else:
        pred = None
    if pred is not None:  # :52-65
        pl, pr = boundaries[pred - 1]
        num_added = min(4, min(pr.shape[0] - 1, pl.shape[0] - 1))
        left = np.vstack([pl[-1 - num_added : -1], left])
        right = np.vstack([pr[-1 - num_added : -1], right])
    return left.T.copy(), right.T.copy()


def randomize_path_ids(amount: int, seed: int, enforce_crossing_intersection=True):
    """Config.randomize_path_ids (Config.m:127-152) with numpy's generator instead of MATLAB's mt19937ar stream."""
    possible = list(range(9, 42)) if enforce_crossing_intersection else list(range(1, 42))
    rng = np.random.default_rng(seed)
    return sorted(int(v) for v in rng.choice(possible, size=amount, replace=False))


def commonroad_scenario(options: Config, seed: int = 1, tiles: int = 1) -> Scenario:
    """Commonroad.m:5-48.  `tiles` > 1 lays out translated copies of the 4.5 m x 4 m map on a grid (no coupling between
    tiles) and spreads options.amount vehicles over them — the synthetic 128/512-vehicle configurations."""
    m = lab_map()
    mpa = get_mpa(options)
    speeds = mpa.get_straight_speeds_of_mpa()
    rng = np.random.default_rng(seed + 1000)
    per_tile = int(math.ceil(options.amount / tiles))
    grid = int(math.ceil(math.sqrt(tiles)))
    vehicles = []
    tile_of = []
    for t in range(tiles):
        n_here = min(per_tile, options.amount - len(vehicles))
        if n_here <= 0:
            break
        if options.path_ids and tiles == 1:
            ids = list(options.path_ids)
        else:
            ids = randomize_path_ids(n_here, seed * 131 + t)
        ox, oy = (t % grid) * 12.0, (t // grid) * 12.0  # far beyond any coupling distance: tiles never couple
        for pid in ids:
            li = get_reference_lanelets_loop(pid)
            path, points_index = generate_reference_path_loop(li, m.lanelets)
            path = path + np.array([ox, oy])
            vehicles.append(
                Vehicle(
                    x_start=float(path[0, 0]),
                    y_start=float(path[0, 1]),
                    yaw_start=calculate_yaw_first(path),
                    reference_path=path,
                    reference_speed=float(rng.choice(speeds)),  # Commonroad.m:44-45
                    lanelets_index=li,
                    points_index=points_index,
                    is_loop=m.is_longitudinal(li[0], li[-1]),  # Commonroad.m:27-34
                )
            )
            tile_of.append((ox, oy))
    sc = Scenario(vehicles=vehicles)
    sc.lanelet_boundary = m.boundary
    sc.tile_offset = tile_of
    return sc


def boundary_provider(scenario):
    """Returns f(i, vehicle, ref_points_index, current_point_index) -> (left (2,P), right (2,P)) for the controller
    (HighLevelController.m:220-247)."""
    m = lab_map()

    def f(i, veh, ref_points_index, current_point_index):
        predicted, _ = get_predicted_lanelets(veh.reference_path.shape[0], veh.points_index, veh.lanelets_index, ref_points_index, current_point_index)
        left, right = get_lanelets_boundary(predicted, m.boundary, veh.lanelets_index, veh.is_loop)
        off = np.array(scenario.tile_offset[i]).reshape(2, 1)
        return left + off, right + off

    return f
