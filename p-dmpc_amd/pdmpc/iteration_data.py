"""Value types on either side of the optimizer boundary.

VehicleIter        the 1-vehicle `IterationData` slice produced by IterationData.filter
                   (hlc/controller/common/IterationData.m:4-33,95-114) with the obstacle lists
                   PrioritizedController.plan appends (PrioritizedController.m:323-324).
ControlResultsInfo hlc/controller/common/ControlResultsInfo.m:5-17,36-41.
Tree               the path nodes in the layout OptimizerInterface.create_control_results_info_from_mex
                   builds (OptimizerInterface.m:63-101): node i+1 is the child of node i.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np


@dataclass
class VehicleIter:
    x0: np.ndarray  # (4,) x, y, yaw, speed                      IterationData.m:8
    trim_index: int  # 1-based                                      IterationData.m:9
    reference_trajectory_points: np.ndarray  # (Hp, 2)               IterationData.m:5
    v_ref: np.ndarray  # (Hp,)                                       IterationData.m:10
    predicted_lanelet_boundary: tuple = (None, None)  # (left 2xP, right 2xP) IterationData.m:13
    obstacles: List[np.ndarray] = field(default_factory=list)  # [(2,V)]       IterationData.m:19
    dynamic_obstacle_area: List[List[np.ndarray]] = field(default_factory=list)  # n_d x Hp  IterationData.m:20
    hdv_reachable_sets: List[List[np.ndarray]] = field(default_factory=list)  # adjacent HDVs x Hp  IterationData.m:29
    amount: int = 1  # IterationData.m:32


@dataclass
class Tree:
    """Nodes of the selected path only (ids 1..Hp+1, parent of i+1 is i); arrays are 1 x n as in Tree.m:3-13."""

    x: np.ndarray
    y: np.ndarray
    yaw: np.ndarray
    trim: np.ndarray
    k: np.ndarray
    g: np.ndarray
    h: np.ndarray
    parent: np.ndarray

    def size(self) -> int:  # Tree.m:102-104
        return int(self.parent.shape[0])


@dataclass
class ControlResultsInfo:
    tree: Optional[Tree]
    tree_path: np.ndarray  # 1-based ids in the SEARCH tree, (Hp+1,)        ControlResultsInfo.m:8
    n_expanded: int  # tree size                                             ControlResultsInfo.m:9
    shapes: List[np.ndarray]  # Hp arrays (2, V)                             ControlResultsInfo.m:10
    predicted_trims: np.ndarray  # (Hp,) 1-based                             ControlResultsInfo.m:12
    y_predicted: np.ndarray  # (3, Hp), NaN if exhausted                     ControlResultsInfo.m:14
    is_exhausted: bool = False  # ControlResultsInfo.m:15
    needs_fallback: bool = False  # ControlResultsInfo.m:16
    n_popped: int = 0  # backend counter (no reference field)
    status: int = 0  # PDMPC_OK / EXHAUSTED / ARENA_OVERFLOW


def info_from_record(rec, Hp: int) -> ControlResultsInfo:
    """Decode one pdmpc_vehicle_out record (numpy structured scalar, abi.VEHICLE_OUT_DTYPE)."""
    status = int(rec["status"])
    if status not in (0, 1):
        # PDMPC_ARENA_OVERFLOW (the reference's unbounded tree would have kept searching, Tree.m:54-70) and device-side error
        # statuses are not planning results: they must not be mistaken for an exhausted open list (GraphSearch.m:57-61)
        raise ValueError("result record carries status %d: not a planning result" % status)
    exhausted = status == 1
    y = np.full((3, Hp), np.nan)
    shapes: List[np.ndarray] = []
    trims = np.zeros(Hp, dtype=np.int64)
    path = np.zeros(Hp + 1, dtype=np.int64)
    tree = None
    if not exhausted:
        y[:, :] = rec["y_predicted"][:Hp].T
        trims[:] = rec["predicted_trims"][:Hp]
        path[:] = rec["tree_path"][: Hp + 1]
        for k in range(Hp):
            nc = int(rec["shape_cols"][k])
            shapes.append(np.array(rec["shapes"][k][:, :nc], dtype=np.float64))
        rows = np.array(rec["path_nodes"][: Hp + 1])
        tree = Tree(
            x=rows[:, 0].copy(),
            y=rows[:, 1].copy(),
            yaw=rows[:, 2].copy(),
            trim=rows[:, 3].astype(np.int64),
            g=rows[:, 4].copy(),
            h=rows[:, 5].copy(),
            k=rows[:, 6].astype(np.int64),
            parent=np.arange(0, Hp + 1, dtype=np.uint32),
        )
    return ControlResultsInfo(
        tree=tree,
        tree_path=path,
        n_expanded=int(rec["n_expanded"]),
        shapes=shapes,
        predicted_trims=trims,
        y_predicted=y,
        is_exhausted=exhausted,
        needs_fallback=False,
        n_popped=int(rec["n_popped"]),
        status=status,
    )
