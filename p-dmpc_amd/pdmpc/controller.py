"""Host-side step driver: the caller of the optimizer boundary.

Restates what the reference does around `run_optimizer` for prioritized planning so whole MPC steps can
be produced and timed without MATLAB:

    traffic info per step        HighLevelController.update_controlled_vehicles_traffic_info (HighLevelController.m:167-270)
    coupling                     Coupler (full_coupling: Coupler.m:31-32; distance: DistanceCoupler.m:15-50)
    priorities -> DAG            ConstantPrioritizer.m:14-20, Prioritizer.directed_coupling_from_priorities (Prioritizer.m:64-77)
    computation levels           utility/kahn.m:1-24
    level loop                   PrioritizedSequentialController.controller (PrioritizedSequentialController.m:77-94)
    obstacle assembly            PrioritizedController.plan / consider_predecessors / consider_successors
                                 (PrioritizedController.m:297-324, 449-566)
    exhaustion handling          handle_graph_search_exhaustion / plan_fallback (PrioritizedController.m:568-616, 678-718)
    plant                        Simulation.apply (plant/Simulation.m:86-100)

The planner is injected (`plan_level`: list[VehicleIter] -> list[ControlResultsInfo]) — the product passes
GraphSearchHip.run_optimizer_batch; tests may pass the CPU oracle to obtain the expected closed loop.
ROS 2 messaging is replaced by in-memory hand-off of `info.shapes`.
"""
import math
from dataclasses import dataclass
from typing import Callable, List, Optional

import numpy as np

from .config import Config, ConstraintFromSuccessor, ScenarioType
from .iteration_data import ControlResultsInfo, Tree, VehicleIter
from .reference_trajectory import get_occupied_areas, get_reference_trajectory


def kahn(A: np.ndarray) -> np.ndarray:
    """utility/kahn.m:1-24: computation level (1-based) of every vertex of the DAG A (A[i,j]=1: i before j)."""
    A = np.array(A, dtype=np.int64)
    n = A.shape[0]
    L = np.zeros(n, dtype=np.int64)
    in_d = A.sum(axis=0)
    done = np.zeros(n, dtype=bool)
    level = 1
    while not done.all():
        src = in_d == 0
        if not src.any():
            raise ValueError("coupling graph has a cycle")
        L[src] = level
        A[src, :] = 0
        done[src] = True
        in_d = A.sum(axis=0)
        in_d[done] = 1
        level += 1
    return L


def directed_coupling_from_priorities(adjacency, priorities):
    """Prioritizer.m:64-77: keep edge i->j only if priority(i) <= priority(j) (lower number plans first)."""
    p = np.asarray(priorities)
    removed = p[None, :] < p[:, None]
    d = np.array(adjacency, dtype=np.int64)
    d[removed] = 0
    return d


def del_first_rpt_last(seq, n=1):
    """utility del_first_rpt_last: drop the first n entries, repeat the last one n times."""
    seq = list(seq)
    n = min(n, len(seq))
    return seq[n:] + [seq[-1]] * n


@dataclass
class Measurement:  # plant PlantMeasurement
    x: float
    y: float
    yaw: float
    speed: float
    steering: float


class PrioritizedSequentialController:
    def __init__(
        self,
        options: Config,
        scenario,
        mpa,
        plan_level: Callable[[List[VehicleIter]], List[ControlResultsInfo]],
        coupling: str = "full",
        priorities: Optional[List[int]] = None,
        priority_strategy: str = "constant",
        boundary_provider=None,
        weight_strategy: str = "distance",
    ):
        self.options = options
        self.scenario = scenario
        self.mpa = mpa
        self.plan_level = plan_level
        self.coupling = coupling
        self.n = options.amount
        self.priorities = list(priorities) if priorities is not None else list(range(1, self.n + 1))
        # "constant" (ConstantPrioritizer.m), "coloring" (ColoringPrioritizer.m), "random" (RandomPrioritizer.m),
        # "fca" (FcaPrioritizer.m)
        self.priority_strategy = priority_strategy
        # "constant" | "distance" | "random" (weight/*.m; Config.m:25 defaults to distance); only matters when the
        # coupling DAG is deeper than options.max_num_CLs and has to be cut
        self.weight_strategy = weight_strategy
        self.boundary_provider = boundary_provider  # road networks: (vehicle, path, points_index, cpi) -> (left, right)
        # Simulation.setup: initial speed = steering = 0 (Simulation.m:52-65)
        self.meas = [Measurement(v.x_start, v.y_start, v.yaw_start, 0.0, 0.0) for v in scenario.vehicles]
        self.k = 0
        self.info_old: List[Optional[ControlResultsInfo]] = [None] * self.n
        self.infos: List[Optional[ControlResultsInfo]] = [None] * self.n
        self.last_iters: List[Optional[VehicleIter]] = [None] * self.n
        self.last_levels = None

    # ---- HighLevelController.update_controlled_vehicles_traffic_info (HighLevelController.m:167-270)
    def _traffic_info(self):
        o = self.options
        n = self.n
        self.x0 = np.zeros((n, 4))
        self.trims = np.zeros(n, dtype=np.int64)
        self.occupied = [None] * n
        self.ref_points = [None] * n
        self.v_ref = [None] * n
        self.boundary = [(None, None)] * n
        for i, m in enumerate(self.meas):
            veh = self.scenario.vehicles[i]
            self.x0[i] = [m.x, m.y, m.yaw, m.speed]
            self.trims[i] = self.mpa.trim_from_values(m.speed, m.steering)
            self.occupied[i] = get_occupied_areas(m.x, m.y, m.yaw, veh.Length, veh.Width, o.offset)
            path, points_index, v_ref, cpi = get_reference_trajectory(
                self.mpa, veh.reference_path, veh.reference_speed, m.x, m.y, int(self.trims[i]), o.dt_seconds
            )
            self.ref_points[i] = path
            self.v_ref[i] = v_ref
            if o.scenario_type != ScenarioType.circle and self.boundary_provider is not None:
                self.boundary[i] = self.boundary_provider(i, veh, points_index, cpi)

    def _couple(self):
        n = self.n
        if self.coupling == "full":  # Coupler.m:31-32
            return np.ones((n, n), dtype=np.int64) - np.eye(n, dtype=np.int64)
        if self.coupling == "none":
            return np.zeros((n, n), dtype=np.int64)
        if self.coupling == "distance":  # DistanceCoupler.m:15-50 (without the lanelet-adjacency pre-filter)
            adj = np.zeros((n, n), dtype=np.int64)
            max_distance = 2 * self.mpa.get_max_speed_of_mpa() * self.options.dt_seconds * self.options.Hp
            for a in range(n):
                for b in range(a + 1, n):
                    d = math.hypot(self.x0[a, 0] - self.x0[b, 0], self.x0[a, 1] - self.x0[b, 1])
                    adj[a, b] = adj[b, a] = int(d <= max_distance)
            return adj
        raise ValueError(self.coupling)

    # ---- PrioritizedController.plan: obstacle assembly (PrioritizedController.m:297-324)
    def _iter_for(self, i, directed, directed_seq, device_handoff=False):
        """device_handoff: sequential predecessors are NOT expanded into polygons here; the kernel appends their
        solved areas on the device (pdmpc_pack_step), so the whole step is one launch."""
        o = self.options
        Hp = o.Hp
        predecessors = [j for j in range(self.n) if directed[j, i] == 1]
        predecessors_seq = [j for j in range(self.n) if directed_seq[j, i]]
        successors = [j for j in range(self.n) if directed[i, j] == 1]
        dyn = []
        for j in predecessors:  # consider_predecessors :449-506
            if j in predecessors_seq:
                if device_handoff:
                    continue
                dyn.append(list(self.infos[j].shapes))  # this step's /vehicle_prediction :476-491
            else:
                old = self.info_old[j]  # parallel_coupling_previous_trajectory :409-447
                if old is not None and self.k > 1:
                    dyn.append(del_first_rpt_last(old.shapes, 1))
        obstacles = list(self.scenario.obstacles)
        for j in successors:  # consider_successors :508-566
            if o.constraint_from_successor == ConstraintFromSuccessor.area_of_standstill:
                if abs(self.x0[j, 3]) < 0.01:  # :536-540
                    obstacles.append(self.occupied[j][0])
            elif o.constraint_from_successor == ConstraintFromSuccessor.area_of_previous_trajectory:
                old = self.info_old[j]
                if old is not None:  # :542-557 (message of the previous step, shifted once)
                    dyn.append(del_first_rpt_last(old.shapes, 1))
        scen_dyn = [list(r) for r in self.scenario.dynamic_obstacle_area]
        return VehicleIter(
            x0=self.x0[i].copy(),
            trim_index=int(self.trims[i]),
            reference_trajectory_points=self.ref_points[i],
            v_ref=self.v_ref[i],
            predicted_lanelet_boundary=self.boundary[i],
            obstacles=obstacles,
            dynamic_obstacle_area=scen_dyn + dyn,
        )

    # ---- handle_graph_search_exhaustion (PrioritizedController.m:568-616)
    def _standstill_info(self, i, it: VehicleIter, info: ControlResultsInfo) -> ControlResultsInfo:
        o = self.options
        Hp = o.Hp
        x, y, yaw = it.x0[0], it.x0[1], it.x0[2]
        cost = 0.0
        g = [0.0]
        for s in range(Hp):
            r = self.ref_points[i][s]
            cost = cost + math.hypot(r[0] - x, r[1] - y) ** 2
            g.append(cost)
        _, rect = get_occupied_areas(x, y, yaw, self.scenario.vehicles[0].Length, self.scenario.vehicles[0].Width, o.offset)
        tree = Tree(
            x=np.full(Hp + 1, x), y=np.full(Hp + 1, y), yaw=np.full(Hp + 1, yaw),
            trim=np.full(Hp + 1, it.trim_index, dtype=np.int64), k=np.arange(Hp + 1), g=np.array(g),
            h=np.full(Hp + 1, -1.0), parent=np.arange(Hp + 1, dtype=np.uint32),
        )
        info.tree = tree
        info.tree_path = np.arange(1, Hp + 2)
        info.y_predicted = np.tile(np.array([[x], [y], [yaw]]), (1, Hp))
        info.shapes = [rect.copy() for _ in range(Hp)]  # transformed_rectangle closed, no offset :602-611
        info.predicted_trims = np.full(Hp, it.trim_index, dtype=np.int64)
        info.needs_fallback = False
        return info

    # ---- plan_fallback (PrioritizedController.m:678-718): shifted previous plan
    def _fallback_info(self, i, info: ControlResultsInfo) -> ControlResultsInfo:
        old = self.info_old[i]
        if old is None:
            raise RuntimeError("vehicle %d needs a fallback in the first step" % (i + 1))
        info.shapes = del_first_rpt_last(old.shapes)
        info.predicted_trims = np.array(del_first_rpt_last(list(old.predicted_trims)))
        info.y_predicted = np.array(del_first_rpt_last(list(old.y_predicted.T))).T
        info.tree = old.tree
        info.tree_path = old.tree_path
        info.needs_fallback = True
        return info

    def _post_plan(self, i, it, info):
        self.last_iters[i] = it
        if info.is_exhausted:  # PrioritizedController.m:344-352
            standstill = self.mpa.trims[it.trim_index - 1].speed == 0
            if standstill and self.options.constraint_from_successor != ConstraintFromSuccessor.none:
                info = self._standstill_info(i, it, info)
            else:
                info = self._fallback_info(i, info)
        self.infos[i] = info

    def _published_on_exhaustion(self, i):
        """The Hp areas vehicle i publishes if its search is exhausted: its standstill rectangle
        (PrioritizedController.m:602-611) or the previous plan shifted by one step (:678-718)."""
        o = self.options
        standstill = self.mpa.trims[int(self.trims[i]) - 1].speed == 0
        if standstill and o.constraint_from_successor != ConstraintFromSuccessor.none:
            return [self.occupied[i][1].copy() for _ in range(o.Hp)]
        if self.info_old[i] is not None:
            return del_first_rpt_last(self.info_old[i].shapes)
        return None

    # ---- HighLevelController.handle_others_fallback (HighLevelController.m:449-463) with
    #      PrioritizedController.check_others_fallback (PrioritizedController.m:623-676)
    def _handle_others_fallback(self):
        """A vehicle that did not fall back itself still takes its fallback (the shifted previous plan, plan_fallback with
        is_fallback_while_planning = false) if a fallback vehicle reaches it in the coupling graph from which the outgoing
        sequential edges of the fallback vehicles -- the ones their successors have already planned against -- are removed."""
        fallbacks = np.array([bool(info.needs_fallback) for info in self.infos])
        if not fallbacks.any():
            return
        adjacency = np.asarray(self.last_adjacency, dtype=np.int64)
        seq = np.asarray(self.last_directed_seq, dtype=np.int64)
        outgoing = seq.copy()
        outgoing[~fallbacks, :] = 0  # :653-654
        fallback_matrix = adjacency - (outgoing + outgoing.T)  # :655-656
        reached = np.zeros(self.n, dtype=bool)
        for f in np.flatnonzero(fallbacks):  # shortestpath(fallback_graph, f, i) non-empty  :661-674
            seen = np.zeros(self.n, dtype=bool)
            seen[f] = True
            stack = [int(f)]
            while stack:
                a = stack.pop()
                for b in np.flatnonzero(fallback_matrix[a] != 0):
                    if not seen[b]:
                        seen[b] = True
                        stack.append(int(b))
            reached |= seen
        for i in range(self.n):
            if reached[i] and not fallbacks[i]:
                info = self._fallback_info(i, self.infos[i])
                info.needs_fallback = False  # plan_fallback(is_fallback_while_planning = false)  :717
                self.infos[i] = info

    def _direct(self, adjacency, priorities):
        """Undirected coupling -> directed coupling (Prioritizer.prioritize): explicit priorities win; otherwise the
        controller's strategy."""
        if priorities is None and self.priority_strategy == "coloring":
            from .prioritizer import coloring_directed_coupling

            return coloring_directed_coupling(adjacency)[0].astype(np.int64)
        if priorities is None and self.priority_strategy == "random":
            from .prioritizer import random_priorities

            priorities = random_priorities(self.n, self.k)
        if priorities is None and self.priority_strategy == "fca":
            from .prioritizer import fca_priorities

            veh = self.scenario.vehicles[0]
            priorities, _ = fca_priorities(
                adjacency, self.ref_points, veh.Length, veh.Width, self.options.offset,
                self.scenario.obstacles, self.scenario.dynamic_obstacle_area,
            )
        return directed_coupling_from_priorities(adjacency, self.priorities if priorities is None else priorities)

    def _group(self, directed):
        """PrioritizedController.group (PrioritizedController.m:375-389): weigh the directed couplings and keep as
        sequential only what fits into options.max_num_CLs computation levels; the remaining couplings are parallel
        (the successor avoids the predecessor's previous plan, :409-447)."""
        directed = np.asarray(directed)
        if int(kahn(directed).max()) <= self.options.max_num_CLs:
            # every sub-graph of the DAG is at most as deep, so GreedyCutter accepts every edge (GreedyCutter.m:64-82)
            return directed != 0
        from . import grouping

        o = self.options
        if self.weight_strategy == "constant":
            W = grouping.constant_weight(directed)
        elif self.weight_strategy == "random":
            W = grouping.random_weight(directed, self.k)
        elif self.weight_strategy == "distance":
            W = grouping.distance_weight(directed, self.x0, self.mpa.get_max_speed_of_mpa(), o.dt_seconds, o.Hp)
        else:
            raise ValueError(self.weight_strategy)
        return grouping.greedy_cut(W, o.max_num_CLs)

    def build_step_problem(self, priorities=None, refresh=True, couplings=None):
        """Everything one launch needs to plan the whole time step: vehicles in level order (slot = position),
        per-slot predecessor slots, per-slot areas to publish on exhaustion.  `priorities` overrides the controller's
        own; `couplings` = (directed, directed_sequential) replaces the prioritization altogether (the explorative driver builds one
        problem per prioritization of the same traffic state by swapping single couplings of the base one)."""
        if refresh:
            self._traffic_info()
            self.last_adjacency = self._couple()
        adjacency = self.last_adjacency
        if couplings is not None:
            directed, directed_seq = np.asarray(couplings[0]) != 0, np.asarray(couplings[1]) != 0
        else:
            directed = self._direct(adjacency, priorities)
            directed_seq = self._group(directed)
        self.last_directed = np.asarray(directed, dtype=np.int64)
        self.last_directed_seq = np.asarray(directed_seq, dtype=np.int64)
        levels = kahn(directed_seq)
        self.last_levels = levels
        order = sorted(range(self.n), key=lambda i: (int(levels[i]), i))
        slot_of = {v: s for s, v in enumerate(order)}
        iters = [self._iter_for(i, directed, directed_seq, device_handoff=True) for i in order]
        preds = [[slot_of[j] for j in range(self.n) if directed_seq[j, i]] for i in order]
        fallback = [self._published_on_exhaustion(i) for i in order]
        level_sizes = [int(np.sum(levels == l)) for l in range(1, int(levels.max()) + 1)]
        return {"order": order, "iters": iters, "preds": preds, "fallback": fallback, "level_sizes": level_sizes, "levels": [int(levels[i]) for i in order],
                "directed_seq": np.array(directed_seq, dtype=np.int64)}

    def step(self, plan_step=None):
        """One pass of HighLevelController.main_control_loop (HighLevelController.m:334-373) in simulation.
        With `plan_step(problem) -> list[ControlResultsInfo]` (slot order) the whole step is planned by one call
        (one kernel launch, hand-off of solved areas on the device); otherwise levels are planned one by one."""
        self.k += 1
        self.infos = [None] * self.n
        if plan_step is not None:
            prob = self.build_step_problem()
            self.last_problem = prob
            results = plan_step(prob)
            for s, i in enumerate(prob["order"]):
                self._post_plan(i, prob["iters"][s], results[s])
        else:
            self._traffic_info()
            adjacency = self._couple()
            self.last_adjacency = adjacency
            directed = self._direct(adjacency, None)
            directed_seq = self._group(directed)
            self.last_directed_seq = np.asarray(directed_seq, dtype=np.int64)
            levels = kahn(directed_seq)
            self.last_levels = levels
            for lvl in range(1, int(levels.max()) + 1):  # PrioritizedSequentialController.m:83-91
                members = [i for i in range(self.n) if levels[i] == lvl]
                iters = [self._iter_for(i, directed, directed_seq) for i in members]
                # the sampled optimizer draws from a stream seeded with time_step + vehicle_index (MonteCarloTreeSearch.m:32):
                # a planner that wants them declares it (plan_level.wants_seeds)
                if getattr(self.plan_level, "wants_seeds", False):
                    results = self.plan_level(iters, [self.k + (i + 1) for i in members])
                else:
                    results = self.plan_level(iters)
                for i, it, info in zip(members, iters, results):
                    self._post_plan(i, it, info)
        self._handle_others_fallback()
        # Simulation.apply (Simulation.m:86-100)
        for i, info in enumerate(self.infos):
            t = self.mpa.trims[int(info.predicted_trims[0]) - 1]
            self.meas[i] = Measurement(float(info.y_predicted[0, 0]), float(info.y_predicted[1, 0]), float(info.y_predicted[2, 0]), t.speed, t.steering)
        self.info_old = list(self.infos)
        return self.infos
