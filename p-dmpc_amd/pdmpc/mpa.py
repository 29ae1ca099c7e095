"""Motion-primitive automaton tables — the read-only input of the search kernel.

The reference builds these offline in MATLAB (hlc/model/motion_primitive_automaton/**) with `ode45`
and `polyshape` and caches them in `library/*.mat`.  MATLAB is not available to this backend, so this
module regenerates the fields the optimizer reads:

    trims(i).{steering, speed}            choose_trims.m:12-131, build_mpa.m:1-72, generate_trim.m
    maneuvers{i,j}.{dx, dy, dyaw, area, area_without_offset, area_large_offset}
                                          generate_maneuver.m:1-105, BicycleModel.m:26-54
    transition_matrix_single(i, j, k)     MotionPrimitiveAutomaton.m:86-87,134-136,143-145,238-250
    distance_to_equilibrium, trims_stop   MotionPrimitiveAutomaton.m:117,134-136

The ODE is integrated with classical RK4 on 50 sub-steps per tick (global error ~1e-15), where the
reference uses ode45 at RelTol = AbsTol = 1e-8: tables agree with MATLAB's to ~1e-8.  They are *inputs*
of the hot path, so kernel-vs-oracle parity does not depend on that difference.  Reachable sets
(MotionPrimitiveAutomaton.m:252-394) are used only by the reference's coupler and are out of scope.
"""
import math
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from .config import Config, MpaType

# scenarios/Vehicle.m:10-13
VEHICLE_LENGTH = 0.22
VEHICLE_WIDTH = 0.1
VEHICLE_LF = 0.1
VEHICLE_LR = 0.1


@dataclass
class Trim:
    steering: float
    speed: float


@dataclass
class Maneuver:
    xs: np.ndarray
    ys: np.ndarray
    yaws: np.ndarray
    dx: float
    dy: float
    dyaw: float
    area: np.ndarray
    area_without_offset: np.ndarray
    area_large_offset: np.ndarray


def _linspace(a, b, n):
    return [a + (b - a) * i / (n - 1) for i in range(n)]


def choose_trims(mpa_type: MpaType, max_acc_per_dt: float, max_dec_per_dt: float):
    """(trim_inputs (n,2) [steering, speed], trim_adjacency (n,n)) — choose_trims.m:1-137."""
    if mpa_type == MpaType.single_speed:  # :12-35
        n_half = 5
        steering = _linspace(-0.6, 0.6, 2 * n_half + 1)
        v_profile = [0.1 * i for i in range(9)]  # 0:0.1:0.8
        speed_left = v_profile[-n_half:]
        speed = speed_left + [0.8] + speed_left[::-1]
        inputs = [[0.0, 0.0]] + [[st, sp] for st, sp in zip(steering, speed)]
        n = len(inputs)
        adj = np.ones((n, n), dtype=np.uint8)
        for i in range(1, n):
            for j in range(1, n):
                if abs(i - j) >= 2:
                    adj[i, j] = 0
        return np.array(inputs), adj
    if mpa_type == MpaType.triple_speed:  # :36-84
        n_sixth = 5
        steering = _linspace(-0.6, 0.6, 2 * n_sixth + 1)
        n_third = len(steering)
        speed = [0.5] * n_third + [0.7] * n_third + [0.9] * n_third
        inputs = [[0.0, 0.0]] + [[st, sp] for st, sp in zip(steering * 3, speed)]
        n = len(inputs)
        adj = np.ones((n, n), dtype=np.uint8)
        for i in range(1, n):
            for j in range(1, n):
                if abs(i - j) >= 2:
                    adj[i, j] = 0
        # 1-based indices below follow the MATLAB source literally; [a-1] converts
        adj[0, n_third + 1 :] = 0  # :69
        adj[n_third + 1 :, 0] = 0  # :70
        for base in (1 + n_third, 1 + 2 * n_third):  # :71-74
            adj[base - 1, base] = 0
            adj[base, base - 1] = 0
        firsts = list(range(2, n_third + 2)) + list(range(n_third + 2, 2 * n_third + 2))  # :76
        seconds = list(range(n_third + 2, 2 * n_third + 2)) + list(range(2 * n_third + 2, n + 1))
        for i, j in zip(firsts, seconds):  # :78-83
            adj[i - 1, j - 1] = 1
            adj[j - 1, i - 1] = 1
        return np.array(inputs), adj
    if mpa_type == MpaType.realistic:  # :85-131
        d_speed = min(max_acc_per_dt, max_dec_per_dt)
        acc_max = 1.05 * max_acc_per_dt
        dec_max = 1.05 * max_dec_per_dt
        speed_max = d_speed * round(0.8 / d_speed)
        n_speeds = int(math.floor(speed_max / d_speed + 1e-9)) + 1
        speed_vec = [d_speed * i for i in range(n_speeds)]
        d_steer = 0.5 * math.pi / 18
        lo = d_steer * round((3 * math.pi / 18) / d_steer)
        hi = d_steer * round((2 * math.pi / 18) / d_steer)
        d_steer_max = 1.05 * d_steer

        def sym_range(mx):
            cnt = int(round(mx / d_steer))
            return [-mx + d_steer * i for i in range(2 * cnt + 1)]

        steer_cla = [sym_range(lo)]
        x0, x1 = d_speed, speed_vec[2]
        for i_speed in (1, 2):  # :112-118 (interp1 between lo at speed 2 and hi at speed 3)
            xq = speed_vec[i_speed]
            mx = lo + (hi - lo) * (xq - x0) / (x1 - x0)
            mx = d_steer * round(mx / d_steer)
            steer_cla.append(sym_range(mx))
        for _ in range(3, n_speeds):  # :121-123
            steer_cla.append(sym_range(hi))
        # build_mpa.m:24-70
        inputs = []
        for sp, steers in zip(speed_vec, steer_cla):
            for st in steers:
                inputs.append([st, sp])
        n = len(inputs)
        adj = np.zeros((n, n), dtype=np.uint8)
        for i in range(n):
            for j in range(n):
                if abs(inputs[j][0] - inputs[i][0]) <= d_steer_max:
                    if inputs[j][1] > inputs[i][1]:
                        ok = (inputs[j][1] - inputs[i][1]) <= acc_max
                    else:
                        ok = (inputs[i][1] - inputs[j][1]) <= dec_max
                    if ok:
                        adj[i, j] = 1
        return np.array(inputs), adj
    raise ValueError("unknown mpa trim type")


def _bicycle_rhs(state, steering_derivative, acceleration):
    """BicycleModel.ode (BicycleModel.m:26-54): state = [x y yaw v delta]."""
    L = VEHICLE_LF + VEHICLE_LR
    R = VEHICLE_LR / L
    _, _, psi, v, delta = state
    beta = math.atan(R * math.tan(delta))
    return (
        v * math.cos(psi + beta),
        v * math.sin(psi + beta),
        v / L * math.tan(delta) * math.cos(beta),
        acceleration,
        steering_derivative,
    )


def _integrate(trim1: Trim, trim2: Trim, dt: float, ticks: int, substeps: int = 50):
    sd = (trim2.steering - trim1.steering) / dt  # generate_maneuver.m:7
    acc = (trim2.speed - trim1.speed) / dt  # :8
    s = (0.0, 0.0, 0.0, trim1.speed, trim1.steering)  # :11-16
    h = dt / ticks / substeps
    xs, ys, yaws = [0.0], [0.0], [0.0]
    for _ in range(ticks):
        for _ in range(substeps):
            k1 = _bicycle_rhs(s, sd, acc)
            k2 = _bicycle_rhs(tuple(a + 0.5 * h * b for a, b in zip(s, k1)), sd, acc)
            k3 = _bicycle_rhs(tuple(a + 0.5 * h * b for a, b in zip(s, k2)), sd, acc)
            k4 = _bicycle_rhs(tuple(a + h * b for a, b in zip(s, k3)), sd, acc)
            s = tuple(a + h / 6.0 * (p + 2 * q + 2 * r + w) for a, p, q, r, w in zip(s, k1, k2, k3, k4))
        xs.append(s[0])
        ys.append(s[1])
        yaws.append(s[2])
    return np.array(xs), np.array(ys), np.array(yaws)


def _translate_global(yaw, x0, y0, xl, yl):
    """utility/translate_global.m:19-22."""
    c, s = math.cos(yaw), math.sin(yaw)
    xg = [c * a + (-s) * b + x0 for a, b in zip(xl, yl)]
    yg = [s * a + c * b + y0 for a, b in zip(xl, yl)]
    return xg, yg


def _maneuver_area(x1, y1, x2, y2, signum, non_convex):
    """get_maneuver_area (generate_maneuver.m:68-105); corner indices are 1-based in the comments."""
    if signum == 0:  # :74-76
        cols = [(x1[0], y1[0]), (x1[1], y1[1]), (x2[2], y2[2]), (x2[3], y2[3]), (x1[0], y1[0])]
    elif signum > 0:  # turn left :77-89
        if non_convex:
            cols = [(x1[0], y1[0]), (x1[1], y1[1]), (x2[1], y2[1]), (x2[2], y2[2]), (x2[3], y2[3]), (x1[3], y1[3]), (x1[0], y1[0])]
        else:
            cols = [(x1[0], y1[0]), (x1[1], y1[1]), (x2[2], y2[2]), (x2[3], y2[3]), (x2[3], y1[3]), (x1[0], y1[0])]
    else:  # turn right :91-101
        if non_convex:
            cols = [(x1[0], y1[0]), (x1[1], y1[1]), (x1[2], y1[2]), (x2[2], y2[2]), (x2[3], y2[3]), (x2[0], y2[0]), (x1[0], y1[0])]
        else:
            cols = [(x1[0], y1[0]), (x1[1], y1[1]), (x2[2], y1[2]), (x2[2], y2[2]), (x2[3], y2[3]), (x1[0], y1[0])]
    return np.array(cols, dtype=np.float64).T.copy()


def generate_maneuver(trim1: Trim, trim2: Trim, options: Config) -> Maneuver:
    """generate_maneuver.m:1-66."""
    xs, ys, yaws = _integrate(trim1, trim2, options.dt_seconds, options.tick_per_step)
    dx, dy, dyaw = float(xs[-1]), float(ys[-1]), float(yaws[-1])
    signum = (dyaw > 0) - (dyaw < 0)
    non_convex = options.are_any_obstacles_non_convex
    areas = []
    for off_l, off_w in ((options.offset, options.offset), (0.0, 0.0), (0.05, 0.0)):  # :39-41, :49-50, :58-59
        xr = [sgn * (VEHICLE_LENGTH / 2 + off_l) for sgn in (-1, -1, 1, 1)]
        yr = [sgn * (VEHICLE_WIDTH / 2 + off_w) for sgn in (-1, 1, 1, -1)]
        x2, y2 = _translate_global(dyaw, dx, dy, xr, yr)
        a = _maneuver_area(xr, yr, x2, y2, signum, non_convex)
        assert np.all(a[:, 0] == a[:, -1])  # must be a closed shape (:46)
        areas.append(a)
    return Maneuver(xs, ys, yaws, dx, dy, dyaw, areas[0], areas[1], areas[2])


class MotionPrimitiveAutomaton:
    """The fields of MotionPrimitiveAutomaton.m:5-17 that the optimizer and its callers read."""

    def __init__(self, options: Config):
        acc = 0.64 * options.dt_seconds  # MotionPrimitiveAutomaton.m:38-41
        inputs, adjacency = choose_trims(options.mpa_type, acc, acc)
        n = inputs.shape[0]
        self.Hp = options.Hp
        self.n_trims = n
        self.recursive_feasibility = options.recursive_feasibility
        self.trims: List[Trim] = [Trim(float(st), float(sp)) for st, sp in inputs]  # :112-114
        self.trims_stop = [i + 1 for i, t in enumerate(self.trims) if t.speed == 0]  # :117 (1-based)
        self.maneuvers: List[List[Optional[Maneuver]]] = [[None] * n for _ in range(n)]
        for i in range(n):  # :119-131
            for j in range(n):
                if adjacency[i, j]:
                    self.maneuvers[i][j] = generate_maneuver(self.trims[i], self.trims[j], options)
        # distance_to_equilibrium: hop count in the undirected trim graph to the nearest zero-speed trim (:134-136)
        und = (adjacency | adjacency.T).astype(bool)
        dist = np.full(n, np.iinfo(np.int32).max, dtype=np.int64)
        frontier = [i for i, t in enumerate(self.trims) if t.speed == 0]
        for i in frontier:
            dist[i] = 0
        d = 0
        while frontier:
            d += 1
            nxt = []
            for i in frontier:
                for j in np.nonzero(und[i])[0]:
                    if dist[j] > d:
                        dist[j] = d
                        nxt.append(int(j))
            frontier = nxt
        self.distance_to_equilibrium = dist
        # transition_matrix_single (n, n, Hp) with the recursive-feasibility mask (:86-87, :143-145, :238-250)
        T = np.repeat(adjacency[:, :, None], options.Hp, axis=2).astype(np.uint8)
        if options.recursive_feasibility:
            N = options.Hp
            for k in range(1, N + 1):
                k_to_go = N - k
                T[:, dist > k_to_go, k - 1] = 0
        self.transition_matrix_single = T
        self.adjacency = adjacency

    def get_max_speed_of_mpa(self) -> float:  # :182-185
        return max(t.speed for t in self.trims)

    def get_straight_speeds_of_mpa(self) -> List[float]:  # :187-191
        return [t.speed for t in self.trims if t.speed > 0 and t.steering == 0]

    def trim_from_values(self, speed: float, steering: float) -> int:
        """1-based index of the closest trim (:193-236)."""
        sp = np.array([t.speed for t in self.trims])
        st = np.array([t.steering for t in self.trims])
        if steering == 0:
            idx = np.nonzero(st == 0)[0]
            return int(idx[int(np.argmin(np.abs(sp - speed)[idx]))]) + 1
        sp_c, sp_s = sp.min(), sp.max() - sp.min()
        st_c, st_s = st.min(), st.max() - st.min()
        d = np.hypot((sp - sp_c) / sp_s - (speed - sp_c) / sp_s, (st - st_c) / st_s - (steering - st_c) / st_s)
        return int(np.argmin(d)) + 1


_CACHE = {}


def get_mpa(options: Config) -> MotionPrimitiveAutomaton:
    """Cached construction, the analogue of the reference's library/*.mat cache (MotionPrimitiveAutomaton.m:67-79);
    the key follows FileNameConstructor.get_mpa_name (utility/FileNameConstructor.m:14-47)."""
    key = (
        options.mpa_type,
        options.Hp,
        options.dt_seconds,
        options.are_any_obstacles_non_convex,
        options.recursive_feasibility,
        options.offset,
        options.time_per_tick,
    )
    if key not in _CACHE:
        _CACHE[key] = MotionPrimitiveAutomaton(options)
    return _CACHE[key]
