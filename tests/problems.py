"""Seeded synthetic optimizer inputs for parity tests (shared by the CPU-only and the GPU tests).

Every generator returns plain VehicleIter objects; the same objects are handed to the oracle and to the
HIP backend, so both see byte-identical inputs.
"""
import math

import numpy as np

from pdmpc.config import Config, MpaType, ScenarioType
from pdmpc.iteration_data import VehicleIter
from pdmpc.mpa import get_mpa
from pdmpc.reference_trajectory import get_occupied_areas, get_reference_trajectory


def rect(x, y, yaw, length, width):
    a, _ = get_occupied_areas(x, y, yaw, length, width, 0.0)
    return a


def corridor(rng, x0=-1.0, x1=7.0, half_width=0.28, n=36, wobble=0.0):
    xs = np.linspace(x0, x1, n)
    phase = rng.uniform(0, 2 * math.pi)
    yc = wobble * np.sin(xs * 1.3 + phase)
    left = np.vstack([xs, yc + half_width])
    right = np.vstack([xs, yc - half_width])
    return left, right, np.column_stack([xs, yc])


def road_problem(rng, options: Config, mpa, n_dyn=None, n_static=None, with_boundary=True, convex=False, n_hdv=0):
    """One vehicle on a (slightly wobbling) corridor with crossing/oncoming obstacle polygons."""
    Hp = options.Hp
    wob = rng.uniform(0.0, 0.08)
    left, right, centre = corridor(rng, wobble=wob, half_width=rng.uniform(0.2, 0.4))
    x = rng.uniform(0.0, 1.0)
    y = float(np.interp(x, centre[:, 0], centre[:, 1])) + rng.uniform(-0.03, 0.03)
    yaw = rng.uniform(-0.15, 0.15)
    # a moving or standing start
    stand = rng.random() < 0.35
    if stand:
        trim = 1
    else:
        straight = [i + 1 for i, t in enumerate(mpa.trims) if t.steering == 0 and t.speed > 0]
        trim = int(rng.choice(straight))
    speeds = mpa.get_straight_speeds_of_mpa()
    ref_speed = float(rng.choice(speeds))
    path, _, v_ref, _ = get_reference_trajectory(mpa, centre, ref_speed, x, y, trim, options.dt_seconds)
    n_dyn = int(rng.integers(0, 5)) if n_dyn is None else n_dyn
    n_static = int(rng.integers(0, 3)) if n_static is None else n_static
    dyn = []
    for _ in range(n_dyn):
        ox = x + rng.uniform(0.5, 2.5)
        oy = float(np.interp(ox, centre[:, 0], centre[:, 1])) + rng.uniform(-0.25, 0.25)
        vx = rng.uniform(-0.6, 0.3) * options.dt_seconds
        vy = rng.uniform(-0.2, 0.2) * options.dt_seconds
        oyaw = rng.uniform(-math.pi, math.pi)
        row = []
        for k in range(Hp):
            r = rect(ox + vx * (k + 1), oy + vy * (k + 1), oyaw, 0.24 + abs(vx), 0.12)
            if (not convex) and rng.random() < 0.4:
                # a 7-column non-convex "L" sweep like generate_maneuver.m:80-83
                r = np.column_stack([r[:, 0], r[:, 1], r[:, 1] + [0.05, 0.06], r[:, 2], r[:, 3], r[:, 3] - [0.02, 0.0], r[:, 0]])
            row.append(r)
        dyn.append(row)
    obstacles = []
    for _ in range(n_static):
        ox = x + rng.uniform(0.8, 3.0)
        oy = float(np.interp(ox, centre[:, 0], centre[:, 1])) + rng.uniform(-0.3, 0.3)
        obstacles.append(rect(ox, oy, rng.uniform(-math.pi, math.pi), 0.24, 0.12))
    hdv = []
    for _ in range(n_hdv):  # reachable sets of adjacent human-driven vehicles: larger polygons growing with the step
        hx = x + rng.uniform(0.6, 2.0)
        hy = float(np.interp(hx, centre[:, 0], centre[:, 1])) + rng.uniform(-0.2, 0.2)
        hyaw = rng.uniform(-math.pi, math.pi)
        hdv.append([rect(hx, hy, hyaw, 0.3 + 0.08 * k, 0.15 + 0.03 * k) for k in range(Hp)])
    boundary = (left, right) if with_boundary else (None, None)
    return VehicleIter(
        hdv_reachable_sets=hdv,
        x0=np.array([x, y, yaw, mpa.trims[trim - 1].speed]),
        trim_index=trim,
        reference_trajectory_points=path,
        v_ref=v_ref,
        predicted_lanelet_boundary=boundary,
        obstacles=obstacles,
        dynamic_obstacle_area=dyn,
    )


def make_options(mode, Hp=6, mpa_type=MpaType.single_speed, **kw):
    """mode 'interx' -> road-network settings (non-convex areas); 'sat' -> circle settings (convex areas)."""
    st = ScenarioType.commonroad if mode == "interx" else ScenarioType.circle
    return Config(scenario_type=st, Hp=Hp, mpa_type=mpa_type, **kw)


def problem_set(mode, seed, count, Hp=6, mpa_type=MpaType.single_speed, n_hdv=0, with_boundary=True, **kw):
    options = make_options(mode, Hp=Hp, mpa_type=mpa_type, **kw)
    mpa = get_mpa(options)
    rng = np.random.default_rng(seed)
    iters = [road_problem(rng, options, mpa, convex=(mode == "sat"), n_hdv=n_hdv, with_boundary=with_boundary) for _ in range(count)]
    return options, mpa, iters


def symmetric_problem(options, mpa, block_x=0.9, half=0.06, length=0.3):
    """A vehicle on a straight reference along the x-axis with an obstacle centred on the axis: the left and the right
    half of the search tree mirror each other bit for bit, so the open list keeps popping TIED minimal keys — the case
    in which the pop order depends on the layout of the reference's binary heap."""
    Hp = options.Hp
    trim = [i + 1 for i, t in enumerate(mpa.trims) if t.steering == 0 and t.speed > 0][0]
    v = mpa.trims[trim - 1].speed
    ref = np.column_stack([v * options.dt_seconds * np.arange(1, Hp + 1), np.zeros(Hp)])
    obst = np.array([[block_x, block_x + length, block_x + length, block_x], [-half, -half, half, half]])
    return VehicleIter(
        x0=np.array([0.0, 0.0, 0.0, v]),
        trim_index=trim,
        reference_trajectory_points=ref,
        v_ref=np.full(Hp, v),
        predicted_lanelet_boundary=(None, None),
        obstacles=[obst],
        dynamic_obstacle_area=[],
    )


def tied_pops(trace):
    """Number of pops of an oracle trace at which the minimal key of the open list was not unique."""
    from sortedcontainers import SortedList

    f = trace.tree["g"] + trace.tree["h"]
    par = trace.tree["parent"]
    order = np.argsort(par, kind="stable")
    start = np.searchsorted(par[order], np.arange(1, len(f) + 2), side="left")
    open_list = SortedList([(f[0], 1)])
    tied = 0
    for nd in trace.pops:
        tied += len(open_list) > 1 and open_list[1][0] == open_list[0][0]
        open_list.remove((f[nd - 1], int(nd)))
        for c in order[start[nd - 1] : start[nd]]:
            open_list.add((f[c], int(c) + 1))
    return tied
