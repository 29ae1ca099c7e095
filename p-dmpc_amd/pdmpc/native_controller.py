"""ctypes wrapper of the native step controller (csrc/step_controller.cpp, declared in include/pdmpc.h).

`NativeController` is the C++ twin of `pdmpc.controller.PrioritizedSequentialController` (constant and colouring priorities,
full / distance coupling, GreedyCutter grouping): a whole MPC time step — traffic info, coupling, levels, obstacle assembly,
one kernel launch, fallbacks, plant update — is one C call (`pdmpc_controller_step`), so a closed loop has no interpreter
on its critical path.  The scenario is handed over once as flat arrays.
"""
import ctypes as C

import numpy as np

from . import abi
from .backend import BackendError, load_library
from .config import ConstraintFromSuccessor

COUPLING = {"full": 0, "distance": 1, "none": 2}
PRIORITY = {"constant": 0, "coloring": 1}
WEIGHT = {"distance": 0, "constant": 1}
SUCCESSOR = {ConstraintFromSuccessor.none: 0, ConstraintFromSuccessor.area_of_standstill: 1, ConstraintFromSuccessor.area_of_previous_trajectory: 2}


class ControllerConfig(C.Structure):
    _fields_ = [
        ("Hp", C.c_int32), ("coupling", C.c_int32), ("priority_strategy", C.c_int32), ("weight_strategy", C.c_int32), ("max_num_CLs", C.c_int32),
        ("constraint_from_successor", C.c_int32), ("dt_seconds", C.c_double), ("offset", C.c_double), ("vehicle_length", C.c_double), ("vehicle_width", C.c_double),
    ]


class ScenarioStruct(C.Structure):
    _fields_ = [
        ("n_vehicles", C.c_int32),
        ("x_start", abi.c_double_p), ("y_start", abi.c_double_p), ("yaw_start", abi.c_double_p), ("reference_speed", abi.c_double_p),
        ("path_offset", abi.c_int32_p), ("path_x", abi.c_double_p), ("path_y", abi.c_double_p),
        ("lanelets_offset", abi.c_int32_p), ("lanelets_index", abi.c_int32_p), ("points_index", abi.c_int32_p), ("is_loop", abi.c_int32_p),
        ("tile_dx", abi.c_double_p), ("tile_dy", abi.c_double_p),
        ("n_lanelets", C.c_int32), ("left_offset", abi.c_int32_p), ("right_offset", abi.c_int32_p),
        ("left_x", abi.c_double_p), ("left_y", abi.c_double_p), ("right_x", abi.c_double_p), ("right_y", abi.c_double_p),
        ("obstacles", abi.PolygonSet),
        ("n_trims", C.c_int32), ("trim_speed", abi.c_double_p), ("trim_steering", abi.c_double_p),
    ]


def _declare(L):
    H = C.c_void_p
    if getattr(L, "_controller_declared", False):
        return L
    L.pdmpc_controller_create.argtypes = [H, C.POINTER(ControllerConfig), C.POINTER(ScenarioStruct), C.POINTER(H)]
    L.pdmpc_controller_destroy.argtypes = [H]
    L.pdmpc_controller_step.argtypes = [H]
    L.pdmpc_controller_run.argtypes = [H, C.c_int32, abi.c_double_p]
    L.pdmpc_controller_run.restype = C.c_int
    L.pdmpc_controller_build_step.argtypes = [H]
    L.pdmpc_controller_apply.argtypes = [H, C.POINTER(abi.VehicleOut)]
    L.pdmpc_controller_problem.argtypes = [H, C.POINTER(C.c_int32), C.POINTER(C.POINTER(abi.VehicleIn)), C.POINTER(abi.c_int32_p), C.POINTER(abi.c_int32_p),
                                           C.POINTER(C.POINTER(abi.PolygonSet)), C.POINTER(abi.c_int32_p), C.POINTER(abi.c_int32_p)]
    L.pdmpc_controller_state.argtypes = [H] + [abi.c_double_p] * 5 + [abi.c_int32_p, C.POINTER(C.c_int32)]
    L.pdmpc_controller_records.argtypes = [H]
    L.pdmpc_controller_records.restype = C.POINTER(abi.VehicleOut)
    L.pdmpc_controller_last_error.restype = C.c_char_p
    L.pdmpc_controller_explore_build.argtypes = [H, C.c_int32, C.c_uint32]
    L.pdmpc_controller_explore_problem.argtypes = [H, C.POINTER(C.c_int32), C.POINTER(C.POINTER(abi.VehicleIn)), C.POINTER(abi.c_int32_p), C.POINTER(abi.c_int32_p),
                                                   C.POINTER(C.POINTER(abi.PolygonSet)), C.POINTER(abi.c_int32_p), C.POINTER(abi.c_int32_p), C.POINTER(abi.c_int32_p)]
    L.pdmpc_controller_explore_choose.argtypes = [H, C.POINTER(abi.VehicleOut), abi.c_int32_p, C.POINTER(C.c_int32), abi.c_double_p]
    L.pdmpc_controller_explore_step.argtypes = [H, C.c_int32]
    L.pdmpc_controller_explore_run.argtypes = [H, C.c_int32, C.c_int32, abi.c_double_p]
    L.pdmpc_controller_explore_result.argtypes = [H, abi.c_int32_p, C.POINTER(C.c_int32), C.POINTER(abi.c_double_p), C.POINTER(C.POINTER(abi.VehicleOut))]
    for name in ("pdmpc_controller_explore_build", "pdmpc_controller_explore_problem", "pdmpc_controller_explore_choose", "pdmpc_controller_explore_step",
                 "pdmpc_controller_explore_run", "pdmpc_controller_explore_result"):
        getattr(L, name).restype = C.c_int
    for name in ("pdmpc_controller_create", "pdmpc_controller_destroy", "pdmpc_controller_step", "pdmpc_controller_build_step", "pdmpc_controller_apply",
                 "pdmpc_controller_problem", "pdmpc_controller_state"):
        getattr(L, name).restype = C.c_int
    L._controller_declared = True
    return L


def _flat(arrays, dtype):
    off = np.zeros(len(arrays) + 1, dtype=np.int32)
    for i, a in enumerate(arrays):
        off[i + 1] = off[i] + len(a)
    data = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=dtype).ravel() for a in arrays] + [np.zeros(1, dtype=dtype)]))
    return off, data


def _polys(ps):
    """PolygonSet -> list of (2, V) arrays."""
    out = []
    for p in range(ps.n_polygons):
        a, b = ps.offset[p], ps.offset[p + 1]
        out.append(np.array([[ps.x[q] for q in range(a, b)], [ps.y[q] for q in range(a, b)]]).reshape(2, b - a))
    return out


class NativeController:
    def __init__(self, options, scenario, mpa, handle=None, coupling="full", priority_strategy="constant", weight_strategy="distance"):
        if scenario.dynamic_obstacle_area:
            raise ValueError("the native controller takes static scenario obstacles only")
        self.L = _declare(load_library())
        self.options, self.mpa, self.n, self.Hp = options, mpa, options.amount, options.Hp
        veh = scenario.vehicles
        keep = []

        def d(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            keep.append(a)
            return a.ctypes.data_as(abi.c_double_p)

        def i32(a):
            a = np.ascontiguousarray(a, dtype=np.int32)
            keep.append(a)
            return a.ctypes.data_as(abi.c_int32_p)

        s = ScenarioStruct()
        s.n_vehicles = self.n
        s.x_start, s.y_start, s.yaw_start = d([v.x_start for v in veh]), d([v.y_start for v in veh]), d([v.yaw_start for v in veh])
        s.reference_speed = d([v.reference_speed for v in veh])
        po, px = _flat([v.reference_path[:, 0] for v in veh], np.float64)
        _, py = _flat([v.reference_path[:, 1] for v in veh], np.float64)
        s.path_offset, s.path_x, s.path_y = i32(po), d(px), d(py)
        if veh[0].lanelets_index is not None:
            lo, li = _flat([v.lanelets_index for v in veh], np.int32)
            _, pi = _flat([v.points_index for v in veh], np.int32)
            s.lanelets_offset, s.lanelets_index, s.points_index = i32(lo), i32(li), i32(pi)
            s.is_loop = i32([1 if v.is_loop else 0 for v in veh])
            off = getattr(scenario, "tile_offset", [(0.0, 0.0)] * self.n)
            s.tile_dx, s.tile_dy = d([o[0] for o in off]), d([o[1] for o in off])
            bl = scenario.lanelet_boundary
            s.n_lanelets = len(bl)
            lo_, lx = _flat([b[0][:, 0] for b in bl], np.float64)
            _, ly = _flat([b[0][:, 1] for b in bl], np.float64)
            ro_, rx = _flat([b[1][:, 0] for b in bl], np.float64)
            _, ry = _flat([b[1][:, 1] for b in bl], np.float64)
            s.left_offset, s.right_offset = i32(lo_), i32(ro_)
            s.left_x, s.left_y, s.right_x, s.right_y = d(lx), d(ly), d(rx), d(ry)
        k2 = abi._Keep()
        s.obstacles = abi.pack_polygon_set(list(scenario.obstacles), k2)
        keep.append(k2)
        s.n_trims = len(mpa.trims)
        s.trim_speed, s.trim_steering = d([t.speed for t in mpa.trims]), d([t.steering for t in mpa.trims])
        cfg = ControllerConfig(
            Hp=options.Hp, coupling=COUPLING[coupling], priority_strategy=PRIORITY[priority_strategy], weight_strategy=WEIGHT[weight_strategy],
            max_num_CLs=options.max_num_CLs, constraint_from_successor=SUCCESSOR[options.constraint_from_successor], dt_seconds=options.dt_seconds,
            offset=options.offset, vehicle_length=veh[0].Length, vehicle_width=veh[0].Width,
        )
        self.handle = handle  # keeps the backend handle alive as long as the controller that drives it
        self.c = C.c_void_p()
        rc = self.L.pdmpc_controller_create(handle.h if handle is not None else None, C.byref(cfg), C.byref(s), C.byref(self.c))
        self._check(rc, "pdmpc_controller_create")
        del keep

    def _check(self, rc, what):
        if rc != 0:
            msg = self.L.pdmpc_controller_last_error()
            raise BackendError("%s failed with status %d: %s" % (what, rc, msg.decode() if msg else ""))

    def close(self):
        if self.c:
            self.L.pdmpc_controller_destroy(self.c)
            self.c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self):
        """One whole MPC time step natively (build, one launch, apply); returns the records in slot order."""
        self._check(self.L.pdmpc_controller_step(self.c), "pdmpc_controller_step")
        p = self.L.pdmpc_controller_records(self.c)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.n * abi.VEHICLE_OUT_DTYPE.itemsize,)).view(abi.VEHICLE_OUT_DTYPE).copy()

    def run(self, n_steps):
        """n_steps closed-loop steps in one native call -> wall-clock milliseconds of every step."""
        ms = np.zeros(max(n_steps, 1))
        self._check(self.L.pdmpc_controller_run(self.c, n_steps, ms.ctypes.data_as(abi.c_double_p)), "pdmpc_controller_run")
        return ms[:n_steps]

    def last_timing(self):
        """Host milliseconds of the last step by part (pdmpc_controller_last_timing)."""
        t = (C.c_double * 6)()
        self.L.pdmpc_controller_last_timing.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self.L.pdmpc_controller_last_timing(self.c, t), "pdmpc_controller_last_timing")
        return dict(zip(("build", "pack", "enqueue", "wait_and_read_back", "choose", "apply"), (float(x) for x in t)))

    def timing_mean(self, reset=True):
        """Mean host milliseconds per step by part over the steps since the last reset (pdmpc_controller_timing_sum)."""
        t = (C.c_double * 6)()
        k = C.c_int64(0)
        self.L.pdmpc_controller_timing_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        self._check(self.L.pdmpc_controller_timing_sum(self.c, t, C.byref(k), 1 if reset else 0), "pdmpc_controller_timing_sum")
        return dict(zip(("build", "pack", "enqueue", "wait_and_read_back", "choose", "apply"), (float(x) / max(k.value, 1) for x in t)))

    def build_step(self):
        self._check(self.L.pdmpc_controller_build_step(self.c), "pdmpc_controller_build_step")

    def apply(self, records):
        recs = np.ascontiguousarray(records)
        self._check(self.L.pdmpc_controller_apply(self.c, abi.out_ptr(recs)), "pdmpc_controller_apply")

    def problem(self):
        """The last built step problem decoded into the dict form of controller.build_step_problem (for tests)."""
        n = C.c_int32()
        vin = C.POINTER(abi.VehicleIn)()
        po, pi, order, levels = abi.c_int32_p(), abi.c_int32_p(), abi.c_int32_p(), abi.c_int32_p()
        fb = C.POINTER(abi.PolygonSet)()
        self._check(self.L.pdmpc_controller_problem(self.c, C.byref(n), C.byref(vin), C.byref(po), C.byref(pi), C.byref(fb), C.byref(order), C.byref(levels)), "pdmpc_controller_problem")
        iters, preds, fallback = self._decode(n.value, vin, po, pi, fb)
        order_l = [int(order[s]) for s in range(n.value)]
        lv = [int(levels[v]) for v in range(n.value)]
        level_sizes = [sum(1 for x in lv if x == l) for l in range(1, max(lv) + 1)]
        return {"order": order_l, "iters": iters, "preds": preds, "fallback": fallback, "level_sizes": level_sizes, "levels": [lv[v] for v in order_l]}

    def _decode(self, n, vin, po, pi, fb):
        from .iteration_data import VehicleIter

        Hp = self.Hp
        iters, preds, fallback = [], [], []
        for s in range(n):
            v = vin[s]
            dyn = _polys(v.dynamic_obstacles)
            left = np.array([[v.left_x[q] for q in range(v.n_left)], [v.left_y[q] for q in range(v.n_left)]]) if v.n_left else None
            right = np.array([[v.right_x[q] for q in range(v.n_right)], [v.right_y[q] for q in range(v.n_right)]]) if v.n_right else None
            iters.append(VehicleIter(
                x0=np.array([v.x0, v.y0, v.yaw0, 0.0]), trim_index=int(v.trim0),
                reference_trajectory_points=np.array([[v.ref_x[q], v.ref_y[q]] for q in range(Hp)]), v_ref=np.array([v.v_ref[q] for q in range(Hp)]),
                predicted_lanelet_boundary=(left, right), obstacles=_polys(v.obstacles),
                dynamic_obstacle_area=[dyn[r * Hp : (r + 1) * Hp] for r in range(len(dyn) // Hp)],
            ))
            preds.append([int(pi[q]) for q in range(po[s], po[s + 1])])
            f = _polys(fb[s])
            fallback.append(f if f else None)
        return iters, preds, fallback

    # ---- the explorative step: twin of pdmpc.explorative
    def explore_build(self, n_perm, seed):
        self._check(self.L.pdmpc_controller_explore_build(self.c, n_perm, int(seed)), "pdmpc_controller_explore_build")
        self.n_perm = n_perm

    def explore_problem(self):
        """The flattened batch of the last explore_build in the dict form of explorative.build_exploration_batch (for tests)."""
        n = C.c_int32()
        vin = C.POINTER(abi.VehicleIn)()
        po, pi, inst, veh, lvl = (abi.c_int32_p() for _ in range(5))
        fb = C.POINTER(abi.PolygonSet)()
        self._check(self.L.pdmpc_controller_explore_problem(self.c, C.byref(n), C.byref(vin), C.byref(po), C.byref(pi), C.byref(fb), C.byref(inst), C.byref(veh), C.byref(lvl)),
                    "pdmpc_controller_explore_problem")
        iters, preds, fallback = self._decode(n.value, vin, po, pi, fb)
        N = n.value
        levels = [int(lvl[s]) for s in range(N)]
        vehicles = [int(veh[s]) for s in range(N)]
        return {"order": vehicles, "iters": iters, "preds": preds, "fallback": fallback, "levels": levels, "instance": [int(inst[s]) for s in range(N)], "vehicle": vehicles,
                "level_sizes": [sum(1 for x in levels if x == l) for l in range(1, max(levels) + 1)], "n_instances": self.n_perm}

    def explore_choose(self, records):
        """-> (instance chosen per vehicle, cost table n_perm x n_graphs)."""
        recs = np.ascontiguousarray(records)
        chosen = np.zeros(self.n, dtype=np.int32)
        g = C.c_int32()
        cost = np.zeros(self.n_perm * self.n)
        self._check(self.L.pdmpc_controller_explore_choose(self.c, abi.out_ptr(recs), chosen.ctypes.data_as(abi.c_int32_p), C.byref(g), cost.ctypes.data_as(abi.c_double_p)),
                    "pdmpc_controller_explore_choose")
        return chosen, cost[: self.n_perm * g.value].reshape(self.n_perm, g.value)

    def explore_step(self, n_perm):
        """One explorative time step natively (batch, ONE launch, choice, apply) -> (records of the batch, chosen instance per vehicle)."""
        self._check(self.L.pdmpc_controller_explore_step(self.c, n_perm), "pdmpc_controller_explore_step")
        chosen = np.zeros(self.n, dtype=np.int32)
        p = C.POINTER(abi.VehicleOut)()
        self._check(self.L.pdmpc_controller_explore_result(self.c, chosen.ctypes.data_as(abi.c_int32_p), None, None, C.byref(p)), "pdmpc_controller_explore_result")
        recs = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.n * n_perm * abi.VEHICLE_OUT_DTYPE.itemsize,)).view(abi.VEHICLE_OUT_DTYPE).copy()
        return recs, chosen

    def explore_follow_own(self, on=True):
        self.L.pdmpc_controller_explore_follow_own.argtypes = [C.c_void_p, C.c_int32]
        self._check(self.L.pdmpc_controller_explore_follow_own(self.c, 1 if on else 0), "pdmpc_controller_explore_follow_own")

    def explore_run(self, n_perm, n_steps):
        ms = np.zeros(max(n_steps, 1))
        self._check(self.L.pdmpc_controller_explore_run(self.c, n_perm, n_steps, ms.ctypes.data_as(abi.c_double_p)), "pdmpc_controller_explore_run")
        return ms[:n_steps]

    def state(self):
        n = self.n
        arr = [np.zeros(n) for _ in range(5)]
        nf = np.zeros(n, dtype=np.int32)
        k = C.c_int32()
        self._check(self.L.pdmpc_controller_state(self.c, *[a.ctypes.data_as(abi.c_double_p) for a in arr], nf.ctypes.data_as(abi.c_int32_p), C.byref(k)), "pdmpc_controller_state")
        return {"x": arr[0], "y": arr[1], "yaw": arr[2], "speed": arr[3], "steering": arr[4], "needs_fallback": nf != 0, "k": k.value}
