"""The MATLAB-shaped entry points (include/pdmpc_matlab.h, csrc/matlab_marshal.cpp) on the CPU: column-major matrices, cells
in linear order, the n x n x Hp transition matrix and the n x n coupling matrix are turned into exactly the step problem the
Python controller builds — and the oracle plans the same records from either.  No GPU calls."""
import ctypes as C
import os
import re

import numpy as np

from pdmpc import abi
from pdmpc.config import Config, MpaType, ScenarioType
from pdmpc.controller import PrioritizedSequentialController
from pdmpc.iteration_data import info_from_record
from pdmpc.mpa import get_mpa

import matlab_shapes as ms
from test_native_controller import same_poly_list

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _polys(ps):
    out = []
    for p in range(ps.n_polygons):
        a, b = ps.offset[p], ps.offset[p + 1]
        out.append(np.array([[ps.x[q] for q in range(a, b)], [ps.y[q] for q in range(a, b)]]))
    return out


def test_library_exports_every_symbol_of_the_matlab_header():
    text = open(os.path.join(ROOT, "include", "pdmpc_matlab.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(pdmpc_ml_[a-z_]+)\s*\(", text)))
    assert len(names) >= 10
    L = ms.lib()
    for name in names:
        assert hasattr(L, name), name


def test_mpa_tables_from_matlab_shapes_equal_the_python_marshalling():
    for mpa_type in (MpaType.single_speed, MpaType.triple_speed):
        options = Config(scenario_type=ScenarioType.commonroad, Hp=6, mpa_type=mpa_type)
        mpa = get_mpa(options)
        keep = ms.Keep()
        T, n, Hp, man = ms.ml_mpa_args(mpa, keep)
        h = C.c_void_p()
        assert ms.lib().pdmpc_ml_mpa_create(T, n, Hp, man, C.byref(h)) == 0
        got = ms.lib().pdmpc_ml_mpa_view(h).contents
        want, keep2 = abi.pack_mpa(mpa)
        assert (got.n_trims, got.Hp, got.n_maneuvers) == (want.n_trims, want.Hp, want.n_maneuvers)
        cnt = Hp * n * n
        assert bytes(C.cast(got.transition, C.POINTER(C.c_uint8 * cnt)).contents) == bytes(C.cast(want.transition, C.POINTER(C.c_uint8 * cnt)).contents)
        assert [got.maneuver_index[q] for q in range(n * n)] == [want.maneuver_index[q] for q in range(n * n)]
        nb = C.sizeof(abi.Maneuver) * got.n_maneuvers
        assert C.string_at(C.addressof(got.maneuvers.contents), nb) == C.string_at(C.addressof(want.maneuvers.contents), nb)
        ms.lib().pdmpc_ml_mpa_destroy(h)
        del keep2


def _closed_loop(options, scenario, boundary, n_steps, check, **kw):
    from oracle import oracle

    mpa = get_mpa(options)
    ctl = PrioritizedSequentialController(options, scenario, mpa, None, coupling="distance", boundary_provider=boundary, **kw)

    def plan_step(prob):
        ref, _ = oracle.plan_step(options, mpa, prob)
        check(prob, ref, mpa)
        return [info_from_record(ref[i], options.Hp) for i in range(len(ref))]

    for _ in range(n_steps):
        ctl.step(plan_step=plan_step)


def test_step_problem_from_matlab_shapes_is_the_controllers_problem():
    """Per-vehicle iters + n x n coupling matrix + n x Hp fallback cell (vehicle order, MATLAB layout) -> the same slots, levels,
    predecessor lists and vehicle inputs as controller.build_step_problem; the oracle plans identical records from the
    marshalled structs (read through the C ABI, no Python marshalling in between)."""
    from oracle import oracle, packing
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=14, Hp=6, max_nodes=1 << 30)
    sc = commonroad_scenario(options, seed=3)
    seen = []

    def check(prob, ref, mpa):
        Hp = options.Hp
        keep = ms.Keep()
        step = ms.step_create(ms.vehicle_order_problem(prob), Hp, keep)
        n, vin, po, pi, fb, order, levels = ms.step_problem(step)
        assert n == len(prob["iters"])
        assert [order[s] - 1 for s in range(n)] == prob["order"]
        assert [levels[prob["order"][s]] for s in range(n)] == prob["levels"]
        for s in range(n):
            assert sorted(pi[q] for q in range(po[s], po[s + 1])) == sorted(prob["preds"][s])
            it, v = prob["iters"][s], vin[s]
            assert (v.x0, v.y0, v.yaw0, v.trim0) == (float(it.x0[0]), float(it.x0[1]), float(it.x0[2]), int(it.trim_index))
            ref_pts = np.asarray(it.reference_trajectory_points)
            assert [v.ref_x[k] for k in range(Hp)] == list(ref_pts[:, 0]) and [v.ref_y[k] for k in range(Hp)] == list(ref_pts[:, 1])
            assert [v.v_ref[k] for k in range(Hp)] == list(np.asarray(it.v_ref, dtype=np.float64))
            for side, (cnt, xs, ys) in enumerate(((v.n_left, v.left_x, v.left_y), (v.n_right, v.right_x, v.right_y))):
                b = it.predicted_lanelet_boundary[side]
                assert cnt == (0 if b is None or np.size(b) == 0 else np.asarray(b).shape[1])
                if cnt:
                    assert np.array_equal(np.array([[xs[q] for q in range(cnt)], [ys[q] for q in range(cnt)]]), np.asarray(b))
            assert same_poly_list(_polys(v.obstacles), it.obstacles)
            dyn = _polys(v.dynamic_obstacles)
            assert len(dyn) == len(it.dynamic_obstacle_area) * Hp
            for r, row in enumerate(it.dynamic_obstacle_area):
                assert same_poly_list(dyn[r * Hp : (r + 1) * Hp], row)  # polygon index i * Hp + (k - 1)
            f = prob["fallback"][s]
            assert same_poly_list(_polys(fb[s]), list(f) if f is not None and len(f) else [])
        # the oracle on the marshalled structs themselves
        mpa_struct, keep_m = packing.pack_mpa(mpa)
        recs = packing.out_array(n)
        lv = np.asarray(prob["level_sizes"], dtype=np.int32)
        ms_, thr = C.c_double(), C.c_double()
        cfg = oracle.make_abi_config(options)
        rc = oracle.lib().oracle_plan_step(C.byref(cfg), C.byref(mpa_struct), n, C.cast(vin, C.c_void_p), po, pi, C.cast(fb, C.c_void_p), len(lv),
                                           lv.ctypes.data_as(packing._iptr), recs.ctypes.data_as(C.c_void_p), 1, C.byref(ms_), C.byref(thr))
        assert rc == 0
        assert recs[:n].tobytes() == ref.tobytes()
        ms.lib().pdmpc_ml_step_destroy(step)
        seen.append(n)
        del keep_m

    _closed_loop(options, sc, boundary_provider(sc), 5, check)
    assert len(seen) == 5


def test_hdv_cells_and_reversed_priorities_keep_their_order():
    """n_h x Hp and n_d x Hp cells with distinct polygons per (row, step): the linear cell order i + k * R must come out as
    polygon i * Hp + k; a coupling matrix whose edges run from high to low vehicle indices still yields level-ordered slots."""
    import problems

    options, mpa, iters = problems.problem_set("interx", 7, 6, Hp=5, n_hdv=2)
    Hp = options.Hp
    n = len(iters)
    seq = np.zeros((n, n))
    for i in range(n - 1):
        seq[i + 1, i] = 1.0  # vehicle i+1 plans before vehicle i
    keep = ms.Keep()
    step = ms.step_create((iters, seq, [None] * n), Hp, keep)
    m, vin, po, pi, fb, order, levels = ms.step_problem(step)
    assert [order[s] for s in range(n)] == list(range(n, 0, -1))
    assert [levels[v] for v in range(n)] == list(range(n, 0, -1))
    for s in range(n):
        v = order[s] - 1
        assert [pi[q] for q in range(po[s], po[s + 1])] == ([s - 1] if s > 0 else [])
        hdv = _polys(vin[s].hdv_reachable_sets)
        for r, row in enumerate(iters[v].hdv_reachable_sets):
            assert same_poly_list(hdv[r * Hp : (r + 1) * Hp], row)
        assert fb[s].n_polygons == 0
    ms.lib().pdmpc_ml_step_destroy(step)


def test_record_arrays_have_matlab_layout():
    Hp = 6
    rec = np.zeros(1, dtype=abi.VEHICLE_OUT_DTYPE)
    rng = np.random.default_rng(0)
    rec["predicted_trims"][0, :Hp] = rng.integers(1, 13, Hp)
    rec["shape_cols"][0, :Hp] = rng.integers(5, 8, Hp)
    rec["tree_path"][0, : Hp + 1] = rng.integers(1, 999, Hp + 1)
    rec["y_predicted"][0] = rng.normal(size=rec["y_predicted"][0].shape)
    rec["shapes"][0] = rng.normal(size=rec["shapes"][0].shape)
    rec["path_nodes"][0] = rng.normal(size=rec["path_nodes"][0].shape)
    out = ms.record_arrays(rec[0], Hp)
    assert np.array_equal(out["predicted_trims"][0], rec["predicted_trims"][0, :Hp])
    assert np.array_equal(out["shape_cols"][0], rec["shape_cols"][0, :Hp])
    assert np.array_equal(out["tree_path"][0], rec["tree_path"][0, : Hp + 1])
    assert np.array_equal(out["y_predicted"], rec["y_predicted"][0, :Hp])          # y(k, c)
    assert np.array_equal(out["shapes"], rec["shapes"][0, :Hp])                    # shapes(k, row, v)
    assert np.array_equal(out["path_nodes"], rec["path_nodes"][0, : Hp + 1])       # nodes(k, c)
