"""The native step controller (csrc/step_controller.cpp) against its Python twin (pdmpc.controller): the step problems the
two build from the same plant state are bit-identical, and so are the closed loops they drive (planner = the oracle, no GPU).
Covers distance coupling with constant priorities (C2), colouring priorities with the coupling DAG cut to two levels (C3),
uncut colouring (C4) and full coupling on the circle (C1), incl. exhaustion with standstill and with fallbacks."""
import numpy as np
import pytest

from pdmpc.config import Config, ScenarioType
from pdmpc.controller import PrioritizedSequentialController
from pdmpc.iteration_data import info_from_record
from pdmpc.mpa import get_mpa
from pdmpc.native_controller import NativeController


def same_poly_list(a, b):
    return len(a) == len(b) and all(np.array_equal(np.asarray(x).view(np.uint64), np.asarray(y).view(np.uint64)) for x, y in zip(a, b))


def assert_same_problem(p, q, ctx):
    assert p["order"] == q["order"], ctx
    assert p["preds"] == q["preds"], ctx
    assert p["levels"] == q["levels"] and p["level_sizes"] == q["level_sizes"], ctx
    for s, (a, b) in enumerate(zip(p["iters"], q["iters"])):
        where = "%s slot %d" % (ctx, s)
        assert np.array_equal(a.x0[:3].view(np.uint64), b.x0[:3].view(np.uint64)), where
        assert a.trim_index == b.trim_index, where
        assert np.array_equal(np.asarray(a.reference_trajectory_points).view(np.uint64), np.asarray(b.reference_trajectory_points).view(np.uint64)), where
        assert np.array_equal(np.asarray(a.v_ref, dtype=np.float64).view(np.uint64), np.asarray(b.v_ref, dtype=np.float64).view(np.uint64)), where
        for side in (0, 1):
            x, y = a.predicted_lanelet_boundary[side], b.predicted_lanelet_boundary[side]
            assert (x is None or np.size(x) == 0) == (y is None or np.size(y) == 0), where
            if x is not None and np.size(x):
                assert np.array_equal(np.asarray(x).view(np.uint64), np.asarray(y).view(np.uint64)), where
        assert same_poly_list(a.obstacles, b.obstacles), where
        assert len(a.dynamic_obstacle_area) == len(b.dynamic_obstacle_area), where
        for ra, rb in zip(a.dynamic_obstacle_area, b.dynamic_obstacle_area):
            assert same_poly_list(ra, rb), where
        fa, fb = p["fallback"][s], q["fallback"][s]
        assert (fa is None or len(fa) == 0) == (fb is None or len(fb) == 0), where
        if fa is not None and len(fa):
            assert same_poly_list(fa, fb), where


def run_both(options, scenario, n_steps, coupling, boundary=None, force_exhaustion=None, **kw):
    from oracle import oracle

    mpa = get_mpa(options)
    py = PrioritizedSequentialController(options, scenario, mpa, None, coupling=coupling, boundary_provider=boundary, **kw)
    nat = NativeController(options, scenario, mpa, None, coupling=coupling, **kw)
    for k in range(n_steps):
        nat.build_step()
        q = nat.problem()

        def plan_step(prob):
            assert_same_problem(prob, q, "step %d" % (k + 1))
            recs, _ = oracle.plan_step(options, mpa, prob)
            if force_exhaustion is not None and force_exhaustion(k + 1) is not None:
                s = prob["order"].index(force_exhaustion(k + 1))
                recs[s]["status"] = 1  # this vehicle's search "ran empty": fallback (moving) or standstill handling
            nat.apply(recs)
            return [info_from_record(recs[i], options.Hp) for i in range(len(recs))]

        py.step(plan_step=plan_step)
        st = nat.state()
        assert np.array_equal(st["x"], np.array([m.x for m in py.meas])) and np.array_equal(st["yaw"], np.array([m.yaw for m in py.meas])), k
        assert np.array_equal(st["speed"], np.array([m.speed for m in py.meas])) and np.array_equal(st["steering"], np.array([m.steering for m in py.meas])), k
        assert st["needs_fallback"].tolist() == [bool(i.needs_fallback) for i in py.infos], k
    nat.close()
    return py


def test_c2_distance_coupling_constant_priorities():
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=6, max_nodes=1 << 20)
    sc = commonroad_scenario(options, seed=1)
    run_both(options, sc, 8, "distance", boundary_provider(sc))


@pytest.mark.parametrize("max_levels,weight", [(2, "distance"), (3, "constant"), (99, "distance")])
def test_colouring_priorities_with_cut_couplings(max_levels, weight):
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=40, Hp=5, max_num_CLs=max_levels, max_nodes=1 << 20)
    sc = commonroad_scenario(options, seed=2, tiles=2)
    py = run_both(options, sc, 6, "distance", boundary_provider(sc), priority_strategy="coloring", weight_strategy=weight)
    assert int(py.last_levels.max()) <= max_levels


def test_circle_full_coupling_with_exhaustion_and_fallbacks():
    from pdmpc.scenario import circle_scenario

    options = Config(scenario_type=ScenarioType.circle, amount=4, Hp=5, max_nodes=1 << 20)
    # step 1: vehicle 3 (standing) exhausts -> standstill handling; step 4: vehicle 2 (moving) exhausts -> its fallback spreads
    run_both(options, circle_scenario(options), 6, "full", force_exhaustion=lambda k: {1: 2, 4: 1}.get(k))


@pytest.mark.parametrize("strategy,max_levels", [("constant", 99), ("coloring", 99), ("coloring", 3)])
def test_explorative_step_native_twin(strategy, max_levels):
    """the native explorative step (pdmpc_controller_explore_*) against pdmpc.explorative — the flattened batch of
    the step's prioritizations bit for bit, the choice per sub-graph and its cost table, and the closed loop that goes on with the
    chosen plans and couplings (planner = the oracle, no GPU)."""
    from oracle import oracle
    from pdmpc.explorative import choose_solution, explore_step
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    # (under the colouring strategy instance 0 is the coloured prioritization and the other instances permute ITS computation levels;
    # with max_num_CLs = 3 the sub-graphs of the cost choice are those of the cut, sequential coupling)
    options = Config(scenario_type=ScenarioType.commonroad, amount=14, Hp=5, max_num_CLs=max_levels, max_nodes=1 << 30)
    sc = commonroad_scenario(options, seed=5)
    mpa = get_mpa(options)
    K = 5
    py = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc), priority_strategy=strategy)
    nat = NativeController(options, sc, mpa, None, coupling="distance", priority_strategy=strategy)
    differing = 0
    for k in range(8):
        nat.explore_build(K, seed=k + 1)
        q = nat.explore_problem()
        base_order = nat.problem()["order"]

        def plan_batch(batch):
            assert_same_problem(batch, q, "explorative step %d" % (k + 1))
            assert batch["instance"] == q["instance"] and batch["vehicle"] == q["vehicle"]
            recs, _ = oracle.plan_step(options, mpa, batch)
            chosen_nat, cost_nat = nat.explore_choose(recs)
            want, cost = choose_solution(batch, recs, options.Hp)
            assert np.array_equal(cost_nat, cost)
            slot = {(p, v): s for s, (p, v) in enumerate(zip(batch["instance"], batch["vehicle"]))}
            nat.apply(recs[[slot[(int(chosen_nat[v]), v)] for v in base_order]])
            plan_batch.chosen_nat = chosen_nat
            return recs

        _, _, chosen = explore_step(py, plan_batch, K)
        assert chosen == plan_batch.chosen_nat.tolist()
        differing += sum(1 for c in chosen if c != 0)
        st = nat.state()
        assert np.array_equal(st["x"], np.array([m.x for m in py.meas])) and np.array_equal(st["y"], np.array([m.y for m in py.meas])), k
        assert np.array_equal(st["yaw"], np.array([m.yaw for m in py.meas])) and np.array_equal(st["speed"], np.array([m.speed for m in py.meas])), k
        assert st["needs_fallback"].tolist() == [bool(i.needs_fallback) for i in py.infos], k
    assert differing > 0, "the exploration never preferred another prioritization: the test would not notice a wrong choice"
    nat.close()


def test_hundred_vehicles_colouring_cut_to_two_levels():
    """A scenario large enough that the controller's sparse paths matter (coupling lists, the colouring's block maxima, the plans'
    reused storage): 100 vehicles on five tiles — not a multiple of the 8-entry / 32-vertex strides those paths step by."""
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=100, Hp=6, max_num_CLs=2, max_nodes=1 << 20)
    sc = commonroad_scenario(options, seed=3, tiles=5)
    py = run_both(options, sc, 4, "distance", boundary_provider(sc), priority_strategy="coloring", weight_strategy="distance")
    assert int(py.last_levels.max()) <= 2


@pytest.mark.parametrize("mode", ["none", "area_of_previous_trajectory"])
def test_other_constraints_from_successors(mode):
    """Config.m:37 constraint_from_successor: successors contribute nothing (none) or their previous plan shifted by one step
    (area_of_previous_trajectory, PrioritizedController.m:541-553) instead of their standstill rectangle — the other branches of the
    per-slot contributor loops, and of what an exhausted vehicle publishes."""
    from pdmpc.config import ConstraintFromSuccessor
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=30, Hp=5, max_num_CLs=2, max_nodes=1 << 20,
                     constraint_from_successor=ConstraintFromSuccessor[mode])
    sc = commonroad_scenario(options, seed=4, tiles=2)
    run_both(options, sc, 6, "distance", boundary_provider(sc), priority_strategy="coloring", weight_strategy="distance")
