"""Whole-step planning in one launch (device-side hand-off of solved areas) vs the oracle's host level loop."""
import copy
import os

import numpy as np
import pytest

from pdmpc.config import Config, MpaType, ScenarioType
from pdmpc.controller import PrioritizedSequentialController
from pdmpc.iteration_data import info_from_record
from pdmpc.mpa import get_mpa

from test_gpu_parity import assert_records_equal

pytestmark = pytest.mark.gpu


def run_closed_loop(options, scenario, coupling, boundary, n_steps, oracle_threads=1, **controller_kw):
    from oracle import oracle
    from pdmpc.optimizer import GraphSearchHip

    mpa = get_mpa(options)
    opt = GraphSearchHip(options)
    opt._ensure_mpa(mpa)
    ctl = PrioritizedSequentialController(options, scenario, mpa, None, coupling=coupling, boundary_provider=boundary, **controller_kw)
    n_checked = [0]

    # the oracle plans with the reference's unbounded tree (Tree.m:54-70); the backend's arenas grow on demand (pdmpc_plan_step)
    unbounded = copy.copy(options)
    unbounded.max_nodes = 1 << 30

    def plan_step(prob):
        n = len(prob["iters"])
        fb = [f if f is not None else [] for f in prob["fallback"]]
        gpu = opt.handle.plan_step(prob["iters"], prob["preds"], fb)
        ref, _ = oracle.plan_step(unbounded, mpa, prob, n_threads=oracle_threads)
        assert (ref["status"] != 2).all()
        assert_records_equal(gpu, ref, "step %d" % n_checked[0])
        n_checked[0] += 1
        return [info_from_record(gpu[i], options.Hp) for i in range(n)]

    for _ in range(n_steps):
        ctl.step(plan_step=plan_step)
    # pdmpc_plan_step plans a call again in resident slices when a search timed out waiting for a predecessor: the safety net must
    # not be what makes these loops pass (a kernel that loses a publication now and then would hide behind it)
    stats = opt.handle.stats()
    assert stats["safe_replans"] == 0
    opt.handle.close()
    ctl.handle_stats = stats
    return ctl


def test_circle_step_in_one_launch():
    from pdmpc.scenario import circle_scenario

    options = Config(scenario_type=ScenarioType.circle, amount=3, Hp=5, T_end=4, max_vehicles=4, max_nodes=1 << 15)
    run_closed_loop(options, circle_scenario(options), "full", None, options.k_end)


def test_circle_8_vehicles_step():
    from pdmpc.scenario import circle_scenario

    options = Config(scenario_type=ScenarioType.circle, amount=8, Hp=6, max_vehicles=8, max_nodes=1 << 16)
    ctl = run_closed_loop(options, circle_scenario(options), "full", None, 12)
    # equal keys are structural here (vehicle 1 drives along the x axis: +dy and -dy cancel exactly): the product kernel, with the
    # separating-axis check items, and its replay through the binary heap for the tied searches -- all in the step's one launch
    assert ctl.handle_stats["kernel"] == 2 and ctl.handle_stats["queue_fallbacks"] > 0


def test_road_network_20_vehicles_step():
    """BASELINE config 1 (C2): 20 vehicles on the lab map, Hp 8, InterX."""
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=8, max_vehicles=32, max_nodes=1 << 17)
    sc = commonroad_scenario(options, seed=1)
    run_closed_loop(options, sc, "distance", boundary_provider(sc), 12)


def test_benchmarked_window_c2_steps_1_to_40():
    """The steps bench.py measures by default: seed 1, closed-loop steps 21-40 (the first 20 are dropped as in eval_phd.m:41-49),
    incl. the 10 k-pop searches that decide the headline's heavy steps.  Every step of the closed loop against the oracle."""
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=8, max_vehicles=32, max_nodes=1 << 17)
    sc = commonroad_scenario(options, seed=1)
    run_closed_loop(options, sc, "distance", boundary_provider(sc), 40, oracle_threads=os.cpu_count() or 1)


def test_road_network_triple_speed_step():
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=10, Hp=6, mpa_type=MpaType.triple_speed, max_vehicles=16, max_nodes=1 << 17)
    sc = commonroad_scenario(options, seed=3)
    run_closed_loop(options, sc, "distance", boundary_provider(sc), 8)


@pytest.mark.parametrize("bound,weight", [(2, "distance"), (3, "constant"), (1, "distance")])
def test_road_network_with_cut_couplings(bound, weight):
    """options.max_num_CLs < depth of the coupling DAG: GreedyCutter keeps what fits, the other predecessors enter as
    literal obstacles built from their previous plans (PrioritizedController.m:409-447).  Wide levels, same records."""
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=6, max_num_CLs=bound, max_vehicles=32, max_nodes=1 << 17)
    sc = commonroad_scenario(options, seed=2)
    ctl = run_closed_loop(options, sc, "distance", boundary_provider(sc), 10, weight_strategy=weight)
    assert int(ctl.last_levels.max()) <= bound


@pytest.mark.parametrize("strategy", ["random", "fca", "coloring"])
def test_road_network_with_other_prioritizers(strategy):
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=16, Hp=6, max_vehicles=32, max_nodes=1 << 17)
    sc = commonroad_scenario(options, seed=4)
    run_closed_loop(options, sc, "distance", boundary_provider(sc), 8, priority_strategy=strategy)


def test_pop_count_with_keys_that_are_not_monotone():
    """Entries known to collide leave the open list without being popped; whether the reference would have popped such an
    entry depends on the keys of the pops made after it left (a child can have a smaller key than its parent, so the
    front does not move monotonically).  On the triple-speed MPA a comparison with the last key alone miscounts in steps
    14 and 16 of this closed loop (found by tools/stress_parity.py)."""
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=6, mpa_type=MpaType.triple_speed, max_vehicles=32, max_nodes=1 << 16)
    sc = commonroad_scenario(options, seed=3)
    run_closed_loop(options, sc, "distance", boundary_provider(sc), 18)


def test_sharded_planner_world1_on_gpu_matches_single_launch():
    """pdmpc.distributed with one rank: launch_range per level + export/import path of the C ABI."""
    import torch

    from oracle import oracle
    from pdmpc.distributed import HipRangePlanner, plan_step_sharded
    from pdmpc.optimizer import GraphSearchHip
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=16, Hp=6, max_vehicles=16, max_nodes=1 << 16)
    mpa = get_mpa(options)
    sc = commonroad_scenario(options, seed=5)
    opt = GraphSearchHip(options)
    planner = HipRangePlanner(opt, mpa, torch.device("cuda", 0))
    ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc))

    def plan_step(prob):
        gpu = plan_step_sharded(prob, planner, None, 0, 1)
        ref, _ = oracle.plan_step(options, mpa, prob)
        assert_records_equal(gpu, ref, "sharded")
        # import path: overwrite level-1 slots with their own exported records, re-plan the rest
        return [info_from_record(gpu[i], options.Hp) for i in range(len(gpu))]

    for _ in range(5):
        ctl.step(plan_step=plan_step)
    opt.handle.close()


def test_config_c3_128_vehicles_two_level_coupling_dag():
    """BASELINE config 2 as named: 128 vehicles on tiled copies of the map, Hp 8, colouring priorities
    (ColoringPrioritizer.m:31-89), coupling DAG cut to 2 computation levels (max_num_CLs = 2, Config.m:28,
    GreedyCutter.m:25-86); the cut couplings enter as previous-trajectory obstacles (PrioritizedController.m:409-447).
    Whole step in one launch, arenas grow on demand, records byte-identical to the oracle's level loop."""
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=128, Hp=8, max_num_CLs=2, max_vehicles=128, max_nodes=1 << 15)
    sc = commonroad_scenario(options, seed=2, tiles=7)
    ctl = run_closed_loop(options, sc, "distance", boundary_provider(sc), 4, oracle_threads=os.cpu_count() or 1, priority_strategy="coloring")
    assert int(ctl.last_levels.max()) == 2


def test_benchmarked_window_c3_steps_1_to_12():
    """bench.py --workload c3 records closed-loop steps 5-12 of seed 1 (7 tiles): the same closed loop, every step against the oracle."""
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=128, Hp=8, max_num_CLs=2, max_vehicles=128, max_nodes=1 << 16)
    sc = commonroad_scenario(options, seed=1, tiles=7)
    ctl = run_closed_loop(options, sc, "distance", boundary_provider(sc), 12, oracle_threads=os.cpu_count() or 1, priority_strategy="coloring")
    assert int(ctl.last_levels.max()) == 2


def test_config_c4_512_vehicles_hp10_colouring_levels():
    """BASELINE config 3 as named: 512 vehicles, Hp 10, computation levels from graph colouring + kahn
    (ColoringPrioritizer.m:31-89, utility/kahn.m:1-24); two workgroups per CU.  The arenas start small and grow until no
    search overflows, as the reference's unbounded tree would (Tree.m:54-70)."""
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=512, Hp=10, max_vehicles=512, max_nodes=1 << 15)
    sc = commonroad_scenario(options, seed=3, tiles=26)
    ctl = run_closed_loop(options, sc, "distance", boundary_provider(sc), 2, oracle_threads=os.cpu_count() or 1, priority_strategy="coloring")
    assert int(ctl.last_levels.max()) >= 3


@pytest.mark.parametrize("tuning", ["helpers_first=0", "helpers_first=200,seat_nodes=16", "helpers_oversub=40,helpers_first=40,share_min=64"])
def test_c4_with_the_helper_workgroups_in_other_places(tuning, monkeypatch):
    """A launch of more searches than CUs puts half a chip's worth of helper workgroups in FRONT of the searches when most searches
    have predecessors (api.cpp: launch_range, helpers_first) — the searches' workgroup indices then start behind them.  The same
    closed loop with none in front, with all of them in front, and with a few that share small rounds: the oracle's records."""
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    monkeypatch.setenv("PDMPC_TUNING", tuning)
    options = Config(scenario_type=ScenarioType.commonroad, amount=512, Hp=10, max_vehicles=512, max_nodes=1 << 15)
    sc = commonroad_scenario(options, seed=3, tiles=26)
    ctl = run_closed_loop(options, sc, "distance", boundary_provider(sc), 2, oracle_threads=os.cpu_count() or 1, priority_strategy="coloring")
    assert int(ctl.last_levels.max()) >= 3


def test_benchmarked_window_c4_steps_1_to_12():
    """bench.py --workload c4 records closed-loop steps 5-12 of seed 1 (26 tiles, 512 vehicles, Hp 10, colouring levels): the
    same closed loop from standstill through step 12, every step against the oracle."""
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=512, Hp=10, max_vehicles=512, max_nodes=1 << 16)
    sc = commonroad_scenario(options, seed=1, tiles=26)
    ctl = run_closed_loop(options, sc, "distance", boundary_provider(sc), 12, oracle_threads=os.cpu_count() or 1, priority_strategy="coloring")
    assert int(ctl.last_levels.max()) >= 3


def test_c4_resident_banks_replayed_without_the_safety_net():
    """What bench.py --workload c4 does in its timed loop: recorded steps stay packed in HBM (one bank each) and are launched again
    and again with pdmpc_launch_packed -- no re-plan on a predecessor time-out, no arena growth.  Every replay must end with the
    recorded records and without a single error status (device-side counter), also for the heavy steps beyond the first dozen."""
    from pdmpc.optimizer import GraphSearchHip
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=512, Hp=10, max_vehicles=512, max_nodes=1 << 16)  # (bench.py's initial arena: it grows)
    sc = commonroad_scenario(options, seed=1, tiles=26)
    mpa = get_mpa(options)
    opt = GraphSearchHip(options)
    opt._ensure_mpa(mpa)
    h = opt.handle
    ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc), priority_strategy="coloring")
    problems = []

    def plan_step(prob):
        problems.append(prob)
        fb = [f if f is not None else [] for f in prob["fallback"]]
        recs = h.plan_step(prob["iters"], prob["preds"], fb)
        return [info_from_record(recs[i], options.Hp) for i in range(len(recs))]

    for _ in range(12):
        ctl.step(plan_step=plan_step)
    assert h.stats()["safe_replans"] == 0
    banks = problems[4:]  # (bench.py --workload c4: 4 steps skipped, 8 recorded)
    recorded = []
    h.allow_overflow = True
    for b, prob in enumerate(banks):
        h.select_bank(b)
        fb = [f if f is not None else [] for f in prob["fallback"]]
        h.pack_step(prob["iters"], prob["preds"], fb)
        while True:  # (bench.py: pack_banks -- a bank recorded before the arenas grew replays in the grown arenas)
            h.launch()
            recs = h.fetch(len(prob["iters"]))
            if not (recs["status"] == 2).any():
                break
            h.grow_arena(2 * h.arena_nodes()[0])
        recorded.append(recs.copy())
        assert ((recs["status"] == 0) | (recs["status"] == 1)).all()
    h.allow_overflow = False
    h.reset_stats()
    for rep in range(5):
        for b, prob in enumerate(banks):
            h.select_bank(b)
            h.launch()
            h.synchronize()
    assert h.stats()["bad_status_plans"] == 0
    for b, prob in enumerate(banks):
        h.select_bank(b)
        h.launch()
        assert_records_equal(h.fetch(len(prob["iters"])), recorded[b], "replay of bank %d" % b)
    h.close()


@pytest.mark.parametrize("n_instances", [12, 64])
def test_explorative_batch_of_prioritizations_one_launch(n_instances):
    """BASELINE config 4 (64 instances = the config as named): several prioritizations of the same traffic state flattened
    into one launch (PrioritizedExplorativeController.m:25-176), records identical to the oracle, same chosen prioritization."""
    from oracle import oracle
    from pdmpc.explorative import build_exploration_batch, choose_solution
    from pdmpc.optimizer import GraphSearchHip
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=8, max_vehicles=20 * n_instances, max_nodes=1 << 15)
    mpa = get_mpa(options)
    sc = commonroad_scenario(options, seed=1)
    opt = GraphSearchHip(options)
    ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc))
    # (bench.py --workload c5 takes its first batch after 4 closed-loop steps with seed = time step: the 64-instance case is that batch)
    n_before = 4 if n_instances == 64 else 3
    for _ in range(n_before):
        ctl.step(plan_step=lambda prob: opt.run_optimizer_step(prob, mpa))
    batch = build_exploration_batch(ctl, n_instances, seed=ctl.k + 1 if n_instances == 64 else 4)
    assert len(batch["iters"]) == 20 * n_instances and batch["n_instances"] == n_instances
    n = len(batch["iters"])
    fb = [f if f is not None else [] for f in batch["fallback"]]
    gpu = opt.handle.plan_step(batch["iters"], batch["preds"], fb)
    unbounded = copy.copy(options)
    unbounded.max_nodes = 1 << 30
    ref, _ = oracle.plan_step(unbounded, mpa, batch, n_threads=os.cpu_count() or 1)
    assert_records_equal(gpu, ref, "explorative batch")
    chosen_gpu, cost_gpu = choose_solution(batch, gpu, options.Hp)
    chosen_ref, cost_ref = choose_solution(batch, ref, options.Hp)
    assert chosen_gpu == chosen_ref and np.array_equal(cost_gpu, cost_ref)
    assert np.isfinite(cost_gpu[0]).all()  # the current prioritization is feasible
    opt.handle.close()


def test_fallback_while_a_predecessor_is_still_planning():
    """Vehicle 1 (mirror-symmetric: tied keys -> binary-heap fallback) starts speculatively while its predecessor,
    a long search, is still running; the predecessor's areas arrive during or after the fallback."""
    from oracle import oracle
    from pdmpc.optimizer import GraphSearchHip
    import problems

    options = problems.make_options("interx", Hp=8)
    options.max_vehicles = 8
    options.max_nodes = 1 << 16
    mpa = problems.get_mpa(options)
    rng = np.random.default_rng(2)
    heavy = max((problems.road_problem(rng, options, mpa) for _ in range(12)), key=lambda it: len(it.dynamic_obstacle_area) + len(it.obstacles))
    sym = problems.symmetric_problem(options, mpa, block_x=0.5)
    prob = {"iters": [heavy, sym, sym], "preds": [[], [0], [0, 1]], "fallback": [None, None, None], "level_sizes": [1, 1, 1]}
    opt = GraphSearchHip(options)
    try:
        opt._ensure_mpa(mpa)
        opt.handle.pack_step(prob["iters"], prob["preds"], [[], [], []])
        opt.handle.launch()
        gpu = opt.handle.fetch(3)
        ref, _ = oracle.plan_step(options, mpa, prob)
        assert_records_equal(gpu, ref, "fallback under speculation")
        assert opt.handle.stats()["kernel"] == 2
    finally:
        opt.handle.close()


def native_closed_loop_on_device(options, scenario, boundary, n_steps, **kw):
    """pdmpc_controller_step with a real handle (C++ build_step -> pdmpc_plan_step on the GPU -> C++ apply: the path behind
    bench.py's value_host_inclusive) next to the Python controller planned by the oracle: the records of every step and the
    plant state after every step are identical."""
    from oracle import oracle
    from pdmpc.native_controller import NativeController
    from pdmpc.optimizer import GraphSearchHip

    mpa = get_mpa(options)
    opt = GraphSearchHip(options)
    opt._ensure_mpa(mpa)
    nat = NativeController(options, scenario, mpa, opt.handle, coupling="distance", **kw)
    py = PrioritizedSequentialController(options, scenario, mpa, None, coupling="distance", boundary_provider=boundary, **kw)
    unbounded = copy.copy(options)
    unbounded.max_nodes = 1 << 30
    for k in range(n_steps):
        gpu = nat.step()  # records of the native step in slot order
        ref_box = []

        def plan_step(prob):
            ref, _ = oracle.plan_step(unbounded, mpa, prob, n_threads=os.cpu_count() or 1)
            ref_box.append(ref)
            return [info_from_record(ref[i], options.Hp) for i in range(len(ref))]

        py.step(plan_step=plan_step)
        assert_records_equal(gpu, ref_box[0], "native step %d" % (k + 1))
        st = nat.state()
        assert st["k"] == k + 1
        assert np.array_equal(st["x"], np.array([m.x for m in py.meas])) and np.array_equal(st["y"], np.array([m.y for m in py.meas])), k
        assert np.array_equal(st["yaw"], np.array([m.yaw for m in py.meas])), k
        assert np.array_equal(st["speed"], np.array([m.speed for m in py.meas])) and np.array_equal(st["steering"], np.array([m.steering for m in py.meas])), k
        assert st["needs_fallback"].tolist() == [bool(i.needs_fallback) for i in py.infos], k
    # pdmpc_controller_run: the same closed loop in one native call ends in the same plant state
    nat2 = NativeController(options, scenario, mpa, opt.handle, coupling="distance", **kw)
    ms = nat2.run(n_steps)
    assert len(ms) == n_steps and (ms > 0).all()
    a, b = nat.state(), nat2.state()
    for key in ("x", "y", "yaw", "speed", "steering"):
        assert np.array_equal(a[key], b[key]), key
    nat.close()
    nat2.close()
    opt.handle.close()


def test_native_controller_on_device_c2_40_steps():
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=8, max_vehicles=32, max_nodes=1 << 17)
    sc = commonroad_scenario(options, seed=1)
    native_closed_loop_on_device(options, sc, boundary_provider(sc), 40)


def test_native_controller_on_device_c3():
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=128, Hp=8, max_num_CLs=2, max_vehicles=128, max_nodes=1 << 16)
    sc = commonroad_scenario(options, seed=1, tiles=7)
    native_closed_loop_on_device(options, sc, boundary_provider(sc), 8, priority_strategy="coloring")


def test_native_explorative_step_on_device():
    """SURVEY.md 8(f)-2 through the C ABI: pdmpc_controller_explore_step (the prioritizations of the step built and flattened in
    C++, ONE launch, choice per sub-graph, apply of the chosen plans: the path behind bench.py --workload c5's
    value_host_inclusive) next to pdmpc.explorative.explore_step planned by the oracle: records of every batch, the chosen
    prioritization of every vehicle and the plant state after every step are identical."""
    from oracle import oracle
    from pdmpc.explorative import explore_step
    from pdmpc.native_controller import NativeController
    from pdmpc.optimizer import GraphSearchHip
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    K = 8
    options = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=8, max_vehicles=20 * K, max_nodes=1 << 16)
    sc = commonroad_scenario(options, seed=1)
    mpa = get_mpa(options)
    opt = GraphSearchHip(options)
    opt._ensure_mpa(mpa)
    nat = NativeController(options, sc, mpa, opt.handle, coupling="distance")
    py = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc))
    unbounded = copy.copy(options)
    unbounded.max_nodes = 1 << 30
    other = 0
    for k in range(10):
        gpu, chosen_nat = nat.explore_step(K)
        _, ref, chosen = explore_step(py, lambda batch: oracle.plan_step(unbounded, mpa, batch, n_threads=os.cpu_count() or 1)[0], K)
        assert_records_equal(gpu, ref, "explorative step %d" % (k + 1))
        assert chosen_nat.tolist() == chosen, k
        other += sum(1 for c in chosen if c != 0)
        st = nat.state()
        assert np.array_equal(st["x"], np.array([m.x for m in py.meas])) and np.array_equal(st["y"], np.array([m.y for m in py.meas])), k
        assert np.array_equal(st["yaw"], np.array([m.yaw for m in py.meas])) and np.array_equal(st["speed"], np.array([m.speed for m in py.meas])), k
    assert other > 0
    # pdmpc_controller_explore_run keeps the chosen plans only (status + final cost of every plan for the choice, then the chosen
    # vehicles' records: pdmpc_plan_step_lean / pdmpc_fetch_records_at): the same closed loop, the same plant state
    nat2 = NativeController(options, sc, mpa, opt.handle, coupling="distance")
    ms = nat2.explore_run(K, 10)
    assert len(ms) == 10 and (ms > 0).all()
    a, b = nat.state(), nat2.state()
    for key in ("x", "y", "yaw", "speed", "steering"):
        assert np.array_equal(a[key], b[key]), key
    tm = nat2.timing_mean()
    assert tm["build"] > 0 and tm["wait_and_read_back"] > 0
    nat2.close()
    nat.close()
    opt.handle.close()


def test_literal_run_optimizer_loop_equals_the_single_launch():
    """pdmpc_plan_step_literal (one pdmpc_plan_batch of one vehicle per run_optimizer call, hand-over on the host: what an
    unmodified reference controller does with GraphSearchHip.m) against pdmpc_plan_step and the oracle over a closed loop."""
    from oracle import oracle
    from pdmpc.optimizer import GraphSearchHip
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=8, max_vehicles=32, max_nodes=1 << 17)
    mpa = get_mpa(options)
    sc = commonroad_scenario(options, seed=3)
    opt = GraphSearchHip(options)
    opt._ensure_mpa(mpa)
    ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc))

    def plan_step(prob):
        fb = [f if f is not None else [] for f in prob["fallback"]]
        lit = opt.handle.plan_step_literal(opt.handle.step_args(prob["iters"], prob["preds"], fb))
        one = opt.handle.plan_step(prob["iters"], prob["preds"], fb)
        ref, _ = oracle.plan_step(options, mpa, prob)
        assert_records_equal(lit, ref, "literal")
        assert_records_equal(one, ref, "single launch")
        return [info_from_record(one[i], options.Hp) for i in range(len(one))]

    for _ in range(10):
        ctl.step(plan_step=plan_step)
    opt.handle.close()


def test_batch_not_in_level_order_is_reordered_by_the_library():
    """Slots handed over against the level order (successors first): pdmpc_pack_step puts the batch into level order itself
    and pdmpc_fetch_results hands the records back in the caller's order."""
    from oracle import oracle
    from pdmpc.optimizer import GraphSearchHip
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=6, max_vehicles=32, max_nodes=1 << 16)
    mpa = get_mpa(options)
    sc = commonroad_scenario(options, seed=2)
    opt = GraphSearchHip(options)
    opt._ensure_mpa(mpa)
    ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc))
    rng = np.random.default_rng(5)

    def plan_step(prob):
        n = len(prob["iters"])
        fb = [f if f is not None else [] for f in prob["fallback"]]
        ref, _ = oracle.plan_step(options, mpa, prob)
        for perm in (list(range(n))[::-1], list(rng.permutation(n))):
            inv = {s: i for i, s in enumerate(perm)}  # slot s of the problem is handed over at position inv[s]
            got = opt.handle.plan_step([prob["iters"][s] for s in perm], [[inv[p] for p in prob["preds"][s]] for s in perm], [fb[s] for s in perm])
            assert_records_equal(got, ref[perm], "permuted batch")
        return [info_from_record(ref[i], options.Hp) for i in range(n)]

    for _ in range(4):
        ctl.step(plan_step=plan_step)
    opt.handle.close()


def test_step_weights_fill_the_slots_by_priority_and_change_nothing():
    """pdmpc_set_step_weights: with an expected work per vehicle the slots of the launch are filled by priority (the largest expected
    work among a vehicle and its descendants in the coupling DAG, descending) — a topological order like the level order, so the
    predecessors still sit in lower slots — and the records come back in the caller's order, equal to the oracle's bit for bit.
    Weights that favour the LAST levels are the adversarial case: a late vehicle drags all its ancestors to the front."""
    from oracle import oracle
    from pdmpc.optimizer import GraphSearchHip
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=40, Hp=6, max_num_CLs=4, max_vehicles=64, max_nodes=1 << 16)
    mpa = get_mpa(options)
    sc = commonroad_scenario(options, seed=2, tiles=2)
    opt = GraphSearchHip(options)
    opt._ensure_mpa(mpa)
    ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc), priority_strategy="coloring")
    rng = np.random.default_rng(7)

    def plan_step(prob):
        n = len(prob["iters"])
        fb = [f if f is not None else [] for f in prob["fallback"]]
        ref, _ = oracle.plan_step(options, mpa, prob, n_threads=os.cpu_count() or 1)
        for weights in ([float(l) for l in prob["levels"]], list(rng.integers(1, 5000, n).astype(float)), [1.0] * n, [float("nan")] * n):
            got = opt.handle.plan_step(prob["iters"], prob["preds"], fb, weights=weights)
            assert_records_equal(got, ref, "slots by priority")
        # ... and as a resident bank launched twice
        opt.handle.pack_step(prob["iters"], prob["preds"], fb, weights=list(rng.integers(1, 5000, n).astype(float)))
        for _ in range(2):
            opt.handle.launch()
            assert_records_equal(opt.handle.fetch(n), ref, "resident bank packed by priority")
        return [info_from_record(ref[i], options.Hp) for i in range(n)]

    for _ in range(4):
        ctl.step(plan_step=plan_step)
    assert opt.handle.stats()["safe_replans"] == 0
    opt.handle.close()


def test_starved_predecessors_are_replanned_in_resident_slices(monkeypatch):
    """Forward progress of oversubscribed launches.  PDMPC_TUNING=reverse_dispatch=1 hands the slots out in reverse workgroup order:
    the successors occupy the chip and spin for predecessors that have not been dispatched -- the adversarial order.  With a
    small spin_limit the watchdog ends them with an error status; pdmpc_plan_step then plans the step again in slices that
    are resident as a whole and the records are the oracle's."""
    from oracle import oracle
    from pdmpc.optimizer import GraphSearchHip
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    monkeypatch.setenv("PDMPC_TUNING", "reverse_dispatch=1,spin_limit=20000")
    options = Config(scenario_type=ScenarioType.commonroad, amount=320, Hp=6, max_vehicles=320, max_nodes=1 << 14)
    mpa = get_mpa(options)
    sc = commonroad_scenario(options, seed=1, tiles=16)
    opt = GraphSearchHip(options)
    opt._ensure_mpa(mpa)
    ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc), priority_strategy="coloring")
    unbounded = copy.copy(options)
    unbounded.max_nodes = 1 << 30

    def plan_step(prob):
        fb = [f if f is not None else [] for f in prob["fallback"]]
        gpu = opt.handle.plan_step(prob["iters"], prob["preds"], fb)
        ref, _ = oracle.plan_step(unbounded, mpa, prob, n_threads=os.cpu_count() or 1)
        assert_records_equal(gpu, ref, "after the safe re-plan")
        return [info_from_record(gpu[i], options.Hp) for i in range(len(gpu))]

    for _ in range(2):
        ctl.step(plan_step=plan_step)
    assert opt.handle.stats()["safe_replans"] >= 1
    # and with safe launches from the start nothing has to be planned twice
    before = opt.handle.stats()["safe_replans"]
    opt.handle.set_safe_launch(True)
    ctl.step(plan_step=plan_step)
    assert opt.handle.stats()["safe_replans"] == before
    opt.handle.close()


def test_matlab_shaped_entry_points_plan_the_step_like_the_oracle():
    """pdmpc_ml_upload_mpa + pdmpc_ml_plan_step (include/pdmpc_matlab.h: what the MEX commands `upload_mpa` and `plan_step` call)
    fed with MATLAB-shaped data in VEHICLE order: records per vehicle equal the oracle's over a closed loop."""
    import ctypes as C

    import matlab_shapes as ms
    from oracle import oracle
    from pdmpc.backend import Handle
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=8, max_vehicles=32, max_nodes=1 << 17)
    mpa = get_mpa(options)
    sc = commonroad_scenario(options, seed=4)
    h = Handle(options)
    keep_m = ms.Keep()
    T, n_trims, Hp, man = ms.ml_mpa_args(mpa, keep_m)
    assert ms.lib().pdmpc_ml_upload_mpa(h.h, T, n_trims, Hp, man) == 0
    ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc))

    last = [None]  # n_popped per VEHICLE of the step before: what PrioritizedSequentialHipController.m hands over as weights

    def plan_step(prob):
        n = len(prob["iters"])
        keep = ms.Keep()
        step = ms.step_create(ms.vehicle_order_problem(prob), options.Hp, keep)
        by_vehicle = ms.plan_step(h, step, n, weights=last[0])
        ms.lib().pdmpc_ml_step_destroy(step)
        last[0] = np.asarray(by_vehicle["n_popped"], dtype=np.float64) + 1.0
        ref, _ = oracle.plan_step(options, mpa, prob)
        gpu = by_vehicle[np.asarray(prob["order"])]  # back into slot order
        assert_records_equal(gpu, ref, "matlab-shaped step")
        return [info_from_record(gpu[i], options.Hp) for i in range(n)]

    for _ in range(8):
        ctl.step(plan_step=plan_step)
    h.close()


def test_rccl_paths_with_one_rank_match_the_single_launch():
    """torch.distributed with backend "nccl" (= RCCL) initialised on this one GPU: the level-sharded planner with its all-gather +
    import per level (always_gather: a 1-rank all-gather is a copy through RCCL), the component path (one launch, asynchronous
    export, all_gather_into_tensor on the handle's own stream wrapped as an ExternalStream) and the hybrid planner, all against
    the single launch and the oracle."""
    import socket

    import torch
    import torch.distributed as dist

    from oracle import oracle
    from pdmpc.distributed import REC_BYTES, HipRangePlanner, plan_step_hybrid, plan_step_sharded
    from pdmpc.optimizer import GraphSearchHip
    from pdmpc.road_network import boundary_provider, commonroad_scenario
    from pdmpc import abi

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        options = Config(scenario_type=ScenarioType.commonroad, amount=40, Hp=6, max_vehicles=64, max_nodes=1 << 16)
        mpa = get_mpa(options)
        sc = commonroad_scenario(options, seed=5, tiles=2)
        opt = GraphSearchHip(options)
        planner = HipRangePlanner(opt, mpa, torch.device("cuda", 0))
        ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc))
        h = opt.handle
        ext = torch.cuda.ExternalStream(h.stream_ptr(), device=torch.device("cuda", 0))

        def plan_step(prob):
            n = len(prob["iters"])
            fb = [f if f is not None else [] for f in prob["fallback"]]
            ref, _ = oracle.plan_step(options, mpa, prob)
            single = h.plan_step(prob["iters"], prob["preds"], fb)
            assert_records_equal(single, ref, "single launch")
            # level-sharded: launch_range per level -> export -> RCCL all-gather -> import, all on the handle's stream
            lv = plan_step_sharded(prob, planner, dist, 0, 1, always_gather=True)
            assert_records_equal(lv, ref, "levels through RCCL")
            # component path as bench.py runs it: one launch, export, all-gather on the ExternalStream, records read from the gathered buffer
            h.pack_step(prob["iters"], prob["preds"], fb)
            send = torch.zeros(n * REC_BYTES, dtype=torch.uint8, device="cuda")
            recv = torch.zeros(n * REC_BYTES, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            h.launch()
            h.export_results_async(0, n, send.data_ptr())
            with torch.cuda.stream(ext):
                dist.all_gather_into_tensor(recv, send)
            h.synchronize()
            got = np.frombuffer(recv.cpu().numpy().tobytes(), dtype=abi.VEHICLE_OUT_DTYPE)
            assert_records_equal(got, ref, "components through RCCL")
            hy = plan_step_hybrid(prob, planner, dist, 0, 1)
            assert_records_equal(hy, ref, "hybrid")
            return [info_from_record(single[i], options.Hp) for i in range(n)]

        for _ in range(4):
            ctl.step(plan_step=plan_step)
        h.close()
    finally:
        dist.destroy_process_group()


def test_bulk_kernel_shares_large_rounds_and_parks_tentative_nodes(monkeypatch):
    """The product kernel on the benchmarked C2 window (closed-loop steps 1-30, incl. the 10 k-pop searches), with the thresholds low
    enough that most rounds are shared with helper workgroups: the statistics say that helpers took part, the records stay the
    oracle's (run_closed_loop compares every step); the same loop without tentative areas and without helpers gives them too, and so
    does one whose far lists feed near through the mid list."""
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    def loop():
        options = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=8, max_vehicles=32, max_nodes=1 << 17)
        sc = commonroad_scenario(options, seed=1)
        return run_closed_loop(options, sc, "distance", boundary_provider(sc), 30).handle_stats

    monkeypatch.setenv("PDMPC_TUNING", "share_min=64,tile=32")
    stats = loop()
    assert stats["kernel"] == 2
    assert stats["shared_rounds"] > 0 and 0 < stats["helper_checked"] < stats["nodes_processed"]
    monkeypatch.setenv("PDMPC_TUNING", "share_min=64,tile=32,tentative=0,helpers=0")
    stats = loop()
    assert stats["kernel"] == 2 and stats["shared_rounds"] == 0
    # ... without the early publication of a finished search's areas
    monkeypatch.setenv("PDMPC_TUNING", "fast_arrival=0")
    stats = loop()
    assert stats["kernel"] == 2
    # ... and with the mid list in the way of every far list of more than 512 entries (the default keeps it for lists of 24 k and more)
    monkeypatch.setenv("PDMPC_TUNING", "share_min=64,tile=32,mid_min=512,mid_fill=2048")
    stats = loop()
    assert stats["kernel"] == 2
    # ... and with it in the way of EVERY far list, a few hundred entries at a time: arrivals, invalidated nodes and reopened open sets all
    # meet a mid list
    monkeypatch.setenv("PDMPC_TUNING", "share_min=64,tile=32,mid_min=0,mid_fill=256")
    stats = loop()
    assert stats["kernel"] == 2
    # ... and with every search sent through the replay on the binary heap (what a search with equal keys ends on)
    monkeypatch.setenv("PDMPC_TUNING", "force_tie=1")
    stats = loop()
    assert stats["kernel"] == 2 and stats["queue_fallbacks"] > 0


def test_tied_search_in_a_level_sharded_step_on_the_device_resident_path():
    """A search with equal keys (mirror-symmetric problem) in the middle of a step that is launched level by level and handed on
    through pdmpc_export_results / pdmpc_import_results without a fetch in between (the multi-GPU path): the tie is
    resolved inside the launch (bk_replay), so the records that leave the device are the oracle's — no internal status, no re-plan."""
    import ctypes
    from oracle import oracle
    from pdmpc import abi
    from pdmpc.backend import Handle
    import problems

    options = problems.make_options("interx", Hp=6)
    options.max_vehicles = 8
    options.max_nodes = 1 << 16
    mpa = problems.get_mpa(options)
    rng = np.random.default_rng(4)
    road = [problems.road_problem(rng, options, mpa) for _ in range(3)]
    sym = problems.symmetric_problem(options, mpa, block_x=0.5)
    prob = {"iters": [road[0], sym, road[1], sym, road[2]], "preds": [[], [], [0, 1], [1], [2, 3]], "fallback": [None] * 5, "level_sizes": [2, 2, 1]}
    ref, _ = oracle.plan_step(options, mpa, prob)
    _, _, traces = oracle.plan_batch(options, mpa, [sym], trace=True)
    assert problems.tied_pops(traces[0]) > 0
    h = Handle(options)
    h.upload_mpa(mpa)
    h.pack_step(prob["iters"], prob["preds"], [[], [], [], [], []])
    h.begin_step()
    hip = ctypes.CDLL("libamdhip64.so")
    nbytes = abi.VEHICLE_OUT_DTYPE.itemsize
    buf = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(buf), 5 * nbytes) == 0
    first = 0
    for n in prob["level_sizes"]:
        h.launch_range(first, n)
        h.export_results(first, n, buf.value + first * nbytes)   # (what the all-gather would send)
        h.import_results(first, n, buf.value + first * nbytes)   # (... and what every rank takes in)
        first += n
    host = np.zeros(5, dtype=abi.VEHICLE_OUT_DTYPE)
    assert hip.hipMemcpy(host.ctypes.data_as(ctypes.c_void_p), buf, 5 * nbytes, 2) == 0
    hip.hipFree(buf)
    assert set(int(x) for x in host["status"]) <= {0, 1}
    assert_records_equal(host, ref, "level-sharded step with a tied search")
    st = h.stats()
    assert st["kernel"] == 2 and st["queue_fallbacks"] >= 2
    h.close()
