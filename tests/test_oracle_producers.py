"""The product's input producers (pdmpc.reference_trajectory, pdmpc.road_network, pdmpc.controller: SURVEY.md 8(f)-3) against
the oracle's independent restatement of the same reference functions (oracle/producers.py, written from the .m files alone)
over closed loops: every per-step input the optimizer receives — pose, trim, reference points, reference speed, predicted
lanelet boundary, standstill obstacles, areas published on exhaustion — and the plant update, bit for bit.  CPU only."""
import numpy as np

from pdmpc.config import Config, ScenarioType
from pdmpc.controller import PrioritizedSequentialController
from pdmpc.iteration_data import info_from_record
from pdmpc.mpa import get_mpa

from oracle import producers as P


def bits(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64)).view(np.uint64)


def same(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def run(options, scenario, coupling, boundary, n_steps, lanelet_boundaries=None):
    from oracle import oracle

    mpa = get_mpa(options)
    ctl = PrioritizedSequentialController(options, scenario, mpa, None, coupling=coupling, boundary_provider=boundary)
    trims_speed = [t.speed for t in mpa.trims]
    trims_steering = [t.steering for t in mpa.trims]
    Hp = options.Hp
    checked = {"steps": 0, "boundaries": 0, "standstill": 0, "shifted": 0}
    old_shapes = [None] * options.amount

    def plan_step(prob):
        # the plant state the controller measured at the beginning of this step
        meas = [(m.x, m.y, m.yaw, m.speed, m.steering) for m in ctl.meas]
        for s, v in enumerate(prob["order"]):
            it = prob["iters"][s]
            veh = scenario.vehicles[v]
            x, y, yaw, speed, steering = meas[v]
            trim = P.trim_from_values(trims_speed, trims_steering, speed, steering)
            assert trim == it.trim_index, (v, trim, it.trim_index)
            assert same(it.x0[:3], [x, y, yaw])
            path, points_index, v_ref, cpi = P.get_reference_trajectory(Hp, trims_speed[trim - 1], veh.reference_path, veh.reference_speed, x, y, options.dt_seconds)
            assert same(it.reference_trajectory_points, path), "reference points of vehicle %d in step %d" % (v + 1, ctl.k)
            assert same(it.v_ref, v_ref)
            if lanelet_boundaries is not None:
                predicted, _ = P.get_predicted_lanelets(veh.reference_path.shape[0], veh.points_index, veh.lanelets_index, points_index, cpi)
                left, right = P.get_lanelets_boundary(predicted, lanelet_boundaries, list(veh.lanelets_index), veh.is_loop)
                off = np.array(scenario.tile_offset[v]).reshape(2, 1)
                assert same(it.predicted_lanelet_boundary[0], left + off) and same(it.predicted_lanelet_boundary[1], right + off), "boundary of vehicle %d" % (v + 1)
                checked["boundaries"] += 1
            else:
                assert it.predicted_lanelet_boundary[0] is None or np.size(it.predicted_lanelet_boundary[0]) == 0
            # areas published on exhaustion: the standstill rectangle (no offset, PrioritizedController.m:602-611) or the previous plan shifted (:678-718)
            fb = prob["fallback"][s]
            if trims_speed[trim - 1] == 0:
                _, plain = P.get_occupied_areas(x, y, yaw, veh.Length, veh.Width, options.offset)
                assert fb is not None and all(same(a, plain) for a in fb)
                checked["standstill"] += 1
            elif old_shapes[v] is not None:
                want = P.del_first_rpt_last(old_shapes[v])
                assert len(fb) == Hp and all(same(a, b) for a, b in zip(fb, want))
                checked["shifted"] += 1
            # standstill successors enter as static obstacles with the offset rectangle (PrioritizedController.m:533-540)
            n_static = len(scenario.obstacles)
            for o in it.obstacles[n_static:]:
                cands = [P.get_occupied_areas(*meas[j][:3], scenario.vehicles[j].Length, scenario.vehicles[j].Width, options.offset)[0]
                         for j in range(options.amount) if abs(meas[j][3]) < 0.01]
                assert any(same(o, c) for c in cands)
        ref, _ = oracle.plan_step(options, mpa, prob)
        checked["steps"] += 1
        return [info_from_record(ref[i], Hp) for i in range(len(ref))]

    for _ in range(n_steps):
        infos = ctl.step(plan_step=plan_step)
        for v, info in enumerate(infos):
            want = P.simulation_apply(info.y_predicted, info.predicted_trims, trims_speed, trims_steering)
            m = ctl.meas[v]
            assert same([m.x, m.y, m.yaw, m.speed, m.steering], want), "plant update of vehicle %d" % (v + 1)
            old_shapes[v] = list(info.shapes)
    return checked


def test_circle_scenario_inputs_over_the_whole_run():
    """Appendix C of SURVEY.md: C1, 3 vehicles on the circle, T_end 4 s = 20 steps."""
    from pdmpc.scenario import circle_scenario

    options = Config(scenario_type=ScenarioType.circle, amount=3, Hp=5, T_end=4, max_nodes=1 << 30)
    c = run(options, circle_scenario(options), "full", None, options.k_end)
    assert c["steps"] == 20 and c["standstill"] >= 3 and c["shifted"] > 30


def test_road_network_inputs_incl_predicted_lanelet_boundaries():
    from pdmpc.road_network import boundary_provider, commonroad_scenario, lab_map

    options = Config(scenario_type=ScenarioType.commonroad, amount=16, Hp=6, max_nodes=1 << 30)
    sc = commonroad_scenario(options, seed=2)
    c = run(options, sc, "distance", boundary_provider(sc), 10, lanelet_boundaries=lab_map().boundary)
    assert c["steps"] == 10 and c["boundaries"] == 160 and c["shifted"] > 100


def test_tiled_road_network_inputs():
    from pdmpc.road_network import boundary_provider, commonroad_scenario, lab_map

    options = Config(scenario_type=ScenarioType.commonroad, amount=40, Hp=5, max_nodes=1 << 30)
    sc = commonroad_scenario(options, seed=3, tiles=2)
    c = run(options, sc, "distance", boundary_provider(sc), 4, lanelet_boundaries=lab_map().boundary)
    assert c["boundaries"] == 160


def test_trim_from_values_with_steering():
    """The branch the simulation never takes with steering == 0 only: a measured steering angle between two trims."""
    options = Config(scenario_type=ScenarioType.commonroad, Hp=5)
    mpa = get_mpa(options)
    sp = [t.speed for t in mpa.trims]
    st = [t.steering for t in mpa.trims]
    rng = np.random.default_rng(0)
    for _ in range(200):
        speed, steering = float(rng.uniform(0, 0.9)), float(rng.uniform(-0.6, 0.6))
        assert P.trim_from_values(sp, st, speed, steering) == mpa.trim_from_values(speed, steering)
