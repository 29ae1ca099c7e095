"""The sampled optimizer (MonteCarloTreeSearch.m) on the GPU against the oracle's restatement: records byte-identical."""
import numpy as np
import pytest

from pdmpc.backend import Handle
from pdmpc.config import OptimizerType
from pdmpc.optimizer import OptimizerInterface

import problems
from test_gpu_parity import assert_records_equal

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import oracle

    return oracle


@pytest.mark.parametrize("mode,seed,Hp", [("interx", 1, 6), ("interx", 2, 8), ("sat", 3, 5), ("sat", 4, 8)])
def test_sampled_matches_oracle(mode, seed, Hp):
    options, mpa, iters = problems.problem_set(mode, seed, 12, Hp=Hp)
    options.max_vehicles = 16
    seeds = [7 + 3 * i for i in range(len(iters))]
    h = Handle(options)
    h.upload_mpa(mpa)
    gpu = h.plan_batch_sampled(iters, seeds)
    _, ref = _oracle().plan_batch_sampled(options, mpa, iters, seeds)
    h.close()
    assert_records_equal(gpu, ref, "sampled %s" % mode)
    assert (ref["status"] == 0).any()


def test_sampled_triple_speed_and_realistic_mpa():
    from pdmpc.config import MpaType

    for mpa_type, Hp in ((MpaType.triple_speed, 6), (MpaType.realistic, 5)):
        options, mpa, iters = problems.problem_set("interx", 5, 4, Hp=Hp, mpa_type=mpa_type)
        options.max_vehicles = 8
        seeds = [100 + i for i in range(len(iters))]
        h = Handle(options)
        h.upload_mpa(mpa)
        gpu = h.plan_batch_sampled(iters, seeds)
        _, ref = _oracle().plan_batch_sampled(options, mpa, iters, seeds)
        h.close()
        assert_records_equal(gpu, ref, "sampled %s" % mpa_type)


def test_sampled_exhaustion_and_plugin_interface():
    options = problems.make_options("interx", Hp=6)
    options.optimizer_type = OptimizerType.HipSampled
    options.max_vehicles = 4
    mpa = problems.get_mpa(options)
    rng = np.random.default_rng(9)
    it = problems.road_problem(rng, options, mpa)
    # an obstacle through the vehicle itself: every first edge collides -> exhausted (MonteCarloTreeSearch.m:212-215)
    x, y = float(it.x0[0]), float(it.x0[1])
    it.obstacles = [np.array([[x - 0.5, x + 0.5, x + 0.5, x - 0.5], [y - 0.02, y - 0.02, y + 0.02, y + 0.02]])]
    opt = OptimizerInterface.get_optimizer(options)
    info = opt.run_optimizer(2, it, mpa, options, time_step=5)
    ref_infos, ref = _oracle().plan_batch_sampled(options, mpa, [it], [7])
    assert info.is_exhausted and ref_infos[0].is_exhausted
    free = problems.road_problem(rng, options, mpa, n_dyn=0, n_static=0)
    info = opt.run_optimizer(1, free, mpa, options, time_step=3)
    ref_infos, _ = _oracle().plan_batch_sampled(options, mpa, [free], [4])
    assert not info.is_exhausted
    assert np.array_equal(info.y_predicted.view(np.uint64), ref_infos[0].y_predicted.view(np.uint64))
    assert list(info.tree_path) == list(ref_infos[0].tree_path)
    opt.handle.close()


def test_sampled_closed_loop_level_by_level():
    """PrioritizedSequentialController with the sampled optimizer, level by level (the reference's own loop,
    PrioritizedSequentialController.m:77-94): the GPU closed loop equals the oracle's closed loop step for step."""
    from pdmpc.config import Config, ScenarioType
    from pdmpc.controller import PrioritizedSequentialController
    from pdmpc.iteration_data import info_from_record
    from pdmpc.mpa import get_mpa
    from pdmpc.scenario import circle_scenario

    options = Config(scenario_type=ScenarioType.circle, amount=4, Hp=5, max_vehicles=8)
    mpa = get_mpa(options)
    h = Handle(options)
    h.upload_mpa(mpa)
    trajectories = {}
    for who in ("gpu", "oracle"):

        def plan_level(iters, seeds, who=who):
            if who == "gpu":
                recs = h.plan_batch_sampled(iters, seeds)
            else:
                _, recs = _oracle().plan_batch_sampled(options, mpa, iters, seeds)
            return [info_from_record(recs[i], options.Hp) for i in range(len(iters))]

        plan_level.wants_seeds = True
        ctl = PrioritizedSequentialController(options, circle_scenario(options), mpa, plan_level, coupling="full")
        states = []
        for _ in range(8):
            infos = ctl.step()
            states.append(np.array([[m.x, m.y, m.yaw] for m in ctl.meas]))
        trajectories[who] = np.array(states)
    h.close()
    assert np.array_equal(trajectories["gpu"].view(np.uint64), trajectories["oracle"].view(np.uint64))
    assert not np.array_equal(trajectories["gpu"][0], trajectories["gpu"][-1])  # the vehicles moved
