"""HIP backend against the COMMITTED golden plans (tests/golden/oracle_plans_*.npz) — no oracle in the loop."""
import os

import numpy as np
import pytest

from pdmpc.backend import Handle
from pdmpc.config import MpaType

import problems

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["interx_single_hp6", "sat_single_hp5", "interx_triple_hp8"])
def test_gpu_reproduces_golden_plans(name):
    g = np.load(os.path.join(GOLDEN, "oracle_plans_%s.npz" % name))
    options, mpa, iters = problems.problem_set(str(g["mode"]), int(g["seed"]), int(g["count"]), Hp=int(g["Hp"]), mpa_type=MpaType[str(g["mpa_type"])])
    options.max_nodes = 1 << 15
    options.max_vehicles = len(iters)
    options.trace_pops = 64
    h = Handle(options)
    h.upload_mpa(mpa)
    recs = h.plan_batch(iters)
    assert np.array_equal(recs.view(np.uint8).reshape(len(iters), -1), g["records"])
    for v in range(len(iters)):
        pops = h.pop_trace(v, capacity=64)
        assert np.array_equal(pops, g["first_pops"][v, : len(pops)])
        assert h.tree(v)["x"].shape[0] == g["tree_sizes"][v]
    h.close()


@pytest.mark.parametrize("name", ["sampled_interx_hp6", "sampled_sat_hp8"])
def test_gpu_reproduces_golden_sampled_plans(name):
    g = np.load(os.path.join(GOLDEN, "oracle_plans_%s.npz" % name))
    options, mpa, iters = problems.problem_set(str(g["mode"]), int(g["seed"]), int(g["count"]), Hp=int(g["Hp"]))
    options.max_vehicles = len(iters)
    h = Handle(options)
    h.upload_mpa(mpa)
    recs = h.plan_batch_sampled(iters, g["rng_seeds"].tolist())
    h.close()
    assert np.array_equal(recs.view(np.uint8).reshape(len(iters), -1), g["records"])
