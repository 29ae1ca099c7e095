"""Diagnostic: the sampled optimizer — kernel time of one computation level on the GPU against the oracle on the host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from pdmpc.backend import Handle
from oracle import oracle
import problems
for n in (1, 20, 256, 1024):
    options, mpa, iters = problems.problem_set("interx", 1, n, Hp=8)
    options.max_vehicles = max(n, 32)
    options.max_nodes = 4096
    h = Handle(options)
    h.upload_mpa(mpa)
    seeds = list(range(5, 5 + n))
    h.plan_batch_sampled(iters, seeds)
    h.reset_stats()
    t0 = time.perf_counter()
    rec = h.plan_batch_sampled(iters, seeds)
    wall = time.perf_counter() - t0
    st = h.stats()
    t0 = time.perf_counter()
    _, ref = oracle.plan_batch_sampled(options, mpa, iters, seeds, n_threads=os.cpu_count())
    cpu = time.perf_counter() - t0
    same = all(np.array_equal(rec[k].view(np.uint8), ref[k].view(np.uint8)) for k in range(0))
    print("%5d vehicles: kernel %.3f ms (call %.1f ms incl. host-side random numbers and packing), oracle on %d threads %.1f ms, expansions/vehicle %.0f"
          % (n, st["kernel_ms"], 1e3 * wall, os.cpu_count(), 1e3 * cpu, rec["n_expanded"].mean()))
    h.close()

# closed loop: 20 vehicles on the road network, the controller's level loop calling the sampled optimizer per level
from pdmpc.config import Config, ScenarioType
from pdmpc.controller import PrioritizedSequentialController
from pdmpc.iteration_data import info_from_record
from pdmpc.mpa import get_mpa
from pdmpc.road_network import boundary_provider, commonroad_scenario

options = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=8, max_vehicles=32, max_nodes=4096)
mpa = get_mpa(options)
h = Handle(options)
h.upload_mpa(mpa)
for who in ("gpu", "oracle"):
    def plan_level(iters, seeds, who=who):
        if who == "gpu":
            recs = h.plan_batch_sampled(iters, seeds)
        else:
            _, recs = oracle.plan_batch_sampled(options, mpa, iters, seeds, n_threads=os.cpu_count())
        return [info_from_record(recs[i], options.Hp) for i in range(len(iters))]
    plan_level.wants_seeds = True
    sc = commonroad_scenario(options, seed=1)
    ctl = PrioritizedSequentialController(options, sc, mpa, plan_level, coupling="distance", boundary_provider=boundary_provider(sc))
    for _ in range(5):
        ctl.step()
    t0 = time.perf_counter()
    for _ in range(30):
        ctl.step()
    dt = time.perf_counter() - t0
    print("closed loop, 20 vehicles, sampled optimizer, level by level, %s: %.1f MPC steps/s (host driver in Python included)" % (who, 30 / dt))
h.close()
