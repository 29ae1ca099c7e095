"""Host cost of building a step problem in the native controller (csrc/step_controller.cpp), without a GPU: the closed loop is driven by the
oracle as planner for a few steps, then pdmpc_controller_build_step is timed on the reached traffic state (the build does not advance the
state, so it can be repeated).  usage: python tools/dbg_build_cost.py c3|c4 [steps]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "p-dmpc_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from oracle import oracle
from pdmpc.config import Config, ScenarioType
from pdmpc.mpa import get_mpa
from pdmpc.native_controller import NativeController
from pdmpc.road_network import commonroad_scenario

w = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n, hp, cl = {"c3": (128, 8, 2), "c4": (512, 10, 99), "c2": (20, 8, 99), "c5": (20, 8, 99)}[w]
options = Config(scenario_type=ScenarioType.commonroad, amount=n, Hp=hp, max_vehicles=max(n, 32), max_nodes=1 << 22, max_num_CLs=cl)
mpa = get_mpa(options)
nat = NativeController(options, commonroad_scenario(options, seed=1, tiles=max(1, (n + 19) // 20)), mpa, None, coupling="distance",
                       priority_strategy="constant" if w in ("c2", "c5") else "coloring")
for k in range(steps):
    nat.build_step()
    recs, _ = oracle.plan_step(options, mpa, nat.problem())
    nat.apply(recs)
if w == "c5":  # the explorative step's 64 prioritizations (the build advances the step counter only)
    best = 1e9
    for _ in range(8):
        t0 = time.perf_counter()
        for _ in range(50):
            nat.explore_build(64, 7)
        best = min(best, (time.perf_counter() - t0) / 50 * 1e3)
    print("c5: explore_build(64) %.3f ms" % best)
    sys.exit(0)
reps, best = 100, 1e9
for _ in range(8):
    t0 = time.perf_counter()
    for _ in range(reps):
        nat.build_step()
    best = min(best, (time.perf_counter() - t0) / reps * 1e3)
ta = 1e9
recs = np.ascontiguousarray(recs)
for _ in range(8):
    t0 = time.perf_counter()
    for _ in range(reps):
        nat.apply(recs)
    ta = min(ta, (time.perf_counter() - t0) / reps * 1e3)
print("%s: apply %.3f ms" % (w, ta))
print("%s: build_step %.3f ms (best of 8 batches of %d; n = %d, Hp = %d, after %d closed-loop steps)" % (w, best, reps, n, hp, steps))
