"""Diagnostic: a C4-like closed loop (Hp 10, colouring levels) with as many vehicles as fit the chip at once, i.e. every search starts
with the launch and speculates on all its predecessors — against the oracle, every step.  usage: dbg_c4_resident.py [vehicles] [seed] [steps]"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from oracle import oracle
from pdmpc.config import Config, ScenarioType
from pdmpc.controller import PrioritizedSequentialController
from pdmpc.iteration_data import info_from_record
from pdmpc.mpa import get_mpa
from pdmpc.optimizer import GraphSearchHip
from pdmpc.road_network import boundary_provider, commonroad_scenario

n = int(sys.argv[1]) if len(sys.argv) > 1 else 240
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
options = Config(scenario_type=ScenarioType.commonroad, amount=n, Hp=10, max_vehicles=n, max_nodes=1 << 15)
sc = commonroad_scenario(options, seed=seed, tiles=(n + 19) // 20)
mpa = get_mpa(options)
opt = GraphSearchHip(options)
opt._ensure_mpa(mpa)
ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc), priority_strategy="coloring")
unbounded = copy.copy(options)
unbounded.max_nodes = 1 << 30
bad = [0]
def plan_step(prob):
    fb = [f if f is not None else [] for f in prob["fallback"]]
    gpu = opt.handle.plan_step(prob["iters"], prob["preds"], fb)
    ref, _ = oracle.plan_step(unbounded, mpa, prob, n_threads=os.cpu_count() or 1)
    for name in ("status", "n_expanded", "n_popped", "tree_path", "predicted_trims"):
        d = np.argwhere(np.asarray(gpu[name]) != np.asarray(ref[name]))
        if len(d):
            v = int(d[0][0])
            print("MISMATCH step", ctl.k, "field", name, "vehicle", v, "gpu", gpu[name][v], "ref", ref[name][v], "preds", len(prob["preds"][v]), "level", prob["levels"][v], flush=True)
            bad[0] += 1
            break
    return [info_from_record(ref[i], options.Hp) for i in range(len(ref))]
for _ in range(steps):
    ctl.step(plan_step=plan_step)
print("done", n, "vehicles seed", seed, "mismatching steps", bad[0], "safe_replans", opt.handle.stats()["safe_replans"])
