"""Diagnostic (profile build): per-vehicle timing inside one-launch time steps of the C2 workload."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd")]
import numpy as np
from pdmpc import backend
backend.LIB_PATH = os.path.join(ROOT, "p-dmpc_amd", "csrc", "libpdmpc_hip_prof.so")
from pdmpc.config import Config, ScenarioType
from pdmpc.controller import PrioritizedSequentialController
from pdmpc.mpa import get_mpa
from pdmpc.optimizer import GraphSearchHip
from pdmpc.road_network import boundary_provider, commonroad_scenario
from pdmpc.iteration_data import info_from_record

o = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=8, max_vehicles=32, max_nodes=1 << 17)
mpa = get_mpa(o)
sc = commonroad_scenario(o, seed=1)
opt = GraphSearchHip(o)
opt._ensure_mpa(mpa)
ctl = PrioritizedSequentialController(o, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc))
rows = []
def plan_step(prob):
    n = len(prob["iters"])
    fb = [f if f is not None else [] for f in prob["fallback"]]
    opt.handle.pack_step(prob["iters"], prob["preds"], fb)
    opt.handle.launch()
    rec = opt.handle.fetch(n)
    t = rec["path_nodes"][:, 16, :4]
    rows.append((rec["n_popped"].copy(), rec["n_expanded"].copy(), t.copy(), opt.handle.stats()["kernel_ms"], [len(p) for p in prob["preds"]]))
    return [info_from_record(rec[i], o.Hp) for i in range(n)]
for k in range(32):
    ctl.step(plan_step=plan_step)
tot_search = tot_pops = 0
for k, (pops, nodes, t, ms, npred) in enumerate(rows[20:]):
    t0 = t[:, 2].min()
    end = (t[:, 3] - t0) / 100.0  # us
    start_search = (t[:, 2] - t0 + t[:, 0]) / 100.0
    search_us = t[:, 1] / 100.0
    order = np.argsort(end)
    last = order[-1]
    print("step %d kernel %.2f ms; slowest-finishing slot %d: search %.0f us for %d pops (%.2f us/pop), started at %.0f us" % (k, ms, last, search_us[last], pops[last], search_us[last] / pops[last], start_search[last]))
    heavy = np.argsort(-search_us)[:4]
    print("   heaviest:", [(int(i), int(pops[i]), int(nodes[i]), round(search_us[i]), round(search_us[i] / pops[i], 2)) for i in heavy], "sum search %.0f us" % search_us.sum())
    tot_search += search_us.sum(); tot_pops += pops.sum()
print("overall us per pop (sum of search time / pops):", tot_search / tot_pops)
