"""Diagnostic: C4 steps recorded in a closed loop, kept packed in HBM and launched again and again without the re-plan safety net
(what bench.py's timed loop does); prints the debug tail (flags, rounds, nodes) of every search that ends with an error status.
PDMPC_LIB selects a library variant."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
os.environ["PDMPC_DEBUG_TAIL"] = "1"
import numpy as np
if os.environ.get("DIAG_TORCH"):
    import torch
    torch.zeros(1, device="cuda")  # (bench.py runs with torch's CUDA context up: its own streams and allocations)
    torch.cuda.synchronize()
from pdmpc.config import Config, ScenarioType
from pdmpc.controller import PrioritizedSequentialController
from pdmpc.iteration_data import info_from_record
from pdmpc.mpa import get_mpa
from pdmpc.optimizer import GraphSearchHip
from pdmpc.road_network import boundary_provider, commonroad_scenario

options = Config(scenario_type=ScenarioType.commonroad, amount=512, Hp=10, max_vehicles=512, max_nodes=1 << 16)
sc = commonroad_scenario(options, seed=1, tiles=26)
mpa = get_mpa(options)
opt = GraphSearchHip(options); opt._ensure_mpa(mpa); h = opt.handle
ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc), priority_strategy="coloring")
problems = []
def plan_step(prob):
    problems.append(prob)
    fb = [f if f is not None else [] for f in prob["fallback"]]
    recs = h.plan_step(prob["iters"], prob["preds"], fb)
    return [info_from_record(recs[i], options.Hp) for i in range(len(recs))]
for _ in range(int(os.environ.get("DIAG_STEPS", "12"))):
    ctl.step(plan_step=plan_step)
print("recorded; safe_replans", h.stats()["safe_replans"], flush=True)
banks = problems[-8:]
h.allow_overflow = True
for b, prob in enumerate(banks):
    h.select_bank(b)
    h.pack_step(prob["iters"], prob["preds"], [f if f is not None else [] for f in prob["fallback"]])
    while True:
        h.launch()
        recs = h.fetch(len(prob["iters"]))
        if not (recs["status"] == 2).any():
            break
        h.grow_arena(2 * h.arena_nodes()[0])
        print("arena grown to", h.arena_nodes()[0], flush=True)
n_bad = 0
for rep in range(int(os.environ.get("DIAG_REPS", "12"))):
    for b, prob in enumerate(banks):
        h.select_bank(b)
        h.launch()
        recs = h.fetch(len(prob["iters"]))
        bad = [s for s in range(len(recs)) if int(recs[s]["status"]) not in (0, 1)]
        for s in bad[:8]:
            t = np.asarray(recs[s]["path_nodes"])
            print("BAD rep", rep, "bank", b, "slot", s, "level", prob["levels"][s], "preds", prob["preds"][s][:6], "status", int(recs[s]["status"]), "flags", hex(int(t[16][5])), "rounds", t[16][0], "nodes", t[16][2], "tail", [float(x) for x in t[16][:12]], flush=True)
        n_bad += len(bad)
        if bad and n_bad > 24:
            print("stopping", flush=True); os._exit(1)
print("done; bad plans", n_bad, "helper guard hits", h.stats().get("tie_fallbacks"), flush=True)
