import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
os.environ["PDMPC_DEBUG_TAIL"] = "1"
import numpy as np
import bench
class A: pass
args = A(); args.vehicles = 20; args.hp = 8; args.mpa = "single_speed"; args.instances = 1; args.workload = "c2"; args.max_nodes = 1 << 17; args.seed = 1; args.max_levels = 99; args.priorities = "constant"
options, mpa, ctl = bench.build_world(args, 0)
from pdmpc.optimizer import GraphSearchHip
opt = GraphSearchHip(options); opt._ensure_mpa(mpa); h = opt.handle
probs = bench.record_steps(options, mpa, ctl, opt, 20, 20)
tot_nodes = tot_proc = tot_pop = tot_exp = 0
for b, prob in enumerate(probs):
    fb = [f if f is not None else [] for f in prob["fallback"]]
    h.pack_step(prob["iters"], prob["preds"], fb)
    h.launch(); recs = h.fetch(len(prob["iters"]))
    for v in range(len(recs)):
        t = np.asarray(recs[v]["path_nodes"])
        tot_nodes += int(t[16][2]); tot_proc += int(t[16][1]); tot_pop += int(recs[v]["n_popped"]); tot_exp += int(recs[v]["n_expanded"])
n = len(probs)
print("per step: nodes created %.0f, processed %.0f, reference pops %.0f, reference tree size %.0f" % (tot_nodes / n, tot_proc / n, tot_pop / n, tot_exp / n))
