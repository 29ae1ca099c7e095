"""Diagnostic: per recorded C2 step — kernel time, pops (total / heaviest vehicle), speculation arrivals, restarts and the
pops they threw away.  Usage: python tools/step_breakdown.py [--hp 8]"""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import bench
from pdmpc.optimizer import GraphSearchHip

ap = argparse.ArgumentParser()
ap.add_argument("--hp", type=int, default=8)
ap.add_argument("--record", type=int, default=20)
a = ap.parse_args()
class A: pass
args = A(); args.vehicles = 20; args.hp = a.hp; args.mpa = "single_speed"; args.instances = 1; args.workload = "c2"; args.max_nodes = 1 << 17; args.seed = 1
options, mpa, ctl = bench.build_world(args, 0)
opt = GraphSearchHip(options)
problems = bench.record_steps(options, mpa, ctl, opt, 20, a.record)
h = opt.handle
rows = []
for i, prob in enumerate(problems):
    n = len(prob["iters"])
    fb = [f if f is not None else [] for f in prob["fallback"]]
    h.pack_step(prob["iters"], prob["preds"], fb)
    ms = []
    for rep in range(3):
        h.reset_stats()
        h.launch()
        rec = h.fetch(n)
        st = h.stats()
        ms.append(st["kernel_ms"])
    pops = rec["n_popped"]
    rows.append((i, min(ms), int(pops.sum()), int(pops.max()), len(prob["level_sizes"]), st["speculation_arrivals"], st["speculation_restarts"], st["speculation_wasted_pops"]))
print("step  kernel_ms  pops_total  pops_max  levels  arrivals  restarts  wasted_pops  ms_per_1k_maxpops")
for r in rows:
    print("%4d  %9.3f  %10d  %8d  %6d  %8d  %8d  %11d  %8.2f" % (r + (1e3 * r[1] / max(r[3], 1),)))
print("mean kernel ms %.3f; sum over steps of heaviest-vehicle pops %d" % (np.mean([r[1] for r in rows]), sum(r[3] for r in rows)))
h.close()
