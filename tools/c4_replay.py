"""Diagnostic: bench.py's replay of the recorded C4 steps (banks resident in HBM), printing the slow launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
os.environ["PDMPC_DEBUG_TAIL"] = "1"
import numpy as np
import bench
class A: pass
args = A(); args.vehicles = 512; args.hp = 10; args.mpa = "single_speed"; args.instances = 1; args.workload = "c4"; args.max_nodes = 1 << 16; args.seed = 1; args.max_levels = 99; args.priorities = "coloring"
options, mpa, ctl = bench.build_world(args, 0)
from pdmpc.optimizer import GraphSearchHip
opt = GraphSearchHip(options); opt._ensure_mpa(mpa); h = opt.handle
h.allow_overflow = True
probs = bench.record_steps(options, mpa, ctl, opt, 4, 8)
for b, prob in enumerate(probs):
    h.select_bank(b)
    fb = [f if f is not None else [] for f in prob["fallback"]]
    h.pack_step(prob["iters"], prob["preds"], fb)
    h.launch(); h.fetch(len(prob["iters"]))
lat = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    h.select_bank(i % len(probs))
    t0 = time.time(); h.launch(); h.synchronize(); dt = 1e3 * (time.time() - t0)
    lat.append(dt)
    if dt > 400:
        recs = h.fetch(len(probs[i % len(probs)]["iters"]))
        print("launch", i, "bank", i % len(probs), "ms %.0f" % dt, "error records", int((recs["status"] < 0).sum()), "slots", list(np.flatnonzero(recs["status"] < 0)[:10]), flush=True)
        tot = np.array([np.asarray(recs[v]["path_nodes"])[15][4] for v in range(len(recs))])
        for v in list(np.argsort(-tot)[:6]):
            t = np.asarray(recs[v]["path_nodes"])
            print("   slot", v, "tail rounds %d processed %d nodes %d near %d far %d flags %d" % tuple(int(x) for x in t[16][:6]), "ticks work %.0f arrival %.0f select %.0f wait %.0f total %.0f (x10us)" % tuple(t[15][:5] / 1000.0), "slowest node %d took %.0f us" % (int(t[15][6]), t[15][5] / 100.0), "preds", probs[i % len(probs)]["preds"][v][:6])
            if t[15][5] > 1e6:
                raw = h.raw_tree(int(v))
                nd = int(t[15][6]) - 1
                print("      node", nd + 1, "parent", raw["parent"][nd], "trim", raw["trim"][nd], "k", raw["k"][nd], "validity", raw["validity"][nd], "key", raw["key"][nd], "x", raw["x"][nd], "nodes", len(raw["x"]))
                p = raw["parent"][nd] - 1
                print("      its parent", p + 1, "parent", raw["parent"][p], "trim", raw["trim"][p], "k", raw["k"][p], "validity", raw["validity"][p])
print("mean %.1f max %.1f" % (np.mean(lat), np.max(lat)))
