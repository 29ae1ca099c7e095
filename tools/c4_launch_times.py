"""Diagnostic: kernel time and LDS layout of every recorded C4 step (finds the launches that stall)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
if os.environ.get("C4_TAIL"): os.environ["PDMPC_DEBUG_TAIL"] = "1"
import numpy as np
import bench
class A: pass
args = A(); args.vehicles = 512; args.hp = 10; args.mpa = "single_speed"; args.instances = 1; args.workload = "c4"; args.max_nodes = 1 << 16; args.seed = 1; args.max_levels = 99; args.priorities = "coloring"
options, mpa, ctl = bench.build_world(args, 0)
from pdmpc.optimizer import GraphSearchHip
opt = GraphSearchHip(options); opt._ensure_mpa(mpa); h = opt.handle
probs = bench.record_steps(options, mpa, ctl, opt, 4, 8)
print("arena", h.arena_nodes())
for rep in range(3):
    for b, prob in enumerate(probs):
        fb = [f if f is not None else [] for f in prob["fallback"]]
        h.pack_step(prob["iters"], prob["preds"], fb)
        h.allow_overflow = True
        t0 = time.time(); h.launch(); recs = h.fetch(len(prob["iters"])); dt = time.time() - t0
        bad = np.flatnonzero(recs["status"] < 0)
        for v in bad[:6]:
            t = np.asarray(recs[v]["path_nodes"])
            print("   slot", v, "level", prob["levels"][v], "preds", prob["preds"][v][:8], "tail", list(t[16][:8]), "phase ticks", list(t[15][:5]))
        st = h.stats()
        print("rep", rep, "step", b, "wall ms %.1f kernel ms %.1f launches %d lds %d err %d" % (1e3 * dt, st["kernel_ms"], st["n_launches"], st["lds_bytes"], int((recs["status"] < 0).sum())), flush=True)
