"""Per recorded C2 step: kernel time (min / median / max over repeats) and the nodes helper workgroups checked."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import bench
class A: pass
args = A(); args.vehicles = 20; args.hp = 8; args.mpa = "single_speed"; args.instances = 1; args.workload = "c2"; args.max_nodes = 1 << 17; args.seed = 1; args.max_levels = 99; args.priorities = "constant"
options, mpa, ctl = bench.build_world(args, 0)
from pdmpc.optimizer import GraphSearchHip
opt = GraphSearchHip(options); opt._ensure_mpa(mpa); h = opt.handle
probs = bench.record_steps(options, mpa, ctl, opt, 20, 20)
rows = []
prev = 0
for it in range(300):
    prob = probs[it % len(probs)]
    fb = [f if f is not None else [] for f in prob["fallback"]]
    h.pack_step(prob["iters"], prob["preds"], fb)
    t0 = time.perf_counter()
    h.launch(); recs = h.fetch(len(prob["iters"]))
    wall = (time.perf_counter() - t0) * 1e3
    st = h.stats()
    rows.append((it % len(probs), st["kernel_ms"], wall, st["helper_checked"] - prev))
    prev = st["helper_checked"]
rows = rows[20:]
by = {}
for b, k, w, hc in rows: by.setdefault(b, []).append((k, hc))
for b in sorted(by):
    ks = np.array([x[0] for x in by[b]]); hs = np.array([x[1] for x in by[b]])
    print("step %2d kernel ms min %.2f med %.2f max %.2f | helper-checked min %d med %d max %d" % (b, ks.min(), np.median(ks), ks.max(), hs.min(), np.median(hs), hs.max()))
