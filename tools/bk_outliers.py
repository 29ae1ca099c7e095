"""Diagnostic: wall time of every replayed C2 step (banks resident in HBM), several passes: which steps are slow, and are they the same ones every time?"""
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
for b, prob in enumerate(probs):
    h.select_bank(b)
    fb = [f if f is not None else [] for f in prob["fallback"]]
    h.pack_step(prob["iters"], prob["preds"], fb)
    h.launch(); h.fetch(len(prob["iters"]))
S = len(probs)
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = np.zeros((passes, S))
for p in range(passes):
    for b in range(S):
        h.select_bank(b)
        t0 = time.perf_counter()
        h.launch(); h.synchronize()
        T[p, b] = 1e3 * (time.perf_counter() - t0)
print("bank  median   min    max   (ms)")
for b in range(S):
    print("%3d  %6.3f %6.3f %6.3f" % (b, np.median(T[:, b]), T[:, b].min(), T[:, b].max()))
print("mean of medians %.3f ms -> %.1f steps/s; mean of all %.3f ms" % (np.median(T, axis=0).mean(), 1e3 / np.median(T, axis=0).mean(), T.mean()))
slow = np.argwhere(T > 3 * np.median(T, axis=0)[None, :])
print("outliers (pass, bank, ms):", [(int(p), int(b), round(float(T[p, b]), 2)) for p, b in slow][:20])
