"""Diagnostic: the explorative step through the C ABI against the resident replay of the same steps: kernel time (HIP events) per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import bench
class A: pass
args = A(); args.vehicles = 20; args.hp = 8; args.mpa = "single_speed"; args.instances = 64; args.workload = "c5"; args.max_nodes = 1 << 15; args.seed = 1; args.max_levels = 99; args.priorities = "constant"
options, mpa, ctl = bench.build_world(args, 0)
from pdmpc.optimizer import GraphSearchHip
from pdmpc.native_controller import NativeController
from pdmpc.road_network import commonroad_scenario
opt = GraphSearchHip(options); opt._ensure_mpa(mpa); h = opt.handle
probs = bench.record_steps(options, mpa, ctl, opt, 20, 12, explore_instances=64)
res = []
for b, prob in enumerate(probs):
    h.select_bank(b)
    fb = [f if f is not None else [] for f in prob["fallback"]]
    h.pack_step(prob["iters"], prob["preds"], fb, weights=prob.get("prev_pops"))
    h.allow_overflow = True
    while True:
        h.launch(); r = h.fetch(len(prob["iters"]))
        if not (r["status"] == 2).any():
            break
        h.grow_arena(2 * h.arena_nodes()[0])
for b, prob in enumerate(probs):
    h.select_bank(b)
    h.reset_stats()
    h.launch(); h.synchronize()
    st = h.stats()
    res.append((st["kernel_ms"], st["nodes_processed"]))
h.select_bank(0)
nat = NativeController(options, commonroad_scenario(options, seed=1, tiles=1), mpa, h, coupling="distance", priority_strategy="constant")
nat.run(20)
nat.explore_follow_own(True)
for i in range(12):
    h.reset_stats()
    ms = nat.explore_run(64, 1)
    st = h.stats()
    t = nat.last_timing()
    print("step", 21 + i, "closed loop: total %.3f ms kernel %.3f ms nodes %d | resident replay of the same step: kernel %.3f ms nodes %d" % (ms[0], st["kernel_ms"], st["nodes_processed"], res[i][0], res[i][1]), flush=True)
