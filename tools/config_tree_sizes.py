"""Diagnostic (CPU, oracle): status counts and the largest search tree of a BASELINE config's closed loop with an
effectively unbounded arena — what bench.py / the tests must size `max_nodes` for."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

from oracle import oracle  # noqa: E402
from pdmpc import abi  # noqa: E402
from pdmpc.iteration_data import info_from_record  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c4")
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--max-levels", type=int, default=99)
ap.add_argument("--priorities", default="coloring")
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()


class A:
    pass


args = A()
args.vehicles, args.hp = {"c2": (20, 8), "c3": (128, 8), "c4": (512, 10)}[a.workload]
args.mpa = "single_speed"
args.instances = 1
args.workload = a.workload
args.max_nodes = 1 << 24
args.seed = a.seed
args.max_levels = a.max_levels
args.priorities = a.priorities
options, mpa, ctl = bench.build_world(args, 0)
mpa_struct, keep = abi.pack_mpa(mpa)
Hp = options.Hp
for k in range(a.steps):
    t0 = time.time()

    def ps(prob):
        recs, ms = oracle.plan_step(options, mpa, prob, n_threads=8, mpa_struct=mpa_struct)
        st = np.array([int(r["status"]) for r in recs])
        ne = np.array([int(r["n_expanded"]) for r in recs])
        npop = np.array([int(r["n_popped"]) for r in recs])
        print("step %d levels %d ok %d exhausted %d other %d max tree %d (p99 %d) pops %d max pops %d  %.1f s" % (
            k + 1, len(prob["level_sizes"]), int((st == 0).sum()), int((st == 1).sum()), int((st > 1).sum() + (st < 0).sum()), ne.max(), int(np.percentile(ne, 99)), npop.sum(), npop.max(), time.time() - t0), flush=True)
        return [info_from_record(recs[i], Hp) for i in range(len(recs))]

    ctl.step(plan_step=ps)
