"""How sensitive are the selected plans to the last bit of MATLAB's built-ins?  

north_star asks for trajectories "within 1e-6" of the MATLAB optimizer.  The backend is bit-identical to the oracle; the
oracle reads MATLAB's closed-source cos / sin / norm / vecnorm / matrix products as single IEEE operations in source order
(expand_node.m:50-51,61,71; intersect_sat.m:23-32; GraphSearch.m:155-159).  Real MATLAB may differ from that reading in
the last bits.  This script plans the recorded closed-loop steps of a BASELINE config again with three other plausible
readings and reports, per variant, how many plans change and by how much:

    libm    glibc sin / cos instead of include/pdmpc_math.h (differs by <= 1 ulp, tests/test_math.py)
    hypot   hypot(dx, dy) for norm / vecnorm instead of sqrt(dx^2 + dy^2)
    fma     every a*b + c contracted to a fused multiply-add (what an optimised BLAS / vectorised kernel would do)

Every plan is computed on IDENTICAL inputs (the baseline's predecessor areas), so a changed plan is a near-tie that flipped,
not a propagated difference.  CPU only, test infrastructure only.
"""
import argparse
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

from oracle import oracle  # noqa: E402
from pdmpc import abi  # noqa: E402
from pdmpc.iteration_data import info_from_record  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c2", choices=["c2", "c3"])
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--out", default=None)
a = ap.parse_args()


class A:
    pass


args = A()
args.vehicles, args.hp = {"c2": (20, 8), "c3": (128, 8)}[a.workload]
args.mpa = "single_speed"
args.instances = 1
args.workload = a.workload
args.max_nodes = 1 << 24
args.seed = a.seed
args.max_levels = 99 if a.workload == "c2" else 2
args.priorities = "constant" if a.workload == "c2" else "coloring"
options, mpa, ctl = bench.build_world(args, 0)
mpa_struct, keep = abi.pack_mpa(mpa)
Hp = options.Hp
variants = ["libm", "hypot", "fma"]
tot = {v: {"plans": 0, "path_changed": 0, "n_expanded_changed": 0, "status_changed": 0, "max_dy": 0.0, "max_dy_same_path": 0.0, "bits_changed": 0} for v in variants}


def level_iters(problem, recs, slots):
    iters = []
    for s in slots:
        it = copy.copy(problem["iters"][s])
        dyn = list(it.dynamic_obstacle_area)
        for p in problem["preds"][s]:
            if int(recs[p]["status"]) == 0:
                dyn.append([np.array(recs[p]["shapes"][k][:, : int(recs[p]["shape_cols"][k])]) for k in range(Hp)])
            else:
                fb = problem["fallback"][p]
                if fb is not None and len(fb):
                    dyn.append([np.asarray(x, dtype=np.float64) for x in fb])
        it.dynamic_obstacle_area = dyn
        iters.append(it)
    return iters


def plan_step(problem):
    base, _ = oracle.plan_step(options, mpa, problem, n_threads=8, mpa_struct=mpa_struct)
    first = 0
    for size in problem["level_sizes"]:
        slots = list(range(first, first + size))
        iters = level_iters(problem, base, slots)  # the baseline's predecessor areas for every variant
        arr, keep_v = abi.pack_vehicles(iters, Hp)
        for v in variants:
            out, _, _ = oracle.plan_batch_raw(options, mpa_struct, arr, size, n_threads=8, variant=v)
            t = tot[v]
            for q, s in enumerate(slots):
                b, o = base[s], out[q]
                t["plans"] += 1
                if int(b["status"]) != int(o["status"]):
                    t["status_changed"] += 1
                    continue
                if int(b["status"]) != 0:
                    continue
                same_path = np.array_equal(b["predicted_trims"][:Hp], o["predicted_trims"][:Hp])
                t["path_changed"] += 0 if same_path else 1
                t["n_expanded_changed"] += int(b["n_expanded"]) != int(o["n_expanded"]) or not np.array_equal(b["tree_path"], o["tree_path"])
                dy = float(np.max(np.abs(np.asarray(b["y_predicted"][:Hp]) - np.asarray(o["y_predicted"][:Hp]))))
                t["max_dy"] = max(t["max_dy"], dy)
                if same_path:
                    t["max_dy_same_path"] = max(t["max_dy_same_path"], dy)
                t["bits_changed"] += 0 if np.array_equal(np.asarray(b["y_predicted"][:Hp]).view(np.uint64), np.asarray(o["y_predicted"][:Hp]).view(np.uint64)) else 1
        del keep_v
        first += size
    return [info_from_record(base[i], Hp) for i in range(len(base))]


for k in range(a.steps):
    ctl.step(plan_step=plan_step)
print("workload %s, %d closed-loop steps, %d plans per variant" % (a.workload, a.steps, tot["libm"]["plans"]))
print("%-6s %8s %14s %20s %16s %12s %18s %12s" % ("", "plans", "path changed", "ids/tree changed", "status changed", "max |dy|", "max |dy| same path", "bits differ"))
for v in variants:
    t = tot[v]
    print("%-6s %8d %14d %20d %16d %12.3e %18.3e %12d" % (v, t["plans"], t["path_changed"], t["n_expanded_changed"], t["status_changed"], t["max_dy"], t["max_dy_same_path"], t["bits_changed"]))
if a.out:
    json.dump({"workload": a.workload, "steps": a.steps, "variants": tot}, open(a.out, "w"), indent=1)
