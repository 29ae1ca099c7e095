"""Diagnostic: how often is the minimal key of the open list not unique at pop time? (closed loop C2 with the oracle)"""
import os, sys, copy, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from sortedcontainers import SortedList
from oracle import oracle
from pdmpc import abi
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
class A: pass
args = A(); args.vehicles=20; args.hp=8; args.mpa="single_speed"; args.instances=1; args.workload="c2"; args.max_nodes=1<<17; args.seed=1
options, mpa, ctl = bench.build_world(args, 0)
mpa_struct, keep = abi.pack_mpa(mpa)
Hp = options.Hp
stats = dict(searches=0, tied_searches=0, pops=0, tied_pops=0, tied_valid_pops=0, first_tie_frac=[])
def plan_step(problem):
    n = len(problem["iters"])
    recs = abi.out_array(n)
    first = 0
    for size in problem["level_sizes"]:
        slots = list(range(first, first + size))
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
        arr, keep_v = abi.pack_vehicles(iters, Hp)
        out, traces, ms = oracle.plan_batch_raw(options, mpa_struct, arr, size, n_threads=8, trace=True, trace_capacity=1 << 17)
        for q, s in enumerate(slots):
            recs[s] = out[q]
            if int(out[q]["status"]) != 0:
                fb = problem["fallback"][s]
                if fb is not None and len(fb):
                    for k in range(Hp):
                        x = np.asarray(fb[k], dtype=np.float64)
                        recs[s]["shape_cols"][k] = x.shape[1]
                        recs[s]["shapes"][k][:, : x.shape[1]] = x
            t = traces[q]
            f = t.tree["g"] + t.tree["h"]
            par = t.tree["parent"]
            nn = len(f)
            # children lists
            order = np.argsort(par, kind="stable")
            ps = par[order]
            start = np.searchsorted(ps, np.arange(1, nn + 2), side="left")
            open_ = SortedList([(f[0], 1)])
            tied = 0; first_tie = None
            for j, nd in enumerate(t.pops):
                k0 = open_[0][0]
                is_tie = len(open_) > 1 and open_[1][0] == k0
                has_children = start[nd] > start[nd - 1]
                if is_tie:
                    tied += 1
                    if first_tie is None: first_tie = j
                    if has_children: stats["tied_valid_pops"] += 1
                open_.remove((f[nd - 1], int(nd)))
                for c in order[start[nd - 1]:start[nd]]:
                    open_.add((f[c], int(c) + 1))
            stats["searches"] += 1; stats["pops"] += len(t.pops); stats["tied_pops"] += tied
            if tied:
                stats["tied_searches"] += 1
                stats["first_tie_frac"].append((first_tie, len(t.pops)))
        first += size
    return recs
from pdmpc.iteration_data import info_from_record
def ps(prob):
    recs = plan_step(prob)
    return [info_from_record(recs[i], Hp) for i in range(len(recs))]
for k in range(a.steps):
    ctl.step(plan_step=ps)
    print(k, {kk: v for kk, v in stats.items() if kk != "first_tie_frac"}, flush=True)
print(stats["first_tie_frac"][:50])
