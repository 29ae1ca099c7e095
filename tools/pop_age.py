"""Diagnostic: how old (in generated nodes) is a node when it is popped? (closed loop C2 with the oracle)"""
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
ages = []  # (pops of the search, age array)
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
            par = np.asarray(t.tree["parent"])
            nn = len(par)
            nchild = np.bincount(par, minlength=nn + 1)  # children of node id (1-based)
            pops = np.asarray(t.pops, dtype=np.int64)
            gen_before = 1 + np.concatenate([[0], np.cumsum(nchild[pops])[:-1]])
            ages.append((len(pops), gen_before - pops))
        first += size
    return recs
from pdmpc.iteration_data import info_from_record
def ps(prob):
    recs = plan_step(prob)
    return [info_from_record(recs[i], Hp) for i in range(len(recs))]
for k in range(a.steps):
    ctl.step(plan_step=ps)
def report(sel, name):
    al = np.concatenate([x for n, x in ages if sel(n)])
    print(name, "pops", len(al), " ".join("<%d: %.3f" % (K, np.mean(al < K)) for K in (8, 16, 32, 64, 128, 256, 1024)))
report(lambda n: True, "all searches")
report(lambda n: n >= 1000, "searches with >= 1000 pops")
report(lambda n: n < 1000, "searches with < 1000 pops")
