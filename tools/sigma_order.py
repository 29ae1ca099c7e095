"""Diagnostic / theory check (CPU, oracle traces of the closed-loop C2 workload).

Claim behind the frontier kernel: with pairwise distinct keys, a best-first search pops node X
before node Y  iff  X is an ancestor of Y, or  max key on the path (LCA, X]  <  max key on the path (LCA, Y].
Consequences checked here against the oracle's pop sequence and tree:
  * the pop sequence is the tree's nodes sorted by that order,
  * n_popped, n_expanded and the ids along tree_path follow from counting nodes relative to the goal path,
  * how far a threshold exploration (all nodes whose path maximum is <= L) overshoots the reference's pops.
"""
import argparse
import copy
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
ap.add_argument("--steps", type=int, default=25)
a = ap.parse_args()


class A:
    pass


args = A()
args.vehicles = 20
args.hp = 8
args.mpa = "single_speed"
args.instances = 1
args.workload = "c2"
args.max_nodes = 1 << 17
args.seed = 1
options, mpa, ctl = bench.build_world(args, 0)
mpa_struct, keep = abi.pack_mpa(mpa)
Hp = options.Hp
stats = []


def check_trace(t, rec):
    par = np.asarray(t.tree["parent"], dtype=np.int64)  # 1-based, 0 for the root
    f = np.asarray(t.tree["g"]) + np.asarray(t.tree["h"])
    kk = np.asarray(t.tree["k"])
    nn = len(par)
    pops = np.asarray(t.pops, dtype=np.int64)  # 1-based ids in pop order
    popped = np.zeros(nn + 1, dtype=bool)
    popped[pops] = True
    ok = int(rec["status"]) == 0
    # pairwise order check on consecutive pops: X before Y
    def path(x):
        p = []
        while x:
            p.append(x)
            x = par[x - 1]
        return p[::-1]

    bad = 0
    for i in range(len(pops) - 1):
        px, py = path(pops[i]), path(pops[i + 1])
        c = 0
        while c < len(px) and c < len(py) and px[c] == py[c]:
            c += 1
        if c == len(px):
            continue  # X is an ancestor of Y
        if c == len(py):
            bad += 1
            continue
        bx = max(f[q - 1] for q in px[c:])
        by = max(f[q - 1] for q in py[c:])
        if not bx < by:
            bad += 1
    res = {"pops": len(pops), "nodes": nn, "order_violations": bad}
    if ok:
        goal = int(pops[-1])
        P = path(goal)
        assert len(P) == Hp + 1
        on_path = {q: j for j, q in enumerate(P)}
        Mp = np.zeros((Hp + 1, Hp + 2))  # Mp[d][j] = max key of P_{d+1..j}
        for d in range(Hp + 1):
            m = -1.0
            for j in range(d + 1, Hp + 1):
                m = max(m, f[P[j] - 1])
                Mp[d][j] = m
        # (d, bx) per node, t = number of path nodes popped before it
        d_of = np.zeros(nn + 1, dtype=np.int64)
        bx_of = np.zeros(nn + 1)
        t_of = np.zeros(nn + 1, dtype=np.int64)
        ties = 0
        for x in range(1, nn + 1):
            if x in on_path:
                d_of[x] = on_path[x]
                bx_of[x] = -1.0
                t_of[x] = on_path[x]  # P_j: j path nodes before it
                continue
            p = par[x - 1]
            if p in on_path:
                d_of[x] = on_path[p]
                bx_of[x] = f[x - 1]
            else:
                d_of[x] = d_of[p]
                bx_of[x] = max(bx_of[p], f[x - 1])
            d = d_of[x]
            t = Hp + 1
            for j in range(d + 1, Hp + 1):
                if bx_of[x] == Mp[d][j]:
                    ties += 1
                if bx_of[x] < Mp[d][j]:
                    t = j
                    break
            t_of[x] = t
        # the reference's tree only holds nodes whose parent was popped & valid; popped set = t <= Hp (or the goal)
        pred_popped = np.array([x for x in range(1, nn + 1) if (x in on_path) or t_of[x] <= Hp])
        res["popset_equal"] = set(pred_popped.tolist()) == set(pops.tolist())
        nchild = np.bincount(par, minlength=nn + 1)
        # S_j = tree size when P_j is popped = 1 + children of nodes popped before P_j
        ids = [1]
        for j in range(Hp):
            before = [x for x in pred_popped if (x in on_path and on_path[x] < j) or (x not in on_path and t_of[x] <= j)]
            S = 1 + int(sum(nchild[x] for x in before))
            sib = [c for c in range(1, nn + 1) if par[c - 1] == P[j]]
            ids.append(S + 1 + sib.index(P[j + 1]))
        res["ids_equal"] = ids == [int(v) for v in rec["tree_path"][: Hp + 1]] and ids == P
        res["ties"] = ties
        # threshold exploration: everything with path maximum <= B1(goal) is explored at least; the generated-but-unpopped
        # nodes (open list at the end) with key <= L would be processed too
        Bstar = max(f[q - 1] for q in P)
        open_nodes = np.array([x for x in range(1, nn + 1) if not popped[x]], dtype=np.int64)
        res["open"] = len(open_nodes)
        res["open_le_Bstar"] = int(np.sum(f[open_nodes - 1] <= Bstar)) if len(open_nodes) else 0
        res["Bstar"] = Bstar
        res["fgoal"] = f[goal - 1]
    return res


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
            if len(traces[q].pops) < 6000:
                stats.append(check_trace(traces[q], out[q]))
        first += size
    return recs


def ps(prob):
    recs = plan_step(prob)
    return [info_from_record(recs[i], Hp) for i in range(len(recs))]


for k in range(a.steps):
    ctl.step(plan_step=ps)
print("searches checked:", len(stats))
print("order violations:", sum(s["order_violations"] for s in stats))
ok = [s for s in stats if "ids_equal" in s]
print("goal searches:", len(ok), "popset equal:", sum(s["popset_equal"] for s in ok), "ids equal:", sum(s["ids_equal"] for s in ok), "ties:", sum(s["ties"] for s in ok))
tot_p = sum(s["pops"] for s in ok)
print("pops %d, open at end %d, open nodes with key <= B1(goal): %d (%.1f %% of pops)" % (tot_p, sum(s["open"] for s in ok), sum(s["open_le_Bstar"] for s in ok), 100.0 * sum(s["open_le_Bstar"] for s in ok) / tot_p))
big = sorted(ok, key=lambda s: -s["pops"])[:5]
for s in big:
    print(s)
