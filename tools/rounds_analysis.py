"""Offline (CPU, oracle as planner): how many rounds a search needs under two selection policies, counted on the reference trees of C2's
steps 21-40 — rounds of the 24 smallest keys (what the kernel does) against "walk the greedy chain first, then everything below the
goal candidate's key per round" (tools/experiments/README.md).  python tools/rounds_analysis.py"""
import sys, copy
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd")]
import numpy as np
from oracle import oracle, packing
from pdmpc.config import Config, ScenarioType
from pdmpc.controller import PrioritizedSequentialController
from pdmpc.iteration_data import info_from_record
from pdmpc.mpa import get_mpa
from pdmpc.road_network import boundary_provider, commonroad_scenario
n, hp = 20, 8
options = Config(scenario_type=ScenarioType.commonroad, amount=n, Hp=hp, max_vehicles=32, max_nodes=1 << 22)
mpa = get_mpa(options)
sc = commonroad_scenario(options, seed=1, tiles=1)
ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc), priority_strategy="constant")
mpa_struct, keep_m = packing.pack_mpa(mpa)
stats = []
def plan_step_traced(problem):
    Hp = options.Hp
    nn = len(problem["iters"])
    recs = packing.out_array(nn)
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
                    if fb is not None and len(fb): dyn.append([np.asarray(a, dtype=np.float64) for a in fb])
            it.dynamic_obstacle_area = dyn
            iters.append(it)
        arr, keep_v = packing.pack_vehicles(iters, Hp)
        out, traces, ms = oracle.plan_batch_raw(options, mpa_struct, arr, size, trace=record, trace_capacity=1 << 18)
        for q, s in enumerate(slots):
            recs[s] = out[q]
            if int(out[q]["status"]) != 0:
                fb = problem["fallback"][s]
                if fb is not None and len(fb):
                    for k in range(Hp):
                        a = np.asarray(fb[k], dtype=np.float64)
                        recs[s]["shape_cols"][k] = a.shape[1]; recs[s]["shapes"][k][:, : a.shape[1]] = a
            if record: analyse(traces[q], out[q])
        first += size
    return [info_from_record(recs[i], options.Hp) for i in range(nn)]
def analyse(tr, rec):
    T = tr.tree; pops = np.asarray(tr.pops)  # 1-based node ids in pop order
    nn = len(T["k"])
    if nn == 0 or int(rec["status"]) != 0: return
    key = T["g"] + T["h"]; par = T["parent"]; k = T["k"]
    popped = np.zeros(nn + 1, bool); popped[pops] = True
    children = [[] for _ in range(nn + 1)]
    for i in range(1, nn + 1):
        p = int(par[i - 1])
        if p: children[p].append(i)
    goal = int(pops[-1])
    # greedy chain from the root (node 1): the smallest-key child at every level, as long as it is a popped node
    chain = [1]; cur = 1
    while k[cur - 1] < hp and children[cur]:
        c = min(children[cur], key=lambda j: (key[j - 1], j))
        if not popped[c]: break
        chain.append(c); cur = c
    on_chain = set(chain)
    # branch depth of every popped node off the chain
    bd = {}
    def branch_depth(i):
        if i in on_chain: return 0
        if i in bd: return bd[i]
        bd[i] = branch_depth(int(par[i - 1])) + 1
        return bd[i]
    sys.setrecursionlimit(10000)
    mb = max([branch_depth(int(i)) for i in pops] + [0])
    rounds_B = 1 + mb + (0 if goal in on_chain else 0)
    # policy A restricted to the popped set: rounds of the 24 smallest keys (growing by half of what has been processed)
    import heapq
    openh = [(key[0], 1)]; done = 0; rounds_A = 0
    while openh:
        size = max(24, done // 2)
        batch = [heapq.heappop(openh) for _ in range(min(size, len(openh)))]
        rounds_A += 1; done += len(batch)
        for _, i in batch:
            for c in children[i]:
                if popped[c]: heapq.heappush(openh, (key[c - 1], c))
    stats.append((len(pops), rounds_A, rounds_B, len(chain) - 1, goal in on_chain))
record = False
for t in range(40):
    record = t >= 20
    ctl.step(plan_step=plan_step_traced)
S = np.array([(a, b, c, d, int(e)) for a, b, c, d, e in stats])
print("searches", len(S), "pops median", np.median(S[:, 0]), "rounds A (24 smallest, restricted to the popped set) mean %.1f median %d" % (S[:, 1].mean(), np.median(S[:, 1])),
      "| rounds B (chain probe + everything below the bound) mean %.1f median %d" % (S[:, 2].mean(), np.median(S[:, 2])), "| greedy chain reaches depth: mean %.1f, is the goal's path in %d%%" % (S[:, 3].mean(), 100 * S[:, 4].mean()))
for lo, hi in ((0, 300), (300, 2000), (2000, 10 ** 9)):
    m = (S[:, 0] >= lo) & (S[:, 0] < hi)
    if m.any(): print("  pops in [%d, %d): %d searches, rounds A %.1f, rounds B %.1f, chain depth %.1f" % (lo, hi, m.sum(), S[m, 1].mean(), S[m, 2].mean(), S[m, 3].mean()))
