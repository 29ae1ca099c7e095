"""Debug driver: raw arena of the frontier kernel against the oracle's tree, node by node (matched by trim sequence)."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np

import problems
from oracle import oracle
from pdmpc.backend import Handle

mode = sys.argv[1] if len(sys.argv) > 1 else "interx"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 1
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
Hp = int(sys.argv[4]) if len(sys.argv) > 4 else 6
options, mpa, iters = problems.problem_set(mode, seed, count, Hp=Hp)
options.max_nodes = 1 << 15
options.max_vehicles = max(count, 1)
h = Handle(options)
h.allow_overflow = True
h.upload_mpa(mpa)
gpu = h.plan_batch(iters)
unb = copy.copy(options)
unb.max_nodes = 1 << 30
_, ref, traces = oracle.plan_batch(unb, mpa, iters, trace=True)
for v in range(count):
    raw = h.raw_tree(v)
    t = traces[v].tree
    n = len(raw["x"])
    print("vehicle", v, "raw nodes", n, "ref nodes", len(t["x"]), "status", gpu[v]["status"], ref[v]["status"], "tail", list(np.asarray(gpu[v]["path_nodes"])[16][:6]))
    # signature = tuple of trims from the root
    def sigs(parent, trim, one_based_parent=True):
        out = [None] * len(parent)
        for i in range(len(parent)):
            p = int(parent[i])
            out[i] = (int(trim[i]),) if p == 0 else out[p - 1] + (int(trim[i]),)
        return out
    bad_parent = [i for i in range(1, n) if not (0 < raw["parent"][i] <= i)]
    print("  raw nodes with a bad parent:", bad_parent[:10], len(bad_parent))
    if bad_parent:
        lo = max(bad_parent[0] - 12, 0)
        for i in range(lo, min(lo + 60, n)):
            print("   raw", i, "parent", raw["parent"][i], "trim", raw["trim"][i], "k", raw["k"][i], "x %.4f g %.5f key %.5f" % (raw["x"][i], raw["g"][i], raw["key"][i]), "val", raw["validity"][i])
        continue
    rs = sigs(raw["parent"], raw["trim"])
    os_ = sigs(t["parent"], t["trim"])
    rmap = {s: i for i, s in enumerate(rs)}
    print("  duplicate signatures in raw:", n - len(rmap))
    missing = [s for s in os_ if s not in rmap]
    print("  reference nodes missing in raw:", len(missing), missing[:3])
    popped = set(int(p) for p in traces[v].pops)
    # validity / keys of reference nodes
    nbad = 0
    for j, s in enumerate(os_):
        if s not in rmap:
            continue
        i = rmap[s]
        fk = t["g"][j] + t["h"][j]
        same = raw["g"][i] == t["g"][j] and raw["h"][i] == t["h"][j] and raw["x"][i] == t["x"][j] and (j == 0 or raw["key"][i] == fk)
        was_popped = (j + 1) in popped
        if not same or (was_popped and raw["validity"][i] == 0):
            nbad += 1
            if nbad < 8:
                print("   ref node", j + 1, "raw", i, "same", same, "popped", was_popped, "validity", raw["validity"][i], "key", raw["key"][i], fk)
    print("  reference nodes with wrong data or unprocessed although popped:", nbad)
    # validity of popped nodes: the reference expands a popped node iff its edge is valid: children exist
    nchild = np.bincount(t["parent"], minlength=len(t["x"]) + 1)
    wrongv = 0
    for p in traces[v].pops:
        s = os_[p - 1]
        if s in rmap:
            i = rmap[s]
            ref_valid = nchild[p] > 0 or (t["k"][p - 1] == Hp and p == traces[v].pops[-1] and ref[v]["status"] == 0)
            if t["k"][p - 1] < Hp and (raw["validity"][i] == 1) != bool(nchild[p] > 0):
                wrongv += 1
    print("  popped nodes whose verdict differs:", wrongv)
    goals = [i for i in range(n) if raw["k"][i] == Hp and raw["validity"][i] == 1]
    print("  valid horizon nodes in raw:", len(goals), "reference goal sig", os_[traces[v].pops[-1] - 1] if ref[v]["status"] == 0 else None)
    if ref[v]["status"] == 0:
        gs = os_[traces[v].pops[-1] - 1]
        print("  reference goal present in raw:", gs in rmap, "validity", raw["validity"][rmap[gs]] if gs in rmap else None, "key", raw["key"][rmap[gs]] if gs in rmap else None)
