"""Debug driver: a few small batches through the frontier kernel, compared with the oracle (prints instead of asserting)."""
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
options.trace_pops = 1 << 15
h = Handle(options)
h.allow_overflow = True
h.upload_mpa(mpa)
print("planning", count, "vehicles", flush=True)
gpu = h.plan_batch(iters)
print("planned", flush=True)
unb = copy.copy(options)
unb.max_nodes = 1 << 30
_, ref, traces = oracle.plan_batch(unb, mpa, iters, trace=True)
for v in range(count):
    same = all(np.array_equal(np.asarray(gpu[v][n]).view(np.uint8) if False else gpu[v][n], ref[v][n]) or (np.asarray(gpu[v][n]).dtype.kind == "f" and np.array_equal(np.asarray(gpu[v][n]).view(np.uint64), np.asarray(ref[v][n]).view(np.uint64))) for n in gpu.dtype.names)
    print(v, "status", gpu[v]["status"], ref[v]["status"], "n_exp", gpu[v]["n_expanded"], ref[v]["n_expanded"], "n_pop", gpu[v]["n_popped"], ref[v]["n_popped"],
          "path", list(gpu[v]["tree_path"][: Hp + 1]), list(ref[v]["tree_path"][: Hp + 1]), "SAME" if same else "DIFF", flush=True)
    if not same:
        for n in gpu.dtype.names:
            a, b = np.asarray(gpu[v][n]), np.asarray(ref[v][n])
            eq = np.array_equal(a.view(np.uint64), b.view(np.uint64)) if a.dtype.kind == "f" else np.array_equal(a, b)
            if not eq:
                print("   field", n, "differs")
    pops = h.pop_trace(v)
    ok = np.array_equal(pops, traces[v].pops[: len(pops)]) and len(pops) == len(traces[v].pops)
    tree = h.tree(v)
    okt = all(np.array_equal(tree[k].view(np.uint64), traces[v].tree[k].view(np.uint64)) for k in ("x", "y", "yaw", "g", "h")) and all(np.array_equal(tree[k], traces[v].tree[k]) for k in ("trim", "k", "parent"))
    print("   pops", len(pops), len(traces[v].pops), "OK" if ok else "DIFF", "tree", len(tree["x"]), len(traces[v].tree["x"]), "OK" if okt else "DIFF", flush=True)
print(h.stats())
for v in range(count):
    print("tail", v, list(np.asarray(gpu[v]["path_nodes"])[16][:6]))
