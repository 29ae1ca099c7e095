"""Diagnostic: N copies of a few independent searches (no predecessors) in one launch; every copy must give the record of the first.
usage: dbg_coresident.py [copies] [distinct]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import problems
from pdmpc.backend import Handle
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 600
distinct = int(sys.argv[2]) if len(sys.argv) > 2 else 4
options, mpa, iters = problems.problem_set("interx", 11, distinct, Hp=8)
options.max_vehicles = copies
options.max_nodes = 1 << 15
h = Handle(options)
h.upload_mpa(mpa)
batch = [iters[i % distinct] for i in range(copies)]
for rep in range(3):
    recs = h.plan_step(batch, [[] for _ in batch], None)
    bad = 0
    for i in range(copies):
        a, b = recs[i], recs[i % distinct]
        for name in ("status", "n_expanded", "n_popped", "tree_path", "predicted_trims"):
            if np.any(np.asarray(a[name]) != np.asarray(b[name])):
                if bad < 6:
                    print("rep", rep, "slot", i, "copy of", i % distinct, "field", name, "got", a[name] if np.ndim(a[name]) == 0 else list(a[name])[:6], "want", b[name] if np.ndim(b[name]) == 0 else list(b[name])[:6], flush=True)
                bad += 1
                break
    print("rep", rep, "copies", copies, "bad", bad, "n_expanded of the originals", [int(recs[i]["n_expanded"]) for i in range(distinct)], flush=True)
h.close()
