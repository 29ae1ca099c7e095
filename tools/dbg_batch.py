"""Diagnostic: the triple-speed batch of tests/test_gpu_parity.py through the library given by PDMPC_LIB, with the kernel's own counters."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
os.environ["PDMPC_DEBUG_TAIL"] = "1"
import numpy as np
import problems
from pdmpc.backend import Handle
from pdmpc.config import MpaType
options, mpa, iters = problems.problem_set("interx", 5, 20, Hp=8, mpa_type=MpaType.triple_speed)
options.max_nodes = 1 << 15
options.max_vehicles = 20
h = Handle(options); h.upload_mpa(mpa); h.allow_overflow = True
recs = h.plan_batch(iters)
for v in (8, 9, 10):
    t = np.asarray(recs[v]["path_nodes"])
    print(v, "status", int(recs[v]["status"]), "n_exp", int(recs[v]["n_expanded"]), "n_pop", int(recs[v]["n_popped"]), "path", list(recs[v]["tree_path"][:9]),
          "rounds", t[16][0], "processed", t[16][1], "raw nodes", t[16][2], "near", t[16][3], "far", t[16][4], "flags", t[16][5])
