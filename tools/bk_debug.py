"""Diagnostic: the bulk kernel against the oracle on one batch of independent InterX problems, field by field (first GPU contact of a kernel change)."""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "p-dmpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import problems  # noqa: E402
from oracle import oracle  # noqa: E402
from pdmpc.backend import Handle  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    Hp = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    options, mpa, iters = problems.problem_set("interx", seed, n, Hp=Hp)
    options.trace_pops = 0
    options.max_nodes = 1 << 15
    options.max_vehicles = n
    h = Handle(options)
    h.upload_mpa(mpa)
    gpu = h.plan_batch(iters)
    st = h.stats()
    print("kernel", st["kernel"], "kernel_ms", st["kernel_ms"], "rounds", st["rounds"], "processed", st["nodes_processed"], "popped", st["nodes_popped"])
    unb = copy.copy(options)
    unb.max_nodes = 1 << 30
    _, ref, _ = oracle.plan_batch(unb, mpa, iters, trace=False)
    bad = 0
    for v in range(n):
        diffs = []
        for name in gpu.dtype.names:
            a, b = gpu[name][v], ref[name][v]
            if a.dtype.kind == "f":
                same = np.all(np.asarray(a).view(np.uint64) == np.asarray(b).view(np.uint64))
            else:
                same = np.all(a == b)
            if not same:
                diffs.append(name)
        if diffs:
            bad += 1
            print("veh", v, "differs in", diffs)
            print("   gpu status", gpu["status"][v], "n_exp", gpu["n_expanded"][v], "n_pop", gpu["n_popped"][v], "path", gpu["tree_path"][v][: Hp + 1], "trims", gpu["predicted_trims"][v][:Hp])
            print("   ref status", ref["status"][v], "n_exp", ref["n_expanded"][v], "n_pop", ref["n_popped"][v], "path", ref["tree_path"][v][: Hp + 1], "trims", ref["predicted_trims"][v][:Hp])
    print("vehicles", n, "mismatching", bad)
    h.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
