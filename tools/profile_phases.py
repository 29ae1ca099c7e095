"""Diagnostic: per-phase cycle shares of the search kernel (build with -DPDMPC_PROFILE -> libpdmpc_hip_prof.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from pdmpc import backend
backend.LIB_PATH = os.environ.get("PDMPC_PROF_LIB", os.path.join(ROOT, "p-dmpc_amd", "csrc", "libpdmpc_hip_prof.so"))
from pdmpc.backend import Handle
import problems

names_heap = ["heap_pop", "-", "validity(cache or check)", "node-load", "sincos", "children", "pushes", "loop", "-", "-", "-", "-", "-", "cache-misses(count)", "-", "-"]
names_bm = ["Q loop head", "Q validity lookup", "Q remove", "Q find", "Q wait for expander", "Q children visible", "-", "Q-only invalid pops (count)", "E idle", "E validity", "E expansion+reply", "-", "-", "-", "E nodes (count)", "E self-validated (count)"]
names = names_heap if os.environ.get("PDMPC_QUEUE") == "0" else names_bm
for mode, seed, hp in (("interx", 1, 8), ("interx", 2, 8)):
    options, mpa, iters = problems.problem_set(mode, seed, 24, Hp=hp)
    options.max_vehicles = 32
    options.max_nodes = 1 << 17
    h = Handle(options)
    h.upload_mpa(mpa)
    rec = h.plan_batch(iters)
    prof = rec["shapes"][:, 15, :, :].reshape(len(rec), 16)
    pops = rec["n_popped"].astype(float)
    if names is names_heap:
        rounds = prof[:, 13].copy(); prof[:, 13] = 0
        tot = prof.sum(axis=1)
        print(mode, seed, "pops", int(pops.sum()), "cycles/pop (s_memtime ticks)", tot.sum() / pops.sum())
        for i, nm in enumerate(names):
            print("  %-14s %8.1f ticks/pop  %5.1f %%" % (nm, prof[:, i].sum() / pops.sum(), 100 * prof[:, i].sum() / tot.sum()))
        print("  validity-cache misses per pop", rounds.sum() / pops.sum())
        big = np.argmax(pops)
        print("  largest plan: pops", int(pops[big]), "nodes", rec["n_expanded"][big], "ticks/pop", tot[big] / pops[big], (prof[big] / pops[big]).round(0))
    else:
        for sel, label in ((np.ones(len(pops), bool), "all plans"), (pops == pops.max(), "largest plan")):
            P = prof[sel].sum(axis=0); n = pops[sel].sum()
            print(mode, seed, label, "pops", int(n), "nodes", int(rec["n_expanded"][sel].sum()))
            for i, nm in enumerate(names):
                if nm == "-": continue
                if "count" in nm: print("    %-30s %8.3f per pop" % (nm, P[i] / n))
                else: print("    %-30s %8.1f ticks/pop" % (nm, P[i] / n))
            print("    Q total %.0f ticks/pop, E busy %.0f ticks/pop, E busy per handed node %.0f" % (P[0:6].sum() / n, (P[9] + P[10]) / n, (P[9] + P[10]) / max(P[14], 1)))
    h.close()
