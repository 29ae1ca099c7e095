"""Diagnostic: per-phase cycle shares of the search kernel (build with -DPDMPC_PROFILE -> libpdmpc_hip_prof.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from pdmpc import backend
backend.LIB_PATH = os.environ.get("PDMPC_PROF_LIB", os.path.join(ROOT, "p-dmpc_amd", "csrc", "libpdmpc_hip_prof.so"))
from pdmpc.backend import Handle
import problems

names = ["heap_pop", "-", "validity(cache or check)", "node-load", "sincos", "children", "pushes", "loop", "-", "-", "-", "-", "-", "cache-misses(count)", "-", "-"]
for mode, seed, hp in (("interx", 1, 8), ("interx", 2, 8)):
    options, mpa, iters = problems.problem_set(mode, seed, 24, Hp=hp)
    options.max_vehicles = 32
    options.max_nodes = 1 << 17
    h = Handle(options)
    h.upload_mpa(mpa)
    rec = h.plan_batch(iters)
    prof = rec["shapes"][:, 15, :, :].reshape(len(rec), 16)
    pops = rec["n_popped"].astype(float)
    rounds = prof[:, 13].copy(); prof[:, 13] = 0
    tot = prof.sum(axis=1)
    print(mode, seed, "pops", int(pops.sum()), "cycles/pop (s_memtime ticks)", tot.sum() / pops.sum())
    for i, nm in enumerate(names):
        print("  %-14s %8.1f ticks/pop  %5.1f %%" % (nm, prof[:, i].sum() / pops.sum(), 100 * prof[:, i].sum() / tot.sum()))
    print("  validity-cache misses per pop", rounds.sum() / pops.sum())
    small = pops < 400
    print("  small plans (<400 pops): ticks/pop", tot[small].sum() / pops[small].sum(), (prof[small].sum(axis=0) / pops[small].sum()).round(0)[:12], "misses/pop", rounds[small].sum() / pops[small].sum())
    big = np.argmax(pops)
    print("  largest plan: pops", int(pops[big]), "nodes", rec["n_expanded"][big], "ticks/pop", tot[big] / pops[big], (prof[big] / pops[big]).round(0))
    h.close()
