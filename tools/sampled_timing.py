"""Diagnostic: the sampled optimizer — kernel time of one computation level on the GPU against the oracle on the host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from pdmpc.backend import Handle
from oracle import oracle
import problems
for n in (1, 20, 256, 1024):
    options, mpa, iters = problems.problem_set("interx", 1, n, Hp=8)
    options.max_vehicles = max(n, 32)
    options.max_nodes = 4096
    h = Handle(options)
    h.upload_mpa(mpa)
    seeds = list(range(5, 5 + n))
    h.plan_batch_sampled(iters, seeds)
    h.reset_stats()
    t0 = time.perf_counter()
    rec = h.plan_batch_sampled(iters, seeds)
    wall = time.perf_counter() - t0
    st = h.stats()
    t0 = time.perf_counter()
    _, ref = oracle.plan_batch_sampled(options, mpa, iters, seeds, n_threads=os.cpu_count())
    cpu = time.perf_counter() - t0
    same = all(np.array_equal(rec[k].view(np.uint8), ref[k].view(np.uint8)) for k in range(0))
    print("%5d vehicles: kernel %.3f ms (call %.1f ms incl. host-side random numbers and packing), oracle on %d threads %.1f ms, expansions/vehicle %.0f"
          % (n, st["kernel_ms"], 1e3 * wall, os.cpu_count(), 1e3 * cpu, rec["n_expanded"].mean()))
    h.close()
