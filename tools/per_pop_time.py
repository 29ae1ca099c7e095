import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from pdmpc.backend import Handle
import problems
for mode, seed, hp in (("interx", 1, 8), ("interx", 2, 8)):
    options, mpa, iters = problems.problem_set(mode, seed, 24, Hp=hp)
    options.max_vehicles = 32
    options.max_nodes = 1 << 17
    h = Handle(options)
    h.upload_mpa(mpa)
    rec = h.plan_batch(iters)
    rec = h.plan_batch(iters)
    st = h.stats()
    pops = rec["n_popped"]
    print(mode, seed, "kernel_ms %.3f" % st["kernel_ms"], "pops total", int(pops.sum()), "max", int(pops.max()), "fallbacks", st["queue_fallbacks"], "us/pop of heaviest %.3f" % (1e3 * st["kernel_ms"] / pops.max()))
    # heaviest alone
    big = int(np.argmax(pops))
    rec1 = h.plan_batch([iters[big]])
    st = h.stats()
    print("   heaviest alone: kernel_ms %.3f pops %d -> %.3f us/pop" % (st["kernel_ms"], int(rec1["n_popped"][0]), 1e3 * st["kernel_ms"] / int(rec1["n_popped"][0])))
    h.close()
