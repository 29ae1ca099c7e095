"""Diagnostic (profile build): timeline of the hand-overs between the queue wave and the expander wave."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from pdmpc import backend
backend.LIB_PATH = os.environ.get("PDMPC_PROF_LIB", os.path.join(ROOT, "p-dmpc_amd", "csrc", "libpdmpc_hip_prof.so"))
from pdmpc.backend import Handle
import problems
options, mpa, iters = problems.problem_set("interx", 1, 24, Hp=8)
options.max_vehicles = 32
options.max_nodes = 1 << 17
options.trace_pops = 12 * 4000
h = Handle(options)
h.upload_mpa(mpa)
rec = h.plan_batch(iters)
big = int(np.argmax(rec["n_popped"]))
C = backend.C; abi = backend.abi
ids = np.zeros(options.trace_pops, dtype=np.int32); n = C.c_int32()
h.L.pdmpc_debug_pop_trace(h.h, big, options.trace_pops, ids.ctypes.data_as(abi.c_int32_p), C.byref(n))
t = ids.reshape(-1, 12).astype(np.int64)
t = t[200:3900]  # steady state
d = lambda a, b: ((t[:, b] - t[:, a]) & 0x7FFFFFFF)
print("vehicle", big, "pops", int(rec["n_popped"][big]))
print("Q post -> E detects          %7.0f" % np.median(d(0, 3)))
print("E detects -> E replies       %7.0f" % np.median(d(3, 4)))
ok = t[:, 7] != 0
print("  E: detect -> verdict known   %7.0f" % np.median(d(3, 5)))
print("  E: verdict -> record loaded  %7.0f" % np.median(d(5, 6)[ok]))
print("  E: record -> children done   %7.0f" % np.median(d(6, 7)[ok]))
print("  E: children -> reply         %7.0f" % np.median(d(7, 4)[ok]))
print("  handed nodes that were expanded: %.2f" % ok.mean())
print("Q post -> Q has popped tent  %7.0f" % np.median(d(0, 1)))
print("Q post -> Q sees the reply   %7.0f" % np.median(d(0, 2)))
print("E replies -> Q sees it       %7.0f" % np.median((t[:, 2] - t[:, 4]) & 0x7FFFFFFF))
nx = lambda a, b: ((t[1:, b] - t[:-1, a]) & 0x7FFFFFFF)
print("  Q: reply seen -> children visible      %7.0f" % np.median(d(2, 8)))
print("  Q: children visible -> loop head done  %7.0f" % np.median(nx(8, 9)))
print("  Q: loop head -> verdict looked up      %7.0f" % np.median((t[1:, 10] - t[1:, 9]) & 0x7FFFFFFF))
print("  Q: verdict -> posted                   %7.0f" % np.median((t[1:, 0] - t[1:, 10]) & 0x7FFFFFFF))
print("post -> next post (same wave)%7.0f" % np.median((t[1:, 0] - t[:-1, 0]) & 0x7FFFFFFF))
h.close()
