"""Where the host-inclusive closed loop's time goes (C2): per step the native controller's build_step (host logic), pdmpc_plan_step
(pack + H2D + launch + D2H) and apply, each timed on its own over the driver's window (closed-loop steps 21-40), next to
pdmpc_controller_run's own per-step times."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd")]
import numpy as np
import bench
from pdmpc import abi
class A: pass
args = A(); args.vehicles = 20; args.hp = 8; args.mpa = "single_speed"; args.instances = 1; args.workload = "c2"; args.max_nodes = 1 << 17; args.seed = 1; args.max_levels = 99; args.priorities = "constant"
options, mpa, ctl = bench.build_world(args, 0)
from pdmpc.optimizer import GraphSearchHip
from pdmpc.native_controller import NativeController
from pdmpc.road_network import commonroad_scenario
opt = GraphSearchHip(options); opt._ensure_mpa(mpa); h = opt.handle
nat = NativeController(options, commonroad_scenario(options, seed=1, tiles=1), mpa, h, coupling="distance", priority_strategy="constant")
L = nat.L
nat.run(20)
tb, tp, ta, tk = [], [], [], []
for s in range(20):
    t0 = time.perf_counter(); nat.build_step(); t1 = time.perf_counter()
    n = C.c_int32(); vin = C.POINTER(abi.VehicleIn)(); po, pi, order, levels = abi.c_int32_p(), abi.c_int32_p(), abi.c_int32_p(), abi.c_int32_p(); fb = C.POINTER(abi.PolygonSet)()
    L.pdmpc_controller_problem(nat.c, C.byref(n), C.byref(vin), C.byref(po), C.byref(pi), C.byref(fb), C.byref(order), C.byref(levels))
    out = abi.out_array(n.value)
    t1 = time.perf_counter()
    rc = L.pdmpc_plan_step(h.h, n.value, vin, po, pi, fb, abi.out_ptr(out)); t2 = time.perf_counter()
    assert rc == 0
    tk.append(h.stats()["kernel_ms"])
    t2b = time.perf_counter(); nat.apply(out[: n.value]); t3 = time.perf_counter()
    tb.append(1e3 * (t1 - t0)); tp.append(1e3 * (t2 - t1)); ta.append(1e3 * (t3 - t2b))
print("build_step %.3f ms, plan_step %.3f ms (of which kernel %.3f), apply %.3f ms per step; sum %.3f" % (np.mean(tb), np.mean(tp), np.mean(tk), np.mean(ta), np.mean(tb) + np.mean(tp) + np.mean(ta)))
ms = nat.run(20)
print("pdmpc_controller_run, the 20 steps after those: %.3f ms per step" % float(np.mean(ms)))
