"""Closed loop of the circle scenario (BASELINE config 0 widened to 8 vehicles, SAT checker, sequential levels: one launch per
computation level as the reference's controller drives the optimizer) timed on the GPU: ms per time step and the statistics of the
launches.  PDMPC_PKG=<dir with the pdmpc package> times another build of the library (A/B against an earlier round)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.environ.get("PDMPC_PKG") or os.path.join(ROOT, "p-dmpc_amd")]
import numpy as np
from pdmpc.config import Config, MpaType, ScenarioType
from pdmpc.controller import PrioritizedSequentialController
from pdmpc.mpa import get_mpa
from pdmpc.optimizer import OptimizerInterface
from pdmpc.scenario import circle_scenario

n_veh = int(os.environ.get("N_VEH", "8"))
n_steps = int(os.environ.get("N_STEPS", "30"))
options = Config(scenario_type=ScenarioType.circle, amount=n_veh, Hp=int(os.environ.get("HP", "6")), mpa_type=MpaType.single_speed, max_vehicles=16, max_nodes=1 << 16)
mpa = get_mpa(options)
optimizer = OptimizerInterface.get_optimizer(options)
launch_ms, kernel_ms = [], []
def plan_level(iters):
    t0 = time.perf_counter()
    infos = optimizer.run_optimizer_batch(iters, mpa)
    launch_ms.append(1e3 * (time.perf_counter() - t0))
    kernel_ms.append(optimizer.handle.stats()["kernel_ms"])
    return infos
def plan_step(prob):
    t0 = time.perf_counter()
    infos = optimizer.run_optimizer_step(prob, mpa)
    launch_ms.append(1e3 * (time.perf_counter() - t0))
    kernel_ms.append(optimizer.handle.stats()["kernel_ms"])
    return infos
for mode in ("one launch per level", "one launch per step"):
    launch_ms.clear(); kernel_ms.clear()
    ctl = PrioritizedSequentialController(options, circle_scenario(options), mpa, plan_level)
    step_ms, step_k = [], []
    for s in range(n_steps):
        n0 = len(launch_ms)
        if mode == "one launch per level":
            ctl.step()
        else:
            ctl.step(plan_step=plan_step)
        step_ms.append(sum(launch_ms[n0:])); step_k.append(sum(kernel_ms[n0:]))
    st = optimizer.handle.stats()
    print("circle %d vehicles Hp %d, %s: host %.3f ms per step (median %.3f), kernels %.3f ms per step (median %.3f), kernel %s, queue_fallbacks %s" % (
        n_veh, options.Hp, mode, float(np.mean(step_ms[3:])), float(np.median(step_ms[3:])), float(np.mean(step_k[3:])), float(np.median(step_k[3:])), st.get("kernel"), st.get("queue_fallbacks")))
