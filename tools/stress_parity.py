"""Stress: closed-loop single-launch steps on the GPU against the oracle, several seeds and MPAs (bit-exact records every step).

    python tools/stress_parity.py                 # 3 seeds x 2 MPAs x 40 steps (about a minute on the GPU box)
    python tools/stress_parity.py --seeds 5 6 7 8 9 10 11 12 --steps 60 --realistic
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from pdmpc.config import Config, MpaType, ScenarioType
from pdmpc.road_network import boundary_provider, commonroad_scenario
from test_gpu_step import run_closed_loop

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, nargs="+", default=[2, 3, 4])
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--realistic", action="store_true", help="also the 71-trim realistic MPA (more than one successor-mask word)")
ap.add_argument("--priorities", default="constant")
a = ap.parse_args()
mpas = [(8, MpaType.single_speed), (6, MpaType.triple_speed)] + ([(5, MpaType.realistic)] if a.realistic else [])
for seed in a.seeds:
    for hp, mt in mpas:
        options = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=hp, mpa_type=mt, max_vehicles=32, max_nodes=1 << 16)
        sc = commonroad_scenario(options, seed=seed)
        ctl = run_closed_loop(options, sc, "distance", boundary_provider(sc), a.steps, priority_strategy=a.priorities)
        print("seed", seed, "Hp", hp, mt.name, "%d steps bit-identical to the oracle" % a.steps, flush=True)
