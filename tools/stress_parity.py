"""Stress: closed-loop single-launch steps on the GPU against the oracle, several seeds and MPAs (40 steps each, bit-exact)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from pdmpc.config import Config, MpaType, ScenarioType
from pdmpc.road_network import boundary_provider, commonroad_scenario
from test_gpu_step import run_closed_loop
for seed in (2, 3, 4):
    for hp, mt in ((8, MpaType.single_speed), (6, MpaType.triple_speed)):
        options = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=hp, mpa_type=mt, max_vehicles=32, max_nodes=1 << 16)
        sc = commonroad_scenario(options, seed=seed)
        ctl = run_closed_loop(options, sc, "distance", boundary_provider(sc), 40)
        print("seed", seed, "Hp", hp, mt.name, "40 steps bit-identical to the oracle", flush=True)
