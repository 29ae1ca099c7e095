"""Generates the committed golden vectors for the optimizer path.

Two kinds of data end up in tests/golden/:
  reference_known_answers.json  the known-answer vectors the REFERENCE's own tests hold for this path
                                (tests/unittests/hlc/intersect_unittest.m:8-54): input polygons + expected booleans,
                                plus lanelet 1 of the lab map (rows [rx ry lx ly cx cy], from the labmap fixture).
  oracle_plans_*.npz            outputs of the CPU oracle on seeded synthetic problems (tests/problems.py): the pinned
                                results every later build (oracle and HIP kernel alike) must reproduce bit for bit.
The reference is MATLAB and cannot run here, so there are no reference-generated outputs ("Parity").
Run:  python tests/golden/make_golden_vectors.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd"), os.path.join(ROOT, "tests")]

import problems  # noqa: E402
from oracle import oracle  # noqa: E402
from pdmpc.config import MpaType  # noqa: E402
from pdmpc.road_network import lab_map  # noqa: E402


def main():
    shape1 = [[-7.0749, -2.8728, 9.8889, 21.3024, 15.3469, 7.9387], [-6.4707, -12.1152, -24.4428, -3.0950, 19.3838, 18.7030]]
    ka = {
        "source": "reference tests/unittests/hlc/intersect_unittest.m",
        "intersect_sat": [
            {"shape1": shape1, "shift": [-5, -5], "expected": True, "line": "38-45 testPolygonPos"},
            {"shape1": shape1, "shift": [-40, -40], "expected": False, "line": "47-54 testPolygonNeg"},
        ],
        "intersect_lanelets": [
            {"shape": [[0, 5, 5, 0], [0, 0, 5, 5]], "expected": True, "line": "8-16"},
            {"shape": [[2.4, 2.5, 2.5, 2.4], [3.7, 3.7, 3.8, 3.8]], "expected": False, "line": "18-26"},
            {"shape": [[2.2, 2.4, 2.4, 2.2], [3.7, 3.7, 3.9, 3.9]], "expected": True, "line": "28-36"},
        ],
        "lanelet_1_rows_rx_ry_lx_ly_cx_cy": lab_map().lanelets[0].tolist(),
        "priority_queue": {
            "comment": "SURVEY.md Appendix A probe of the reference comparator with libstdc++: push (id,key) then pop all",
            "push": [[1, 1.0], [2, 0.5], [3, 0.5], [4, 0.5], [5, 2.0], [6, 0.5]],
            "pops": [2, 3, 6, 4, 1, 5],
        },
    }
    with open(os.path.join(HERE, "reference_known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1)

    for name, mode, seed, count, Hp, mt in (
        ("interx_single_hp6", "interx", 101, 12, 6, MpaType.single_speed),
        ("sat_single_hp5", "sat", 102, 12, 5, MpaType.single_speed),
        ("interx_triple_hp8", "interx", 103, 8, 8, MpaType.triple_speed),
    ):
        options, mpa, iters = problems.problem_set(mode, seed, count, Hp=Hp, mpa_type=mt)
        options.max_nodes = 1 << 15
        _, recs, traces = oracle.plan_batch(options, mpa, iters, trace=True)
        pops = np.zeros((count, 64), dtype=np.int32)
        for i, t in enumerate(traces):
            m = min(64, len(t.pops))
            pops[i, :m] = t.pops[:m]
        np.savez_compressed(
            os.path.join(HERE, "oracle_plans_%s.npz" % name),
            mode=mode, seed=seed, count=count, Hp=Hp, mpa_type=mt.name,
            records=recs.view(np.uint8).reshape(count, -1), first_pops=pops,
            tree_sizes=np.array([len(t.tree["x"]) for t in traces]),
        )
        print(name, "status", recs["status"].tolist(), "pops", recs["n_popped"].tolist())

    # the sampled optimizer (MonteCarloTreeSearch.m): records of the oracle for seeded problems and seeds
    for name, mode, seed, count, Hp in (("sampled_interx_hp6", "interx", 111, 10, 6), ("sampled_sat_hp8", "sat", 112, 10, 8)):
        options, mpa, iters = problems.problem_set(mode, seed, count, Hp=Hp)
        seeds = [3 + 2 * i for i in range(count)]
        _, recs = oracle.plan_batch_sampled(options, mpa, iters, seeds)
        np.savez_compressed(
            os.path.join(HERE, "oracle_plans_%s.npz" % name),
            mode=mode, seed=seed, count=count, Hp=Hp, rng_seeds=np.array(seeds),
            records=recs.view(np.uint8).reshape(count, -1),
        )
        print(name, "status", recs["status"].tolist(), "expansions", recs["n_expanded"].tolist())


if __name__ == "__main__":
    main()
