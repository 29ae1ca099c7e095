"""world_size-2 gloo test of the level-sharded step planner (no GPU): partition, per-level all-gather, import.

The per-range planner is a CPU stand-in built on the oracle; what is under test is pdmpc.distributed: every rank must
end with the records of ALL slots, identical to the single-process result.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pdmpc import abi
from pdmpc.distributed import REC_BYTES, level_partition, plan_step_sharded


class OracleRangePlanner:
    """Same interface as HipRangePlanner, planning with the CPU oracle into a host-side record array."""

    def __init__(self, options, mpa):
        self.options, self.mpa = options, mpa

    def begin(self, problem):
        self.problem = problem
        self.recs = abi.out_array(len(problem["iters"]))
        self.known = np.zeros(len(problem["iters"]), dtype=bool)

    def new_buffer(self, n):
        return torch.zeros(max(n, 1) * REC_BYTES, dtype=torch.uint8)

    def buffers(self, per, world):
        return self.new_buffer(per), self.new_buffer(per * world)

    def plan_range(self, first, count, send):
        import copy

        from oracle import oracle

        Hp = self.options.Hp
        iters = []
        for s in range(first, first + count):
            it = copy.copy(self.problem["iters"][s])
            dyn = list(it.dynamic_obstacle_area)
            for p in self.problem["preds"][s]:
                assert self.known[p], "slot %d planned before its predecessor %d arrived" % (s, p)
                if int(self.recs[p]["status"]) == 0:
                    dyn.append([np.array(self.recs[p]["shapes"][k][:, : int(self.recs[p]["shape_cols"][k])]) for k in range(Hp)])
            it.dynamic_obstacle_area = dyn
            iters.append(it)
        if count:
            _, out, _ = oracle.plan_batch(self.options, self.mpa, iters)
            self.recs[first : first + count] = out
            self.known[first : first + count] = True
            raw = np.frombuffer(out.tobytes(), dtype=np.uint8)
            send[: raw.size] = torch.from_numpy(raw.copy())

    def import_records(self, first, count, buf):
        raw = buf.numpy().tobytes()
        self.recs[first : first + count] = np.frombuffer(raw, dtype=abi.VEHICLE_OUT_DTYPE)
        self.known[first : first + count] = True

    def fetch(self, n):
        assert self.known[:n].all()
        return self.recs[:n].copy()


def make_problem():
    from pdmpc.config import Config, ScenarioType
    from pdmpc.controller import PrioritizedSequentialController
    from pdmpc.mpa import get_mpa
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    from oracle import oracle
    from pdmpc.iteration_data import info_from_record

    options = Config(scenario_type=ScenarioType.commonroad, amount=12, Hp=5, max_nodes=1 << 15)
    mpa = get_mpa(options)
    sc = commonroad_scenario(options, seed=4)
    ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc))
    probs = []

    def plan_step(prob):
        probs.append(prob)
        recs, _ = oracle.plan_step(options, mpa, prob)
        return [info_from_record(recs[i], options.Hp) for i in range(len(recs))]

    for _ in range(3):
        ctl.step(plan_step=plan_step)
    return options, mpa, probs[-1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    options, mpa, prob = make_problem()
    recs = plan_step_sharded(prob, OracleRangePlanner(options, mpa), dist, rank, world)
    q.put((rank, recs.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_level_partition_covers_every_slot_once():
    for size in range(0, 23):
        for world in (1, 2, 3, 4, 8):
            per, parts = level_partition(5, size, world)
            covered = [s for lo, hi in parts for s in range(lo, hi)]
            assert covered == list(range(5, 5 + size))
            assert all(hi - lo <= per for lo, hi in parts)


@pytest.mark.timeout(300)
def test_sharded_step_world2_matches_single_process():
    from oracle import oracle

    options, mpa, prob = make_problem()
    assert max(prob["level_sizes"]) >= 2 and len(prob["level_sizes"]) >= 2
    want, _ = oracle.plan_step(options, mpa, prob)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        recs = np.frombuffer(got[r], dtype=abi.VEHICLE_OUT_DTYPE)
        for name in ("status", "n_expanded", "n_popped", "tree_path", "predicted_trims"):
            assert np.array_equal(recs[name], want[name]), (r, name)
        assert np.array_equal(recs["y_predicted"], want["y_predicted"], equal_nan=True)
        assert np.array_equal(recs["shapes"], want["shapes"])


def test_single_rank_is_the_plain_level_loop():
    from oracle import oracle

    options, mpa, prob = make_problem()
    want, _ = oracle.plan_step(options, mpa, prob)
    got = plan_step_sharded(prob, OracleRangePlanner(options, mpa), None, 0, 1)
    assert np.array_equal(got["n_popped"], want["n_popped"])
    assert np.array_equal(got["y_predicted"], want["y_predicted"], equal_nan=True)


# ---- component sharding ------------------------------------------------------------------------------------------
def test_partition_components_keeps_components_whole_and_balances():
    from pdmpc.distributed import partition_components, weak_components

    preds = [[], [0], [], [2], [3], [], [], [6, 5], [], []]  # components {0,1}, {2,3,4}, {5,6,7}, {8}, {9}
    labels = weak_components(preds)
    assert labels == [0, 0, 2, 2, 2, 5, 5, 5, 8, 9]
    for world in (1, 2, 3, 4):
        parts = partition_components(preds, world)
        assert sorted(s for p in parts for s in p) == list(range(10))
        for p in parts:
            for s in p:
                assert all(q in p for q in preds[s])
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 3


def _component_worker(rank, world, port, q):
    from oracle import oracle
    from pdmpc.distributed import gather_records, partition_components, sub_problem

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pdmpc.distributed import assemble_records, shard_problems

    options, mpa, prob = make_tiled_problem()
    prob["pops"] = [7 * (s % 5) + 1 for s in range(len(prob["iters"]))]  # (the weights bench.py's replay partitions by)
    # bench.py's replay path: shard_problems -> plan the rank's sub-problem -> all-gather of the blocks -> assemble_records
    (parts,), (sub,) = shard_problems([prob], world, rank)
    assert parts == partition_components(prob["preds"], world, weights=[w + 1 for w in prob["pops"]]) and sub["slots"] == parts[rank]
    recs, _ = oracle.plan_step(options, mpa, dict(sub, level_sizes=levels_of(sub["preds"])))
    local = torch.from_numpy(np.frombuffer(recs.tobytes(), dtype=np.uint8).copy())
    blocks = gather_records(local, len(parts[rank]), parts, dist, rank, world, lambda n: torch.zeros(max(n, 1) * REC_BYTES, dtype=torch.uint8))
    full = assemble_records([blk.numpy().tobytes() for blk in blocks], parts, len(prob["iters"]))
    q.put((rank, full.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def levels_of(preds):
    """kahn level sizes of a problem whose slots are already in level order."""
    lvl = []
    for s, ps in enumerate(preds):
        lvl.append(1 + max((lvl[p] for p in ps), default=0))
    assert lvl == sorted(lvl)
    return [lvl.count(v) for v in range(1, max(lvl) + 1)]


def make_tiled_problem():
    from pdmpc.config import Config, ScenarioType
    from pdmpc.controller import PrioritizedSequentialController
    from pdmpc.mpa import get_mpa
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=18, Hp=5, max_nodes=1 << 15)
    mpa = get_mpa(options)
    sc = commonroad_scenario(options, seed=6, tiles=3)
    ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc))
    return options, mpa, ctl.build_step_problem()


@pytest.mark.timeout(300)
def test_component_sharding_world2_matches_single_process():
    from oracle import oracle

    options, mpa, prob = make_tiled_problem()
    want, _ = oracle.plan_step(options, mpa, prob)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_component_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        recs = np.frombuffer(got[r], dtype=abi.VEHICLE_OUT_DTYPE)
        assert np.array_equal(recs.view(np.uint8), want.view(np.uint8))


# ---- hybrid sharding: a dominating component is level-sharded over all ranks, the others stay whole ------------------
def make_dominated_problem():
    """One 12-vehicle network (a single component) next to two small tiles... built by merging a coupled problem with a tiled one."""
    options, mpa, big = make_problem()  # 12 vehicles, coupled: one or two components
    return options, mpa, big


def _hybrid_worker(rank, world, port, q):
    from pdmpc.distributed import plan_step_hybrid

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    options, mpa, prob = make_dominated_problem()
    out["dominated"] = plan_step_hybrid(prob, OracleRangePlanner(options, mpa), dist, rank, world, dominance=0.5).tobytes()
    options2, mpa2, prob2 = make_tiled_problem()
    out["tiled"] = plan_step_hybrid(prob2, OracleRangePlanner(options2, mpa2), dist, rank, world).tobytes()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_hybrid_partition_shares_only_a_dominating_component():
    from pdmpc.distributed import hybrid_partition

    preds = [[]] + [[i - 1] for i in range(1, 12)] + [[], [12], [], []]  # a chain of 12, a pair, two singles
    parts, shared = hybrid_partition(preds, 2)
    assert shared == list(range(12))
    assert sorted(s for p in parts for s in p) == [12, 13, 14, 15]
    parts, shared = hybrid_partition(preds, 2, dominance=2.0)
    assert shared == [] and sorted(s for p in parts for s in p) == list(range(16))
    parts, shared = hybrid_partition([[], [0], [], [2], [], [4]], 2)  # three equal components: nothing dominates
    assert shared == []


@pytest.mark.timeout(300)
def test_hybrid_sharding_world2_matches_single_process():
    from oracle import oracle
    from pdmpc.distributed import hybrid_partition

    options, mpa, prob = make_dominated_problem()
    assert hybrid_partition(prob["preds"], 2, dominance=0.5)[1], "the test problem has no dominating component"
    want = oracle.plan_step(options, mpa, prob)[0]
    options2, mpa2, prob2 = make_tiled_problem()
    want2 = oracle.plan_step(options2, mpa2, prob2)[0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hybrid_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        assert np.array_equal(np.frombuffer(got[r]["dominated"], dtype=np.uint8), want.view(np.uint8))
        assert np.array_equal(np.frombuffer(got[r]["tiled"], dtype=np.uint8), want2.view(np.uint8))


def test_partition_instances_deals_instances_round_robin():
    from pdmpc.distributed import partition_instances, sub_problem

    batch = {"instance": [0, 1, 2, 3, 0, 1, 2, 3, 0, 2], "preds": [[], [], [], [], [0], [1], [2], [3], [4], [6]], "iters": list(range(10)),
             "order": list(range(10)), "fallback": [None] * 10}
    parts = partition_instances(batch, 2)
    assert parts == [[0, 2, 4, 6, 8, 9], [1, 3, 5, 7]]
    sub = sub_problem(batch, parts[0])
    assert sub["preds"] == [[], [], [0], [1], [2], [3]]


# ---- C5: the instance-sharded replay of bench.py (shard_problems(explore=True) -> plan -> all-gather -> assemble_records) --------
def make_exploration_problem():
    from pdmpc.config import Config, ScenarioType
    from pdmpc.controller import PrioritizedSequentialController
    from pdmpc.explorative import build_exploration_batch
    from pdmpc.mpa import get_mpa
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=8, Hp=5, max_nodes=1 << 15)
    mpa = get_mpa(options)
    sc = commonroad_scenario(options, seed=3)
    ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc))
    return options, mpa, build_exploration_batch(ctl, 4, 1)


def _instance_worker(rank, world, port, q):
    from oracle import oracle
    from pdmpc.distributed import assemble_records, gather_records, shard_problems

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    options, mpa, batch = make_exploration_problem()
    (parts,), (sub,) = shard_problems([batch], world, rank, explore=True)
    assert all(batch["instance"][s] % world == rank for s in parts[rank])
    recs, _ = oracle.plan_step(options, mpa, dict(sub, level_sizes=levels_of(sub["preds"])))
    local = torch.from_numpy(np.frombuffer(recs.tobytes(), dtype=np.uint8).copy())
    blocks = gather_records(local, len(parts[rank]), parts, dist, rank, world, lambda n: torch.zeros(max(n, 1) * REC_BYTES, dtype=torch.uint8))
    full = assemble_records([blk.numpy().tobytes() for blk in blocks], parts, len(batch["iters"]))
    q.put((rank, full.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_instance_sharding_world2_matches_single_process():
    """Config C5 on two ranks the way bench.py replays it: instances dealt out, each rank plans its own in one batch, one all-gather,
    records of all instances on every rank equal to the single-process batch."""
    from oracle import oracle

    options, mpa, batch = make_exploration_problem()
    want, _ = oracle.plan_step(options, mpa, dict(batch, level_sizes=levels_of(batch["preds"])))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_instance_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        assert np.array_equal(np.frombuffer(got[r], dtype=np.uint8), want.view(np.uint8))
