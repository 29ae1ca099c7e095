"""The multi-GPU split behind the C ABI (csrc/group.cpp, include/pdmpc.h: pdmpc_group_*).

CPU: the partition logic in C++ against its Python twin (pdmpc.distributed, which the world-size-2 gloo tests drive).
GPU: a group of one device plans a time step through the same code path a group of eight takes — sub-problems, banks, RCCL
all-gather on the handle's stream, import, read-back — and must give the single launch's records."""
import numpy as np
import pytest

from pdmpc import backend, distributed


def random_dag(rng, n, p_edge, n_islands):
    """predecessor lists of a random coupling DAG of n vehicles in n_islands weakly connected groups, slots in level order"""
    island = rng.integers(0, n_islands, n)
    preds = [[] for _ in range(n)]
    for j in range(n):
        for i in range(j):
            if island[i] == island[j] and rng.random() < p_edge:
                preds[j].append(i)
    lvl = []
    for ps in preds:
        lvl.append(1 + max((lvl[p] for p in ps), default=0))
    order = sorted(range(n), key=lambda v: (lvl[v], v))
    pos = {v: i for i, v in enumerate(order)}
    return [sorted(pos[p] for p in preds[v]) for v in order]


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_partition_matches_the_python_twin(seed, world):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(5, 120))
    preds = random_dag(rng, n, 0.15, int(rng.integers(1, 9)))
    weights = [float(w) for w in rng.integers(1, 2000, n)] if seed % 2 else None
    # whole components by longest processing time
    rank_of, level_of, block = backend.group_partition(preds, world, backend.SHARD_COMPONENTS, weights)
    parts = distributed.partition_components(preds, world, weights)
    for r, slots in enumerate(parts):
        assert sorted(np.nonzero(rank_of == r)[0].tolist()) == slots
    assert (level_of == 0).all() and (block == -1).all()
    # hybrid: a dominating component by levels over all ranks
    rank_of, level_of, block = backend.group_partition(preds, world, backend.SHARD_AUTO, weights)
    parts, shared = distributed.hybrid_partition(preds, world, weights)
    for r, slots in enumerate(parts):
        assert sorted(np.nonzero(rank_of == r)[0].tolist()) == slots
    assert sorted(np.nonzero(rank_of == -1)[0].tolist()) == shared
    if shared:
        sub = distributed.sub_problem({"order": list(range(n)), "iters": [None] * n, "preds": preds, "fallback": [None] * n}, shared)
        sizes = distributed.level_sizes_of(sub["preds"])
        first = 0
        for lv, size in enumerate(sizes):
            per, blocks = distributed.level_partition(first, size, world)
            for r, (lo, hi) in enumerate(blocks):
                for s in range(lo, hi):
                    assert level_of[shared[s]] == lv + 1 and block[shared[s]] == r
            first += size
    # every level sharded
    rank_of, level_of, block = backend.group_partition(preds, world, backend.SHARD_LEVELS, weights)
    assert (rank_of == -1).all()
    sizes = distributed.level_sizes_of(preds)
    first = 0
    for lv, size in enumerate(sizes):
        per, blocks = distributed.level_partition(first, size, world)
        for r, (lo, hi) in enumerate(blocks):
            for s in range(lo, hi):
                assert level_of[s] == lv + 1 and block[s] == r
        first += size


def test_partition_rejects_a_cycle():
    with pytest.raises(backend.BackendError):
        backend.group_partition([[1], [0]], 2)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [backend.SHARD_COMPONENTS, backend.SHARD_LEVELS, backend.SHARD_AUTO])
def test_group_of_one_device_equals_the_single_launch(mode):
    """C3-like closed loop (tiled road network, colouring, two computation levels): every step through pdmpc_group_plan_step with
    one device (sub-problems, RCCL all-gather with one rank, import of the gathered records) and through pdmpc_plan_step."""
    from pdmpc.config import Config, ScenarioType
    from pdmpc.controller import PrioritizedSequentialController
    from pdmpc.iteration_data import info_from_record
    from pdmpc.mpa import get_mpa
    from pdmpc.optimizer import GraphSearchHip
    from pdmpc.road_network import boundary_provider, commonroad_scenario
    from test_gpu_parity import assert_records_equal

    options = Config(scenario_type=ScenarioType.commonroad, amount=40, Hp=8, max_num_CLs=3, max_vehicles=64, max_nodes=1 << 16)
    mpa = get_mpa(options)
    sc = commonroad_scenario(options, seed=1, tiles=3)
    opt = GraphSearchHip(options)
    opt._ensure_mpa(mpa)
    grp = backend.Group(options, n_devices=1)
    grp.upload_mpa(mpa)
    ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc), priority_strategy="coloring")
    seen = []

    def plan_step(prob):
        fb = [f if f is not None else [] for f in prob["fallback"]]
        ref = opt.handle.plan_step(prob["iters"], prob["preds"], fb)
        weights = [int(p) + 1 for p in ref["n_popped"]]
        got = grp.plan_step(prob["iters"], prob["preds"], fb, weights=weights, mode=mode)
        assert_records_equal(got, ref, "group of one, mode %d, step %d" % (mode, len(seen)))
        seen.append(len(prob["iters"]))
        return [info_from_record(ref[i], options.Hp) for i in range(len(prob["iters"]))]

    kept = []

    def plan_and_keep(prob):
        kept.append(prob)
        return plan_step(prob)

    for _ in range(6):
        ctl.step(plan_step=plan_and_keep)
    assert grp.stats()["kernel"] == 2 and grp.timing()["total"] > 0
    # ... and as resident banks launched again and again (bench.py's replay): pack once, launch without a host-to-device copy
    refs = []
    for b, prob in enumerate(kept[-3:]):
        fb = [f if f is not None else [] for f in prob["fallback"]]
        refs.append(opt.handle.plan_step(prob["iters"], prob["preds"], fb))
        grp.pack_step(b, prob["iters"], prob["preds"], fb, mode=mode)
    for rep in range(2):
        for b, prob in enumerate(kept[-3:]):
            grp.launch(b)
            assert_records_equal(grp.fetch(b, len(prob["iters"])), refs[b], "resident group bank %d, pass %d" % (b, rep))
    grp.close()
    opt.handle.close()


@pytest.mark.gpu
def test_group_arenas_grow_by_themselves_or_on_request():
    """The reference's tree is unbounded (Tree.m:54-70).  pdmpc_group_plan_step plans a step again with arenas twice as large on every
    device when a search outgrows its arena; the resident path (pack_step / launch / fetch) does not plan again: its statuses say so, and
    pdmpc_group_grow_arena sizes the arenas (what bench.py --gpus N does with the size the single handle needed)."""
    import problems
    from pdmpc import abi
    from pdmpc.backend import Handle

    options, mpa, iters = problems.problem_set("interx", 11, 12, Hp=7)
    options.max_nodes = 1 << 16
    single = Handle(options)
    single.upload_mpa(mpa)
    preds = [[] for _ in iters]
    ref = single.plan_step(iters, preds, None)
    need = int(max(ref["n_expanded"]))
    assert need > 64
    options.max_nodes = 64  # far too small for most of these searches
    grp = backend.Group(options, n_devices=1)
    grp.upload_mpa(mpa)
    from test_gpu_parity import assert_records_equal

    grp.pack_step(0, iters, preds, None, mode=backend.SHARD_COMPONENTS)
    grp.launch(0)
    small = grp.fetch(0, len(iters))
    assert (small["status"] == abi.ARENA_OVERFLOW).any()  # the resident path reports, it does not grow
    grp.grow_arena(1 << 16)
    grp.launch(0)
    assert_records_equal(grp.fetch(0, len(iters)), ref, "resident group bank after pdmpc_group_grow_arena")
    grp.close()
    grp = backend.Group(options, n_devices=1)  # 64 nodes again
    grp.upload_mpa(mpa)
    assert_records_equal(grp.plan_step(iters, preds, None), ref, "pdmpc_group_plan_step grows the arenas by itself")
    grp.close()
    single.close()


# ---------------------------------------------------------------------------------------------------------------------------------
# More than one rank.  Two or four LOGICAL ranks on the one GPU of the box: a handle, a stream and arenas each, the exchange as peer
# copies ordered by events (PDMPC_COLLECTIVE_COPY behind the function table of csrc/group.cpp) — slot remapping per rank, block
# partition of a level, export / all-gather / import of the other ranks' blocks, the scatter back into the caller's order: everything
# a group of distinct devices executes except the transport.  (PrioritizedController.m:356-365 send, :476-491 read.)

ALL_MODES = [backend.SHARD_COMPONENTS, backend.SHARD_LEVELS, backend.SHARD_AUTO]


def _current_device():
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")
    d = ctypes.c_int(-1)
    assert hip.hipGetDevice(ctypes.byref(d)) == 0
    return d.value


def _c3_like(amount=40, tiles=3, max_cls=3):
    from pdmpc.config import Config, ScenarioType
    from pdmpc.controller import PrioritizedSequentialController
    from pdmpc.mpa import get_mpa
    from pdmpc.optimizer import GraphSearchHip
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=amount, Hp=8, max_num_CLs=max_cls, max_vehicles=64, max_nodes=1 << 16)
    mpa = get_mpa(options)
    sc = commonroad_scenario(options, seed=1, tiles=tiles)
    opt = GraphSearchHip(options)
    opt._ensure_mpa(mpa)
    ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc), priority_strategy="coloring")
    return options, mpa, opt, ctl


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("mode", ALL_MODES)
def test_group_of_logical_ranks_c3_step(world, mode):
    """A C3-like closed loop (tiled road network = several coupling-graph components, colouring, computation levels): every step through
    a group of 2 / 4 ranks in all three modes, then as resident banks (pack_step / launch / fetch) launched repeatedly — records equal
    to the single launch's, and the caller's current device untouched."""
    from pdmpc.iteration_data import info_from_record
    from test_gpu_parity import assert_records_equal

    options, mpa, opt, ctl = _c3_like()
    dev0 = _current_device()
    grp = backend.Group(options, n_devices=world, devices=[0] * world)
    assert grp.collective == "copy" and _current_device() == dev0
    grp.upload_mpa(mpa)
    kept, refs = [], []

    def plan_step(prob):
        fb = [f if f is not None else [] for f in prob["fallback"]]
        ref = opt.handle.plan_step(prob["iters"], prob["preds"], fb)
        weights = [int(p) + 1 for p in ref["n_popped"]]
        got = grp.plan_step(prob["iters"], prob["preds"], fb, weights=weights, mode=mode)
        assert_records_equal(got, ref, "group of %d logical ranks, mode %d, step %d" % (world, mode, len(kept)))
        kept.append(prob)
        refs.append(ref.copy())
        return [info_from_record(ref[i], options.Hp) for i in range(len(prob["iters"]))]

    for _ in range(5):
        ctl.step(plan_step=plan_step)
    assert _current_device() == dev0
    # the split really is one: more than one rank plans something
    rank_of, level_of, block = backend.group_partition(kept[-1]["preds"], world, mode)
    busy = set(rank_of[rank_of >= 0].tolist()) | set(block[block >= 0].tolist())
    assert len(busy) >= 2, (rank_of, block)
    # resident banks
    for b, prob in enumerate(kept[-3:]):
        fb = [f if f is not None else [] for f in prob["fallback"]]
        grp.pack_step(b, prob["iters"], prob["preds"], fb, weights=[int(p) + 1 for p in refs[-3 + b]["n_popped"]], mode=mode)
    for rep in range(2):
        for b, prob in enumerate(kept[-3:]):
            grp.launch(b)
            assert_records_equal(grp.fetch(b, len(prob["iters"])), refs[-3 + b], "resident bank %d of %d logical ranks, pass %d" % (b, world, rep))
    # the devices hold the records of the bank launched last: another bank's fetch is refused, not mis-scattered
    with pytest.raises(backend.BackendError):
        grp.fetch(0, len(kept[-3]["iters"]))
    st = grp.stats_all()
    assert st["bad_status_plans"] == 0 and len(st["per_rank"]) == world and _current_device() == dev0
    grp.close()
    opt.handle.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_group_of_logical_ranks_c5_batch(world):
    """C5's shape: the prioritization instances of an explorative step are components of their batch and are dealt out over the ranks
    (PrioritizedExplorativeController.m:25-176); one all-gather ends the step.  Records and the chosen prioritization equal the single
    launch's."""
    from pdmpc.config import Config, ScenarioType
    from pdmpc.controller import PrioritizedSequentialController
    from pdmpc.explorative import build_exploration_batch, choose_solution
    from pdmpc.mpa import get_mpa
    from pdmpc.optimizer import GraphSearchHip
    from pdmpc.road_network import boundary_provider, commonroad_scenario
    from test_gpu_parity import assert_records_equal

    K = 12
    options = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=8, max_vehicles=20 * K, max_nodes=1 << 15)
    mpa = get_mpa(options)
    sc = commonroad_scenario(options, seed=1)
    opt = GraphSearchHip(options)
    ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc))
    for _ in range(3):
        ctl.step(plan_step=lambda prob: opt.run_optimizer_step(prob, mpa))
    batch = build_exploration_batch(ctl, K, seed=4)
    fb = [f if f is not None else [] for f in batch["fallback"]]
    ref = opt.handle.plan_step(batch["iters"], batch["preds"], fb)
    grp = backend.Group(options, n_devices=world, devices=[0] * world)
    grp.upload_mpa(mpa)
    for mode in (backend.SHARD_COMPONENTS, backend.SHARD_AUTO):
        got = grp.plan_step(batch["iters"], batch["preds"], fb, weights=[int(p) + 1 for p in ref["n_popped"]], mode=mode)
        assert_records_equal(got, ref, "explorative batch over %d logical ranks, mode %d" % (world, mode))
        a, ca = choose_solution(batch, got, options.Hp)
        b, cb = choose_solution(batch, ref, options.Hp)
        assert a == b and np.array_equal(ca, cb)
    rank_of, _, _ = backend.group_partition(batch["preds"], world, backend.SHARD_COMPONENTS)
    assert len(set(rank_of.tolist())) == world
    grp.close()
    opt.handle.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ALL_MODES)
def test_group_of_logical_ranks_tied_search(mode):
    """Equal keys (a mirror-symmetric vehicle: the result depends on the binary heap's layout, priority_queue_interface_mex.cpp:19-31)
    behind a long search of another rank: the replay through the heap happens on whichever rank owns the vehicle, its predecessor's
    areas arrive through the exchange (levels) or on the device (components)."""
    import problems
    from pdmpc.optimizer import GraphSearchHip
    from test_gpu_parity import assert_records_equal

    options = problems.make_options("interx", Hp=8)
    options.max_vehicles = 8
    options.max_nodes = 1 << 16
    mpa = problems.get_mpa(options)
    rng = np.random.default_rng(2)
    heavy = max((problems.road_problem(rng, options, mpa) for _ in range(12)), key=lambda it: len(it.dynamic_obstacle_area) + len(it.obstacles))
    sym = problems.symmetric_problem(options, mpa, block_x=0.5)
    # two components: {0, 1, 2} chained and {3, 4} chained; levels of width two and one
    iters = [heavy, sym, sym, sym, heavy]
    preds = [[], [0], [0, 1], [], [3]]
    opt = GraphSearchHip(options)
    opt._ensure_mpa(mpa)
    ref = opt.handle.plan_step(iters, preds, [[] for _ in iters])
    assert opt.handle.stats()["queue_fallbacks"] > 0
    grp = backend.Group(options, n_devices=2, devices=[0, 0])
    grp.upload_mpa(mpa)
    got = grp.plan_step(iters, preds, [[] for _ in iters], mode=mode)
    assert_records_equal(got, ref, "tied searches over two logical ranks, mode %d" % mode)
    assert grp.stats_all()["queue_fallbacks"] > 0
    grp.close()
    opt.handle.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [backend.SHARD_COMPONENTS, backend.SHARD_AUTO, backend.SHARD_LEVELS])
def test_group_accepts_any_slot_order(mode):
    """include/pdmpc.h: "any slot order".  The caller's slots shuffled (predecessors in HIGHER slots): a rank's sub-problem is built in
    level order, so the handle does not permute it and the device-resident record path applies."""
    from test_gpu_parity import assert_records_equal

    options, mpa, opt, ctl = _c3_like(amount=30, tiles=2)
    kept = []

    def plan_step(prob):
        kept.append(prob)
        return opt.run_optimizer_step(prob, mpa)

    for _ in range(3):
        ctl.step(plan_step=plan_step)
    prob = kept[-1]
    n = len(prob["iters"])
    assert any(prob["preds"])
    rng = np.random.default_rng(5)
    perm = rng.permutation(n)  # new slot s holds old slot perm[s]
    inv = np.argsort(perm)
    iters = [prob["iters"][int(o)] for o in perm]
    preds = [sorted(int(inv[p]) for p in prob["preds"][int(o)]) for o in perm]
    fb = [prob["fallback"][int(o)] if prob["fallback"][int(o)] is not None else [] for o in perm]
    assert any(p > s for s, ps in enumerate(preds) for p in ps)  # some predecessor sits in a higher slot
    ref = opt.handle.plan_step(iters, preds, fb)
    grp = backend.Group(options, n_devices=2, devices=[0, 0])
    grp.upload_mpa(mpa)
    got = grp.plan_step(iters, preds, fb, mode=mode)
    assert_records_equal(got, ref, "shuffled slots, mode %d" % mode)
    grp.close()
    opt.handle.close()
