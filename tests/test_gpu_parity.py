"""GPU parity tests proper: HIP backend (through the C ABI) vs the CPU oracle on identical inputs.

The bar is bit-exactness: result records byte-identical, node-pop sequence identical, whole search tree
identical (x, y, yaw, g, h as raw IEEE bits; trim, k, parent as integers).
"""
import copy
import os

import numpy as np
import pytest

from pdmpc import abi
from pdmpc.backend import Handle
from pdmpc.config import MpaType

import problems

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import oracle

    return oracle


def assert_records_equal(gpu, ref, ctx=""):
    assert gpu.dtype == ref.dtype
    for name in gpu.dtype.names:
        a, b = gpu[name], ref[name]
        if a.dtype.kind == "f":
            same = a.view(np.uint64) == b.view(np.uint64)
        else:
            same = a == b
        assert np.all(same), "%s field %s differs at %s" % (ctx, name, np.argwhere(~same)[:5])


def compare_tree_and_pops(h, v, trace, pops=True):
    """Pop sequence and whole search tree of vehicle v against the oracle's.  Where two popped nodes carry the same key the
    reference's order is that of its binary heap; the kernel only reproduces it where it decides the result (the search then ends
    on the replay through the libstdc++-faithful heap, which leaves the exact pop sequence behind), so for searches with tied pops
    the pop sequence and the tree are compared as sets — unless every search is replayed (PDMPC_TUNING=force_tie=1): then the order
    is the heap's and is compared as such."""
    cap = 1 << 16
    if len(trace.tree["x"]) >= cap or len(trace.pops) >= cap:
        # the oracle's trace is cut at its capacity: counts only (the records, incl. the ids along the path, are compared by the caller)
        assert len(h.tree(v, capacity=cap)["x"]) == cap or len(h.pop_trace(v, capacity=cap)) == cap
        return
    tied = problems.tied_pops(trace) > 0 and "force_tie=1" not in os.environ.get("PDMPC_TUNING", "")
    if pops:
        got = h.pop_trace(v, capacity=cap)
        want = trace.pops[:cap]
        assert len(got) == len(want), "pop count of vehicle %d" % v
        if tied:
            assert len(trace.pops) > cap or np.array_equal(np.sort(got), np.sort(want)), "popped set of vehicle %d" % v
        else:
            assert np.array_equal(got, want), "pop sequence of vehicle %d" % v
    tree = h.tree(v, capacity=cap)
    n = min(len(trace.tree["x"]), cap)
    assert len(tree["x"]) == n
    if tied:
        # Tied keys: the creation order of the reference's tree follows its binary heap, which the kernel only reproduces
        # where it decides the result.  The trees are then compared as sets of NODES (not of independent per-field multisets): rows
        # (x, y, yaw, g, h, trim, k) sorted lexicographically, and the parent links through the sort -- a node's parent must be the
        # same node (same row) on both sides.
        if len(trace.tree["x"]) <= cap:
            def rows(t, m):
                own = np.stack([t[k][:m].view(np.uint64) for k in ("x", "y", "yaw", "g", "h")] + [t[k][:m].astype(np.uint64) for k in ("trim", "k")], axis=1)
                par = np.where(t["parent"][:m] > 0, t["parent"][:m] - 1, 0)  # (the root's parent id 0 maps to the root itself)
                r = np.concatenate([own, own[par]], axis=1)  # a node and the node it hangs on
                return r[np.lexsort(r.T[::-1])]
            assert np.array_equal(rows(tree, n), rows(trace.tree, n)), "nodes and parent links of vehicle %d" % v
        return
    for key in ("x", "y", "yaw", "g", "h"):
        assert np.array_equal(tree[key].view(np.uint64), trace.tree[key][:n].view(np.uint64)), (v, key)
    for key in ("trim", "k", "parent"):
        assert np.array_equal(tree[key], trace.tree[key][:n]), (v, key)


def check_batch(options, mpa, iters, full_tree=True):
    oracle = _oracle()
    options.trace_pops = 1 << 15
    options.max_nodes = 1 << 15
    options.max_vehicles = max(len(iters), 1)
    h = Handle(options)
    h.upload_mpa(mpa)
    gpu = h.plan_batch(iters)
    unbounded = copy.copy(options)  # the reference's tree is unbounded (Tree.m:54-70); the backend's arenas grow on demand
    unbounded.max_nodes = 1 << 30
    _, ref, traces = oracle.plan_batch(unbounded, mpa, iters, trace=True)
    assert_records_equal(gpu, ref, "batch")
    if full_tree:
        for v in range(len(iters)):
            compare_tree_and_pops(h, v, traces[v])
    stats = h.stats()
    h.close()
    # The product configuration (no pop trace): entries known to collide leave the open list on the side and are counted
    # at the end.  Same records (incl. the pop count) and the same tree.
    options.trace_pops = 0
    h2 = Handle(options)
    h2.upload_mpa(mpa)
    gpu2 = h2.plan_batch(iters)
    assert_records_equal(gpu2, ref, "batch without trace")
    if full_tree:
        for v in range(len(iters)):
            compare_tree_and_pops(h2, v, traces[v], pops=False)
    h2.close()
    options.trace_pops = 1 << 15
    return gpu, stats


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_interx_random_road(seed):
    options, mpa, iters = problems.problem_set("interx", seed, 24, Hp=6)
    gpu, stats = check_batch(options, mpa, iters)
    assert stats["nodes_popped"] == int(gpu["n_popped"].sum())
    assert stats["kernel_ms"] > 0


@pytest.mark.parametrize("seed", [11, 12])
def test_sat_random_road(seed):
    options, mpa, iters = problems.problem_set("sat", seed, 24, Hp=5)
    check_batch(options, mpa, iters)


def test_interx_triple_speed_hp8():
    options, mpa, iters = problems.problem_set("interx", 5, 20, Hp=8, mpa_type=MpaType.triple_speed)
    check_batch(options, mpa, iters)


def test_single_vehicle_is_run_optimizer():
    """n == 1 is the literal GraphSearch.run_optimizer call (GraphSearch.m:14-17)."""
    from pdmpc.optimizer import OptimizerInterface

    options, mpa, iters = problems.problem_set("interx", 7, 1, Hp=6)
    options.max_vehicles = 4
    opt = OptimizerInterface.get_optimizer(options)
    info = opt.run_optimizer(1, iters[0], mpa, options, 1)
    ref, _, _ = _oracle().plan_batch(options, mpa, iters)
    assert info.is_exhausted == ref[0].is_exhausted
    assert np.array_equal(info.tree_path, ref[0].tree_path)
    assert np.array_equal(info.y_predicted, ref[0].y_predicted, equal_nan=True)
    for a, b in zip(info.shapes, ref[0].shapes):
        assert np.array_equal(a, b)


def test_circle_closed_loop_c1():
    """BASELINE config 0: 3-vehicle circle, Hp 5, sequential prioritized, 20 steps; GPU plans, oracle checks each level."""
    from pdmpc.config import Config, ScenarioType
    from pdmpc.controller import PrioritizedSequentialController
    from pdmpc.iteration_data import info_from_record
    from pdmpc.mpa import get_mpa
    from pdmpc.scenario import circle_scenario

    oracle = _oracle()
    options = Config(scenario_type=ScenarioType.circle, amount=3, Hp=5, T_end=4, max_vehicles=4, max_nodes=1 << 15)
    mpa = get_mpa(options)
    h = Handle(options)
    h.upload_mpa(mpa)
    n_levels = [0]

    def plan(iters):
        gpu = h.plan_batch(iters)
        _, ref, _ = oracle.plan_batch(options, mpa, iters)
        assert_records_equal(gpu, ref, "step level %d" % n_levels[0])
        n_levels[0] += 1
        return [info_from_record(gpu[i], options.Hp) for i in range(len(iters))]

    ctl = PrioritizedSequentialController(options, circle_scenario(options), mpa, plan)
    for _ in range(options.k_end):
        ctl.step()
    assert n_levels[0] == 3 * options.k_end
    h.close()


def test_exhaustion_and_overflow_status():
    """A vehicle cut by a static obstacle exhausts its open list (GraphSearch.m:57-61); a tiny arena overflows."""
    options, mpa, iters = problems.problem_set("interx", 21, 4, Hp=6)
    for it in iters:
        x, y = it.x0[0], it.x0[1]
        it.obstacles = [problems.rect(x, y, np.pi / 2, 1.2, 0.05)]  # a wall through the vehicle itself: every edge crosses it
    gpu, _ = check_batch(options, mpa, iters)
    assert (gpu["status"] == abi.EXHAUSTED).any()
    options2, mpa2, iters2 = problems.problem_set("interx", 22, 4, Hp=6)
    oracle = _oracle()
    options2.max_nodes = 64
    options2.max_vehicles = 4
    h = Handle(options2)
    h.upload_mpa(mpa2)
    h.set_arena_limit(64)  # growth off: the status itself is under test
    h.allow_overflow = True
    gpu2 = h.plan_batch(iters2)
    unbounded = copy.copy(options2)
    unbounded.max_nodes = 1 << 30
    _, ref2, _ = oracle.plan_batch(unbounded, mpa2, iters2)
    # a search whose reference tree does not fit must say so (never "exhausted"); one that reports a plan must report the
    # reference's (the frontier kernel may create a few nodes the reference does not, so it can overflow a little earlier)
    assert (gpu2["status"] == abi.ARENA_OVERFLOW).any()
    for v in range(len(iters2)):
        if int(ref2[v]["n_expanded"]) > 64:
            assert gpu2[v]["status"] == abi.ARENA_OVERFLOW
        if gpu2[v]["status"] != abi.ARENA_OVERFLOW:
            assert_records_equal(gpu2[v : v + 1], ref2[v : v + 1], "fits")
    h.close()


def test_arena_grows_like_the_reference_tree():
    """The reference's tree is unbounded (Tree.m:54-70).  A handle created with a tiny arena re-plans overflowed calls
    with doubled arenas until every search fits: the records equal the oracle's with an effectively unbounded tree."""
    options, mpa, iters = problems.problem_set("interx", 22, 6, Hp=6)
    oracle = _oracle()
    options.max_nodes = 1 << 22
    _, ref, _ = oracle.plan_batch(options, mpa, iters)
    assert (ref["status"] != abi.ARENA_OVERFLOW).all()
    options.max_nodes = 64
    options.max_vehicles = 8
    h = Handle(options)
    h.upload_mpa(mpa)
    gpu = h.plan_batch(iters)
    assert_records_equal(gpu, ref, "grown arena")
    nodes, regrows = h.arena_nodes()
    assert regrows >= 1 and nodes >= int(ref["n_expanded"].max())
    h.close()


def test_empty_batch_and_errors():
    from pdmpc.backend import BackendError

    options, mpa, iters = problems.problem_set("interx", 3, 2, Hp=6)
    options.max_vehicles = 2
    h = Handle(options)
    with pytest.raises(BackendError):
        h.plan_batch(iters)  # no MPA uploaded yet
    h.upload_mpa(mpa)
    assert len(h.plan_batch([])) == 0
    with pytest.raises(BackendError):
        h.plan_batch(iters * 2)  # larger than max_vehicles
    h.close()


def test_a_failed_pack_leaves_its_bank_empty():
    """pdmpc_pack_batch writes the batch straight into the bank's staging memory (include/pdmpc.h): a pack that fails half-way — the
    second vehicle's trim is out of range — leaves no batch behind (a launch is refused, a fetch returns nothing of the old batch's),
    and the next pack works as if nothing had happened."""
    import copy

    from pdmpc.backend import BackendError

    options, mpa, iters = problems.problem_set("interx", 7, 6, Hp=6)
    _, ref, _ = _oracle().plan_batch(options, mpa, iters)
    h = Handle(options)
    h.upload_mpa(mpa)
    assert_records_equal(h.plan_batch(iters), ref, "before")
    bad = [copy.copy(it) for it in iters]
    bad[1].trim_index = 10 ** 6
    with pytest.raises(BackendError):
        h.pack_batch(bad)
    with pytest.raises(BackendError):
        h.launch()
    assert_records_equal(h.plan_batch(iters), ref, "after")
    h.close()


def test_interx_with_hdv_reachable_sets():
    """are_constraints_satisfied_interx.m:23-31: the HDV soup is a third curve set checked with the normal-offset area."""
    options, mpa, iters = problems.problem_set("interx", 31, 16, Hp=6, n_hdv=2)
    gpu, _ = check_batch(options, mpa, iters)
    _, _, plain = problems.problem_set("interx", 31, 16, Hp=6, n_hdv=0)
    ref_plain = _oracle().plan_batch(options, mpa, plain)[1]
    assert not np.array_equal(gpu["n_popped"], ref_plain["n_popped"])  # the HDV sets actually change some searches


def test_realistic_mpa_two_mask_words_areas_in_hbm():
    """71 trims -> successor masks span two 64-bit words; 527 maneuvers -> the area tables do not fit LDS and are read
    from HBM/L2.  Both code paths differ from the 12/34-trim MPAs."""
    options, mpa, iters = problems.problem_set("interx", 41, 8, Hp=6, mpa_type=MpaType.realistic)
    assert mpa.n_trims > 64
    check_batch(options, mpa, iters)


def test_long_horizon_hp12_and_no_boundary():
    options, mpa, iters = problems.problem_set("interx", 51, 8, Hp=12, with_boundary=False)
    check_batch(options, mpa, iters)


def test_sat_without_obstacles_or_boundary_is_the_greedy_chain():
    """Free space (circle scenario, first vehicle): nothing is ever rejected."""
    options, mpa, iters = problems.problem_set("sat", 61, 6, Hp=5, with_boundary=False)
    for it in iters:
        it.obstacles = []
        it.dynamic_obstacle_area = []
    gpu, _ = check_batch(options, mpa, iters)
    assert (gpu["status"] == abi.OK).all()


@pytest.mark.parametrize("mode,Hp,block_x", [("interx", 6, 0.5), ("interx", 8, 0.9), ("sat", 6, 0.9)])
def test_tied_minimal_keys_fall_back_to_the_binary_heap(mode, Hp, block_x):
    """Mirror-symmetric searches pop tied minima all the time: the search must notice where the order decides, and the result must
    still be the reference's (pop sequence and tree included) — produced by the replay of its tree through the libstdc++-faithful
    heap, in the same launch, by the product's kernel."""
    options = problems.make_options(mode, Hp=Hp)
    mpa = problems.get_mpa(options)
    sym = problems.symmetric_problem(options, mpa, block_x=block_x)
    _, _, traces = _oracle().plan_batch(options, mpa, [sym], trace=True)
    assert problems.tied_pops(traces[0]) > 0  # the construction does what it says
    rng = np.random.default_rng(11)
    ordinary = [problems.road_problem(rng, options, mpa, convex=(mode == "sat")) for _ in range(3)]
    gpu, stats = check_batch(options, mpa, [ordinary[0], sym, ordinary[1], sym, ordinary[2]])
    assert stats["queue_fallbacks"] >= 2 and stats["kernel"] == 2
    assert np.array_equal(gpu[1:2].tobytes(), gpu[3:4].tobytes())


def test_random_road_problems_never_fall_back():
    options, mpa, iters = problems.problem_set("interx", 21, 24, Hp=6)
    _, stats = check_batch(options, mpa, iters)
    assert stats["queue_fallbacks"] == 0


@pytest.mark.parametrize(
    "tuning",
    [
        "speculate=0",
        "waves=8",
        "waves=12,round0=32",  # (sixteen wavefronts are the default of the InterX kernels, twelve of the separating-axis kernel)
        "waves=5,round0=7",
        "helpers=0",
        "helpers=3,share_min=64,tile=32",
        "helpers=200,share_min=64,own_div=2",
        "share_min=64,own_div=64,tile=128",
        "round0=1,ramp=16",
        "round0=200,round=512,ramp=1",
        "round0=1000,round=1000,ramp=1,share_min=64,tile=32",  # rounds of up to two thousand nodes, most of them shared
        "tentative=0",
        "fast_arrival=0",
        "ready=256,round0=300,helpers=0",  # a ready list smaller than what a round wants: the rest waits in far
        "mid_min=0,mid_fill=256",  # every far list feeds near through the mid list, a few hundred entries at a time
        "mid_min=100,mid_fill=1000,round0=300,tentative=0",
        "force_tie=1",  # every search ends on the replay through the libstdc++-faithful heap
        "force_tie=1,round0=300,share_min=64",
        "compact=1",  # the kernel built for two workgroups per CU (8 wavefronts, 80 KB, near list of 1 024 entries: bulk_kernel_compact.hip)
        "compact=1,share_min=64,round0=200,tile=64",
        "compact=1,force_tie=1",
        "compact=1,mid_min=0,mid_fill=256,round0=300",
    ],
)
def test_tuning_switches_do_not_change_results(tuning, monkeypatch):
    """Round sizes, helper workgroups, tile sizes, the open set's lists, expected areas, the early publication, the number of
    wavefronts, and the replay through the binary heap for every search (PDMPC_TUNING, include/pdmpc.h): all give the reference's
    records, pop sequences and trees, with both checkers."""
    monkeypatch.setenv("PDMPC_TUNING", tuning)
    options, mpa, iters = problems.problem_set("interx", 5, 12, Hp=7)
    check_batch(options, mpa, iters)
    options, mpa, iters = problems.problem_set("sat", 6, 6, Hp=6)
    check_batch(options, mpa, iters)


def test_unknown_tuning_key_is_an_error(monkeypatch):
    monkeypatch.setenv("PDMPC_TUNING", "round0=24,no_such_knob=1")
    options = problems.make_options("interx", Hp=6)
    with pytest.raises(Exception):
        Handle(options)


def test_helper_workgroups_take_part_and_change_nothing(monkeypatch):
    """Large rounds are shared with helper workgroups on the CUs the launch leaves idle (bulk_search.hpp, bulk_helper_body): they
    check tiles of the searches' rounds and the records stay those of the oracle — here with the threshold low enough that most rounds
    of the heavier searches are shared, with both checkers, and the statistics say so."""
    monkeypatch.setenv("PDMPC_TUNING", "share_min=64,round0=128,tile=32")
    for mode, seed in (("interx", 2), ("interx", 5), ("sat", 3)):
        options, mpa, iters = problems.problem_set(mode, seed, 24, Hp=6)
        gpu, stats = check_batch(options, mpa, iters)
        assert stats["kernel"] == 2
        assert stats["shared_rounds"] > 0 and stats["helper_checked"] > 0, stats
        assert stats["helper_checked"] < stats["nodes_processed"]


def test_arena_growth_with_shared_rounds(monkeypatch):
    """A search that outgrows its arena ends with PDMPC_ARENA_OVERFLOW and the call is planned again with doubled arenas — from 256
    nodes up, with most rounds shared with helper workgroups, until every search fits; the records are the oracle's."""
    monkeypatch.setenv("PDMPC_TUNING", "share_min=64,round0=128")
    options, mpa, iters = problems.problem_set("interx", 2, 24, Hp=6)
    oracle = _oracle()
    options.max_nodes = 1 << 22
    _, ref, _ = oracle.plan_batch(options, mpa, iters)
    options.max_nodes = 256
    options.max_vehicles = len(iters)
    h = Handle(options)
    h.upload_mpa(mpa)
    gpu = h.plan_batch(iters)
    assert_records_equal(gpu, ref, "grown arena, shared rounds")
    nodes, regrows = h.arena_nodes()
    assert regrows >= 3 and nodes >= int(ref["n_expanded"].max())
    assert h.stats()["shared_rounds"] > 0
    h.close()


@pytest.mark.parametrize("tuning", ["compact=1", "compact=1,helpers=0", "compact=0"])
def test_co_resident_copies_of_a_search_return_the_same_record(tuning, monkeypatch):
    """600 copies of four independent searches in ONE launch: more searches than the chip holds at once, two workgroups per CU with
    the compact kernel.  Every copy must return the record of the first — which is the oracle's.  (Round 6: with two workgroups on a
    CU a wavefront that had been held up read phase B's vote word one pass late and took the workgroup's barriers apart — wrong
    counts, error statuses, memory faults; DESIGN.md section 3.12.  This is the 8-second reproducer.)"""
    monkeypatch.setenv("PDMPC_TUNING", tuning)
    from oracle import oracle
    from pdmpc.backend import Handle

    distinct, copies = 4, 600
    options, mpa, iters = problems.problem_set("interx", 11, distinct, Hp=8)
    options.max_vehicles = copies
    options.max_nodes = 1 << 15
    unbounded = copy.copy(options)
    unbounded.max_nodes = 1 << 30
    _, ref, _ = oracle.plan_batch(unbounded, mpa, iters)
    h = Handle(options)
    h.upload_mpa(mpa)
    batch = [iters[i % distinct] for i in range(copies)]
    try:
        for rep in range(2):
            recs = h.plan_step(batch, [[] for _ in batch], None)
            assert_records_equal(recs[:distinct], ref, "the originals, pass %d" % rep)
            for name in ("status", "n_expanded", "n_popped", "tree_path", "predicted_trims", "y_predicted", "shapes"):
                a = np.asarray(recs[name])
                want = a[np.arange(copies) % distinct]
                same = (a.view(np.uint64) == want.view(np.uint64)) if a.dtype.kind == "f" else (a == want)
                bad = np.argwhere(~same.reshape(copies, -1).all(axis=1)).ravel()
                assert bad.size == 0, "pass %d: field %s of the copies in slots %s differs from the original's" % (rep, name, bad[:8])
        assert h.stats()["safe_replans"] == 0
    finally:
        h.close()
