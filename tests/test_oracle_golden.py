"""Pins the CPU oracle: reference known-answer vectors, committed golden plans, independent re-derivations.

Runs without a GPU.  The reference's own tests for this path are the two SAT polygon cases and the three
lanelet cases of tests/unittests/hlc/intersect_unittest.m; everything else the reference leaves unpinned
(SURVEY.md section 4), so the remaining tests pin the oracle against independent restatements written here.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle

import problems

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KA = json.load(open(os.path.join(GOLDEN, "reference_known_answers.json")))


# ---- reference known answers (intersect_unittest.m) -------------------------------------------------
@pytest.mark.parametrize("case", KA["intersect_sat"])
def test_reference_intersect_sat_vectors(case):
    s1 = np.array(case["shape1"], dtype=np.float64)
    s2 = s1 + np.array(case["shift"], dtype=np.float64).reshape(2, 1)
    assert oracle.intersect_sat(s1, s2) == case["expected"]
    assert oracle.intersect_sat(s2, s1) == case["expected"]


@pytest.mark.parametrize("case", KA["intersect_lanelets"])
def test_reference_intersect_lanelets_vectors(case):
    lanelet = np.array(KA["lanelet_1_rows_rx_ry_lx_ly_cx_cy"], dtype=np.float64)
    assert oracle.intersect_lanelets(np.array(case["shape"], dtype=np.float64), lanelet) == case["expected"]


def test_reference_priority_queue_tie_order():
    pq = KA["priority_queue"]
    ids = [p[0] for p in pq["push"]]
    keys = [p[1] for p in pq["push"]]
    n = len(ids)
    out = oracle.pq_script([0] * n + [1] * (n + 1), ids + [0] * (n + 1), keys + [0.0] * (n + 1))
    assert out.tolist() == pq["pops"] + [-1]  # empty pop returns -1 (priority_queue_interface_mex.cpp:87-93)


# ---- independent model of libstdc++'s heap ---------------------------------------
class HeapModel:
    def __init__(self):
        self.a = []

    def push(self, id_, key):
        a = self.a
        a.append(None)
        hole = len(a) - 1
        while hole > 0 and a[(hole - 1) // 2][1] > key:
            a[hole] = a[(hole - 1) // 2]
            hole = (hole - 1) // 2
        a[hole] = (id_, key)

    def pop(self):
        a = self.a
        if not a:
            return -1
        top = a[0]
        v = a.pop()
        n = len(a)
        if n == 0:
            return top[0]
        hole = child = 0
        while child < (n - 1) // 2:
            child = 2 * (child + 1)
            if a[child][1] > a[child - 1][1]:
                child -= 1
            a[hole] = a[child]
            hole = child
        if n % 2 == 0 and child == (n - 2) // 2:
            child = 2 * (child + 1)
            a[hole] = a[child - 1]
            hole = child - 1
        while hole > 0 and a[(hole - 1) // 2][1] > v[1]:
            a[hole] = a[(hole - 1) // 2]
            hole = (hole - 1) // 2
        a[hole] = v
        return top[0]


@pytest.mark.parametrize("seed", range(6))
def test_priority_queue_random_scripts_with_many_ties(seed):
    rng = np.random.default_rng(seed)
    n = 3000
    ops = (rng.random(n) < 0.42).astype(np.int32)
    ids = np.arange(1, n + 1, dtype=np.int32)
    keys = rng.integers(0, 12, n).astype(np.float64) * 0.25  # few distinct keys -> ties everywhere
    got = oracle.pq_script(ops, ids, keys)
    model = HeapModel()
    want = []
    for o, i, k in zip(ops, ids, keys):
        if o == 0:
            model.push(int(i), float(k))
        else:
            want.append(model.pop())
    assert got.tolist() == want


# ---- InterX / SAT against independent numpy restatements -------------------------------------------------
def interx_numpy(L1, L2):
    """InterX.m:63-76 transcribed with numpy broadcasting (same operation order)."""
    x1, y1 = L1[0][:, None], L1[1][:, None]
    x2, y2 = L2[0][None, :], L2[1][None, :]
    dx1, dy1 = np.diff(x1, axis=0), np.diff(y1, axis=0)
    dx2, dy2 = np.diff(x2, axis=1), np.diff(y2, axis=1)
    S1 = dx1 * y1[:-1] - dy1 * x1[:-1]
    S2 = dx2 * y2[:, :-1] - dy2 * x2[:, :-1]
    with np.errstate(invalid="ignore"):
        A = dx1 * y2 - dy1 * x2
        C1 = (A[:, :-1] - S1) * (A[:, 1:] - S1) < 0
        B = (y1 * dx2 - x1 * dy2).T
        C2 = ((B[:, :-1] - S2.T) * (B[:, 1:] - S2.T) < 0).T
    return bool(np.any(C1 & C2))


def test_interx_matches_numpy_restatement():
    rng = np.random.default_rng(5)
    hits = 0
    for _ in range(400):
        n1, n2 = int(rng.integers(2, 8)), int(rng.integers(2, 40))
        L1 = rng.uniform(-1, 1, (2, n1))
        L2 = rng.uniform(-1.5, 1.5, (2, n2)) * rng.uniform(0.1, 1.0)
        for c in rng.integers(0, n2, int(rng.integers(0, 4))):
            L2[:, c] = np.nan  # polygon separators
        want = interx_numpy(L1, L2)
        assert oracle.interx(L1, L2) == want
        hits += want
    assert 50 < hits < 350  # both outcomes exercised


def test_interx_edge_semantics():
    sq = np.array([[0, 1, 1, 0, 0], [0, 0, 1, 1, 0.0]])
    assert not oracle.interx(sq, sq * 0.5 + 0.25)  # a shape wholly inside another is NOT detected (Config.m:75-84)
    assert not oracle.interx(sq, sq + np.array([[1.0], [0.0]]))  # touching edges: strict < 0 (InterX.m:72-73)
    assert oracle.interx(sq, sq + 0.5)
    assert not oracle.interx(sq, np.array([[np.nan, np.nan], [np.nan, np.nan]]))
    assert not oracle.interx(sq, np.zeros((2, 1)))  # single column: diff() is empty


# ---- InterX / SAT against exact geometry (an independent specification, not a second reading of the .m files) -------
def _crosses_exactly(a0, a1, b0, b1):
    """Do the open segments a0a1 and b0b1 cross in a single interior point?  Exact rational arithmetic on the doubles."""
    from fractions import Fraction as Fr

    def orient(p, q, r):
        return (Fr(q[0]) - Fr(p[0])) * (Fr(r[1]) - Fr(p[1])) - (Fr(q[1]) - Fr(p[1])) * (Fr(r[0]) - Fr(p[0]))

    o1, o2, o3, o4 = orient(a0, a1, b0), orient(a0, a1, b1), orient(b0, b1, a0), orient(b0, b1, a1)
    margin = min(abs(o1), abs(o2), abs(o3), abs(o4))
    return (o1 * o2 < 0 and o3 * o4 < 0), float(margin)


def test_interx_is_proper_segment_crossing_in_exact_arithmetic():
    """InterX.m's header: the intersection points of two curves.  For curves in general position that is: some segment of
    one properly crosses some segment of the other — decided here with exact rationals, independently of how InterX.m:63-76
    arranges its products.  (Pairs with an orientation within 1e-12 of zero are skipped: there rounding may decide.)"""
    rng = np.random.default_rng(77)
    checked = hits = 0
    for _ in range(250):
        n1, n2 = int(rng.integers(2, 8)), int(rng.integers(2, 12))
        L1 = rng.uniform(-1, 1, (2, n1))
        L2 = rng.uniform(-1.2, 1.2, (2, n2)) * rng.uniform(0.2, 1.0)
        want, generic = False, True
        for i in range(n1 - 1):
            for j in range(n2 - 1):
                c, m = _crosses_exactly(L1[:, i], L1[:, i + 1], L2[:, j], L2[:, j + 1])
                want = want or c
                generic = generic and m > 1e-12
        if not generic:
            continue
        assert oracle.interx(L1, L2) == want
        checked += 1
        hits += want
    assert checked > 200 and 30 < hits < checked - 30


def test_intersect_sat_is_convex_overlap_in_exact_arithmetic():
    """intersect_sat.m: two convex polygons collide iff no edge normal of either separates them (touching counts as
    colliding: the gap must be > 0).  Decided with exact rationals on unnormalised normals (a positive scale does not move
    a sign); cases with a gap within 1e-12 of zero are skipped."""
    from fractions import Fraction as Fr

    def convex(rng):
        n = int(rng.integers(3, 7))
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        r = rng.uniform(0.2, 0.6)
        c = rng.uniform(-0.6, 0.6, 2)
        return np.stack([c[0] + r * np.cos(ang), c[1] + r * np.sin(ang)])

    def separated(p, q):
        best = None
        for poly in (p, q):
            n = poly.shape[1]
            for i in range(n):
                ex, ey = Fr(poly[0, (i + 1) % n]) - Fr(poly[0, i]), Fr(poly[1, (i + 1) % n]) - Fr(poly[1, i])
                ax, ay = -ey, ex
                d1 = [ax * Fr(p[0, k]) + ay * Fr(p[1, k]) for k in range(p.shape[1])]
                d2 = [ax * Fr(q[0, k]) + ay * Fr(q[1, k]) for k in range(q.shape[1])]
                gap = max(min(d1) - max(d2), min(d2) - max(d1))
                best = gap if best is None or gap > best else best
        return best > 0, float(abs(best))

    rng = np.random.default_rng(78)
    checked = hits = 0
    for _ in range(250):
        p, q = convex(rng), convex(rng)
        sep, m = separated(p, q)
        if m < 1e-12:
            continue
        assert oracle.intersect_sat(p, q) == (not sep)
        checked += 1
        hits += not sep
    assert checked > 200 and 30 < hits < checked - 30


def sat_numpy(s1, s2):
    def a_b(p, q):
        e = np.diff(np.hstack([p, p[:, :1]]), axis=1)
        ax = np.vstack([-e[1], e[0]])
        with np.errstate(invalid="ignore", divide="ignore"):
            nax = ax / np.sqrt(ax[0] * ax[0] + ax[1] * ax[1])
            d1 = nax[0][:, None] * p[0][None, :] + nax[1][:, None] * p[1][None, :]
            d2 = nax[0][:, None] * q[0][None, :] + nax[1][:, None] * q[1][None, :]
            ok = ~np.isnan(d1[:, 0])
            sep = (np.nanmin(d1[ok], axis=1) - np.nanmax(d2[ok], axis=1) > 0) | (np.nanmin(d2[ok], axis=1) - np.nanmax(d1[ok], axis=1) > 0)
        return not sep.any()

    return a_b(s1, s2) and a_b(s2, s1)


def test_sat_matches_numpy_restatement_including_closed_polygons():
    rng = np.random.default_rng(9)
    outcomes = set()
    for _ in range(300):
        a = problems.rect(rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-3, 3), rng.uniform(0.1, 1), rng.uniform(0.1, 0.6))
        b = problems.rect(rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-3, 3), rng.uniform(0.1, 1), rng.uniform(0.1, 0.6))
        want = sat_numpy(a, b)  # closed rectangles: the repeated first vertex gives a NaN axis that must be ignored
        assert oracle.intersect_sat(a, b) == want
        outcomes.add(want)
    assert outcomes == {True, False}


def test_lanelet_boundary_empty_and_aabb():
    shape = problems.rect(0.0, 0.0, 0.3, 0.3, 0.1)
    empty = np.zeros((2, 0))
    assert not oracle.intersect_lanelet_boundary(shape, empty, empty)  # circle scenario: cell(nVeh,3) of [] (IterationData.m:52)
    left = np.array([[-1.0, 1.0], [0.2, 0.2]])
    right = np.array([[-1.0, 1.0], [-0.2, -0.2]])
    assert not oracle.intersect_lanelet_boundary(shape, left, right)
    assert oracle.intersect_lanelet_boundary(shape, left - np.array([[0], [0.15]]), right)


# ---- committed golden plans ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["interx_single_hp6", "sat_single_hp5", "interx_triple_hp8"])
def test_oracle_reproduces_golden_plans(name):
    from pdmpc.config import MpaType

    g = np.load(os.path.join(GOLDEN, "oracle_plans_%s.npz" % name))
    options, mpa, iters = problems.problem_set(str(g["mode"]), int(g["seed"]), int(g["count"]), Hp=int(g["Hp"]), mpa_type=MpaType[str(g["mpa_type"])])
    options.max_nodes = 1 << 15
    _, recs, traces = oracle.plan_batch(options, mpa, iters, trace=True)
    assert np.array_equal(recs.view(np.uint8).reshape(len(iters), -1), g["records"])
    for i, t in enumerate(traces):
        m = min(64, len(t.pops))
        assert np.array_equal(t.pops[:m], g["first_pops"][i, :m])
        assert len(t.tree["x"]) == g["tree_sizes"][i]


@pytest.mark.parametrize("name", ["sampled_interx_hp6", "sampled_sat_hp8"])
def test_oracle_reproduces_golden_sampled_plans(name):
    g = np.load(os.path.join(GOLDEN, "oracle_plans_%s.npz" % name))
    options, mpa, iters = problems.problem_set(str(g["mode"]), int(g["seed"]), int(g["count"]), Hp=int(g["Hp"]))
    _, recs = oracle.plan_batch_sampled(options, mpa, iters, g["rng_seeds"].tolist())
    assert np.array_equal(recs.view(np.uint8).reshape(len(iters), -1), g["records"])


# ---- structural invariants derivable from the reference code ---------------------------
def test_search_invariants():
    options, mpa, iters = problems.problem_set("interx", 77, 16, Hp=6)
    infos, recs, traces = oracle.plan_batch(options, mpa, iters, trace=True)
    stop = set(mpa.trims_stop)
    for info, it, tr in zip(infos, iters, traces):
        assert tr.pops[0] == 1  # the root is popped first (GraphSearch.m:45-46)
        assert info.n_expanded == len(tr.tree["x"])  # n_expanded is the tree size (GraphSearch.m:58,89)
        if info.is_exhausted:
            assert np.isnan(info.y_predicted).all()  # ControlResultsInfo.m:40
            continue
        assert info.tree_path[0] == 1
        path = info.tree_path - 1
        assert np.array_equal(tr.tree["parent"][path[1:]], info.tree_path[:-1])  # Tree.path_to_root
        assert np.array_equal(info.predicted_trims, tr.tree["trim"][path[1:]])  # GraphSearch.m:86
        assert int(info.predicted_trims[-1]) in stop  # recursive feasibility: last trim is an equilibrium (MPA :238-250)
        assert np.array_equal(tr.tree["k"][path], np.arange(options.Hp + 1))
        g = tr.tree["g"][path]
        assert np.all(np.diff(g) >= 0)  # cost-to-come accumulates squared distances (expand_node.m:61)


def test_unobstructed_search_equals_obstructed_search_when_obstacles_are_far():
    """Obstacles no popped edge touches leave the pop sequence unchanged — the property the single-launch step
    planner relies on when it appends predecessor areas on the device."""
    import copy

    options, mpa, iters = problems.problem_set("interx", 78, 6, Hp=6)
    far = [copy.copy(it) for it in iters]
    for it in far:
        it.obstacles = list(it.obstacles) + [problems.rect(50.0, 50.0, 0.3, 0.3, 0.2)]
    _, r1, t1 = oracle.plan_batch(options, mpa, iters, trace=True)
    _, r2, t2 = oracle.plan_batch(options, mpa, far, trace=True)
    for a, b in zip(t1, t2):
        assert np.array_equal(a.pops, b.pops)
    assert np.array_equal(r1["y_predicted"], r2["y_predicted"], equal_nan=True)


def test_symmetric_problem_really_ties():
    """tests/problems.symmetric_problem (used by the GPU fallback tests) pops tied minimal keys in the reference algorithm."""
    import problems

    options = problems.make_options("interx", Hp=6)
    mpa = problems.get_mpa(options)
    _, _, traces = oracle.plan_batch(options, mpa, [problems.symmetric_problem(options, mpa, block_x=0.5)], trace=True)
    assert problems.tied_pops(traces[0]) > 50
    rng = np.random.default_rng(3)
    _, _, traces = oracle.plan_batch(options, mpa, [problems.road_problem(rng, options, mpa)], trace=True)
    assert problems.tied_pops(traces[0]) == 0


def test_mt19937ar_stream_is_numpys():
    """MATLAB's RandStream('mt19937ar', Seed = s) + rand draw the reference generator's 53-bit doubles; numpy's RandomState
    implements the same routines (init_genrand, genrand_res53): e.g. seed 42 -> 0.3745, 0.9507, 0.7320 in both worlds."""
    for seed in (1, 5, 42, 2024):
        assert np.array_equal(oracle.mt19937_doubles(seed, 3000), np.random.RandomState(seed).random_sample(3000))
    assert np.allclose(oracle.mt19937_doubles(42, 3), [0.3745401188473625, 0.9507143064099162, 0.7319939418114051], rtol=0, atol=0)


def test_sampled_optimizer_structure():
    """MonteCarloTreeSearch.m invariants: at most 250 + Hp - 1 expansions, the chosen descent starts at the root, its
    cost is the sum of squared distances to the reference points and never beats the optimal search's."""
    import problems

    options, mpa, iters = problems.problem_set("interx", 3, 8, Hp=6)
    infos, recs = oracle.plan_batch_sampled(options, mpa, iters, list(range(10, 18)))
    opt_infos, opt_recs, _ = oracle.plan_batch(options, mpa, iters)
    for i, it in enumerate(iters):
        if recs[i]["status"] != 0:
            continue
        assert recs[i]["n_expanded"] <= 250 + options.Hp - 1
        assert recs[i]["tree_path"][0] == 1
        yp = recs[i]["y_predicted"][: options.Hp]
        cost = 0.0
        for k in range(options.Hp):
            d = np.hypot(yp[k, 0] - it.reference_trajectory_points[k, 0], yp[k, 1] - it.reference_trajectory_points[k, 1])
            cost += d * d
        assert abs(cost - recs[i]["path_nodes"][options.Hp][4]) < 1e-12
        if opt_recs[i]["status"] == 0:
            assert recs[i]["path_nodes"][options.Hp][4] >= opt_recs[i]["path_nodes"][options.Hp][4] - 1e-12


def test_native_step_loop_equals_the_python_level_loop():
    """oracle_plan_step (level loop + hand-over of solved areas + fallback publication in C++, on a persistent thread pool: what
    bench.py times as cpu_baseline) against oracle.plan_step (the same loop in Python around oracle_plan_batch): identical records
    over a closed loop, single- and multi-threaded."""
    from oracle import oracle
    from pdmpc.config import Config, ScenarioType
    from pdmpc.controller import PrioritizedSequentialController
    from pdmpc.iteration_data import info_from_record
    from pdmpc.mpa import get_mpa
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=12, Hp=5, max_nodes=1 << 30)
    mpa = get_mpa(options)
    sc = commonroad_scenario(options, seed=2)
    ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc))
    seen = []

    def plan_step(prob):
        a, _ = oracle.plan_step(options, mpa, prob)
        b, ms, thr = oracle.plan_step_native(options, mpa, prob, n_threads=1)
        c, _, thr4 = oracle.plan_step_native(options, mpa, prob, n_threads=4)
        assert a.tobytes() == b.tobytes() == c.tobytes()
        assert ms > 0 and 0.9 < thr <= 1.0 and 0.9 < thr4 <= 4.0
        seen.append(len(a))
        return [info_from_record(a[i], options.Hp) for i in range(len(a))]

    for _ in range(6):
        ctl.step(plan_step=plan_step)
    assert len(seen) == 6
