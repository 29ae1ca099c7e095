"""Host-side logic: MPA tables, level assignment, reference sampling, the step driver (CPU only)."""
import numpy as np
import pytest

from pdmpc.config import Config, MpaType, ScenarioType
from pdmpc.controller import PrioritizedSequentialController, directed_coupling_from_priorities, kahn
from pdmpc.mpa import get_mpa
from pdmpc.reference_trajectory import get_reference_trajectory, sample_reference_trajectory


@pytest.mark.parametrize("mpa_type,n_trims,n_trans,max_dist", [(MpaType.single_speed, 12, 54, 1), (MpaType.triple_speed, 34, 160, 3)])
def test_mpa_table_shapes(mpa_type, n_trims, n_trans, max_dist):
    """Counts derived from choose_trims.m:12-84 (n = 12 / 34, T = 54 / 160)."""
    o = Config(mpa_type=mpa_type, Hp=6)
    m = get_mpa(o)
    assert m.n_trims == n_trims
    assert sum(x is not None for row in m.maneuvers for x in row) == n_trans
    assert m.distance_to_equilibrium.max() == max_dist
    T = m.transition_matrix_single
    assert T.shape == (n_trims, n_trims, 6)
    # recursive feasibility (MotionPrimitiveAutomaton.m:238-250): the last step may only enter a zero-speed trim
    last = np.nonzero(T[:, :, -1].any(axis=0))[0] + 1
    assert set(last) <= set(m.trims_stop)
    # every allowed transition has a maneuver with closed areas of equal column count
    for i in range(n_trims):
        for j in range(n_trims):
            if T[i, j, 0]:
                man = m.maneuvers[i][j]
                assert man.area.shape == man.area_without_offset.shape == man.area_large_offset.shape
                assert np.array_equal(man.area[:, 0], man.area[:, -1])


def test_mpa_convex_vs_non_convex_columns():
    """generate_maneuver.m:74-101: straight 5 columns; turns 6 (convex) or 7 (non-convex)."""
    conv = get_mpa(Config(scenario_type=ScenarioType.circle, Hp=5))
    nonc = get_mpa(Config(scenario_type=ScenarioType.commonroad, Hp=5))
    cols_c = {m.area.shape[1] for row in conv.maneuvers for m in row if m is not None}
    cols_n = {m.area.shape[1] for row in nonc.maneuvers for m in row if m is not None}
    assert cols_c == {5, 6} and cols_n == {5, 7}


def test_straight_maneuver_is_exact_kinematics():
    m = get_mpa(Config(Hp=5))
    man = m.maneuvers[6][6]  # trim 7: steering 0, speed 0.8
    assert abs(man.dx - 0.8 * 0.2) < 1e-12 and man.dy == 0 and man.dyaw == 0


def test_kahn_levels():
    A = np.zeros((5, 5), dtype=int)
    A[0, 1] = A[0, 2] = A[1, 3] = A[2, 3] = 1  # 0 -> {1,2} -> 3 ; 4 isolated
    assert kahn(A).tolist() == [1, 2, 2, 3, 1]
    with pytest.raises(ValueError):
        kahn(np.array([[0, 1], [1, 0]]))


def test_directed_coupling_from_priorities():
    adj = np.ones((3, 3), dtype=int) - np.eye(3, dtype=int)
    d = directed_coupling_from_priorities(adj, [2, 1, 3])  # vehicle 2 plans first
    assert d.tolist() == [[0, 0, 1], [1, 0, 1], [0, 0, 0]]


def test_reference_sampling_on_a_straight_chord():
    path = np.array([[0.0, 0.0], [4.0, 0.0]])
    pts, idx, cpi = sample_reference_trajectory(4, path, 0.5, 0.3, [0.1, 0.2, 0.2, 0.2])
    assert np.allclose(pts, [[0.6, 0], [0.8, 0], [1.0, 0], [1.2, 0]])
    m = get_mpa(Config(Hp=5, scenario_type=ScenarioType.circle))
    p, _, v, _ = get_reference_trajectory(m, path, 0.8, 0.0, 0.0, 1, 0.2)
    # from standstill the first step covers half a full step (get_reference_trajectory.m:31-36)
    assert np.allclose(p[:, 0], [0.08, 0.24, 0.40, 0.56, 0.72]) and np.all(v == 0.8)


def test_reference_sampling_wraps_around_a_loop():
    sq = np.array([[0, 0], [1, 0], [1, 1], [0, 1], [0, 0]], dtype=float)
    pts, idx, _ = sample_reference_trajectory(6, sq, 0.9, 0.0, [0.3] * 6)
    assert np.allclose(pts[0], [1.0, 0.2]) and np.allclose(pts[1], [1.0, 0.5])
    assert np.all((pts >= -1e-12) & (pts <= 1 + 1e-12))


def _oracle_level(options, mpa):
    from oracle import oracle

    return lambda iters: oracle.plan_batch(options, mpa, iters)[0]


def test_circle_closed_loop_level_mode_equals_step_mode():
    """The single-launch formulation (predecessor lists) and the level loop produce the same closed loop."""
    from oracle import oracle
    from pdmpc.iteration_data import info_from_record
    from pdmpc.scenario import circle_scenario

    o = Config(scenario_type=ScenarioType.circle, amount=4, Hp=5, T_end=3)
    mpa = get_mpa(o)
    a = PrioritizedSequentialController(o, circle_scenario(o), mpa, _oracle_level(o, mpa))
    b = PrioritizedSequentialController(o, circle_scenario(o), mpa, None)

    def plan_step(prob):
        recs, _ = oracle.plan_step(o, mpa, prob)
        return [info_from_record(recs[i], o.Hp) for i in range(len(recs))]

    for _ in range(o.k_end):
        ia = a.step()
        ib = b.step(plan_step=plan_step)
        for x, y in zip(ia, ib):
            assert np.array_equal(x.y_predicted, y.y_predicted, equal_nan=True)
            assert x.n_expanded == y.n_expanded and x.is_exhausted == y.is_exhausted
    # vehicles reached the far side without touching: all pairwise distances stay above the body width
    pos = np.array([[m.x, m.y] for m in a.meas])
    assert np.linalg.norm(pos[0] - pos[2]) > 0.1


def test_road_network_scenario_and_boundaries():
    from pdmpc.road_network import boundary_provider, commonroad_scenario, get_reference_lanelets_loop, lab_map

    m = lab_map()
    assert m.n == 104 and all(l.shape[1] == 6 for l in m.lanelets)
    assert get_reference_lanelets_loop(10)[0] == 10  # path 10 starts at lanelet 10 of loop 2
    o = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=8)
    sc = commonroad_scenario(o, seed=1)
    assert len(sc.vehicles) == 20
    starts = {(round(v.x_start, 6), round(v.y_start, 6)) for v in sc.vehicles}
    assert len(starts) == 20  # distinct start poses
    mpa = get_mpa(o)
    ctl = PrioritizedSequentialController(o, sc, mpa, _oracle_level(o, mpa), coupling="distance", boundary_provider=boundary_provider(sc))
    infos = ctl.step()
    it = ctl.last_iters[0]
    left, right = it.predicted_lanelet_boundary
    assert left.shape[0] == 2 and right.shape[0] == 2 and left.shape[1] >= 12
    assert not any(i.is_exhausted for i in infos)
    # the planned first pose stays between the boundaries: distance to both polylines below the lane width
    for i, info in enumerate(infos):
        p = info.y_predicted[:2, 0]
        l, r = ctl.last_iters[i].predicted_lanelet_boundary
        assert np.min(np.hypot(*(l - p[:, None]))) < 0.45 and np.min(np.hypot(*(r - p[:, None]))) < 0.45


def test_tiled_scenario_for_large_configs():
    from pdmpc.road_network import commonroad_scenario

    o = Config(scenario_type=ScenarioType.commonroad, amount=128, Hp=8)
    sc = commonroad_scenario(o, seed=2, tiles=7)
    assert len(sc.vehicles) == 128
    xs = np.array([v.x_start for v in sc.vehicles])
    assert xs.max() > 20.0  # vehicles spread over translated copies of the map


def test_coloring_prioritizer_levels():
    """ColoringPrioritizer.m: adjacent vertices never share a level, edges run from earlier to later levels, the number
    of levels is at most max degree + 1, and the level whose vertex has the most edges plans first."""
    from pdmpc.controller import kahn
    from pdmpc.prioritizer import coloring_directed_coupling, topological_coloring

    rng = np.random.default_rng(4)
    for n, p in ((8, 0.3), (20, 0.15), (40, 0.1), (12, 0.0), (6, 1.0)):
        A = rng.random((n, n)) < p
        A = np.triu(A, 1)
        A = A | A.T
        color, L = topological_coloring(A)
        for i, j in zip(*np.nonzero(A)):
            assert color[i] != color[j]
        directed, level = coloring_directed_coupling(A)
        assert level.max() <= A.sum(axis=0).max() + 1
        for i, j in zip(*np.nonzero(A)):
            assert directed[i, j] != directed[j, i]  # every coupling keeps exactly one direction
            if directed[i, j]:
                assert level[i] < level[j]
        lv = kahn(directed.astype(np.int64))
        assert lv.max() <= level.max()
        if A.any():
            busiest = int(np.argmax(A.sum(axis=0)))
            assert level[busiest] == 1
    # a path graph 0-1-2-3 needs two colours; constant priorities would chain all four vehicles
    A = np.zeros((4, 4), dtype=bool)
    for i in range(3):
        A[i, i + 1] = A[i + 1, i] = True
    directed, level = coloring_directed_coupling(A)
    assert sorted(set(level.tolist())) == [1, 2]
    assert kahn(directed.astype(np.int64)).max() == 2


def test_controller_with_coloring_priorities_plans_in_fewer_levels():
    from pdmpc.config import Config, ScenarioType
    from pdmpc.controller import PrioritizedSequentialController
    from pdmpc.mpa import get_mpa
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    options = Config(scenario_type=ScenarioType.commonroad, amount=20, Hp=6)
    mpa = get_mpa(options)
    sc = commonroad_scenario(options, seed=1)
    levels = {}
    for strategy in ("constant", "coloring"):
        ctl = PrioritizedSequentialController(options, sc, mpa, None, coupling="distance", boundary_provider=boundary_provider(sc), priority_strategy=strategy)
        prob = ctl.build_step_problem()
        levels[strategy] = len(prob["level_sizes"])
        for s, ps in enumerate(prob["preds"]):
            assert all(prob["levels"][q] < prob["levels"][s] for q in ps)
    assert levels["coloring"] <= levels["constant"]


def _random_dag(rng, n, p):
    A = np.triu(rng.random((n, n)) < p, 1)
    perm = rng.permutation(n)
    return A[np.ix_(perm, perm)].astype(np.int64)


def test_greedy_cutter_bounds_the_number_of_levels():
    """GreedyCutter.m:5-86: the sequential couplings are a subset of the weighed couplings, fit into max_num_CLs levels,
    nothing is cut when the bound is loose, everything when it is 1; heavier couplings are kept first."""
    from pdmpc.controller import kahn
    from pdmpc.grouping import constant_weight, greedy_cut

    rng = np.random.default_rng(11)
    for n, p in ((6, 0.6), (12, 0.4), (25, 0.25), (40, 0.1)):
        D = _random_dag(rng, n, p)
        W = D * rng.random((n, n))
        depth = int(kahn(D).max())
        assert not greedy_cut(W, 1).any()
        assert np.array_equal(greedy_cut(W, depth), D != 0)
        assert np.array_equal(greedy_cut(constant_weight(D), 99), D != 0)
        for bound in range(2, depth):
            S = greedy_cut(W, bound)
            assert not (S & (D == 0)).any()
            assert kahn(S.astype(np.int64)).max() <= bound
            # maximal: no cut edge could still be added without exceeding the bound
            for a, b in zip(*np.nonzero((D != 0) & ~S)):
                T = S.copy()
                T[a, b] = True
                assert kahn(T.astype(np.int64)).max() > bound
    # chain 0 -> 1 -> 2 with bound 2: only one of the two couplings can stay sequential, the heavier one wins
    W = np.zeros((3, 3))
    W[0, 1], W[1, 2] = 0.2, 0.9
    S = greedy_cut(W, 2)
    assert S[1, 2] and not S[0, 1]
    W[0, 1], W[1, 2] = 0.9, 0.2
    S = greedy_cut(W, 2)
    assert S[0, 1] and not S[1, 2]
    # equal weights: column-major order of find() decides (edge into column 1 is visited before the one into column 2)
    W[0, 1] = W[1, 2] = 0.5
    S = greedy_cut(W, 2)
    assert S[0, 1] and not S[1, 2]


def test_weighers():
    from pdmpc.grouping import constant_weight, distance_weight, mt19937ar_doubles, random_weight

    D = np.array([[0, 1, 1], [0, 0, 1], [0, 0, 0]])
    assert np.array_equal(constant_weight(D), D * 0.5)
    x0 = np.array([[0.0, 0.0, 0, 0], [3.0, 4.0, 0, 0], [0.0, 1.0, 0, 0]])
    W = distance_weight(D, x0, max_mpa_speed=1.0, dt_seconds=0.5, Hp=10)  # max distance 10
    assert W[0, 1] == 1 - 5.0 / 10 and W[0, 2] == 1 - 1.0 / 10 and W[1, 0] == 0
    R = random_weight(D, 7)
    r = mt19937ar_doubles(7, 3)
    assert [R[0, 1], R[0, 2], R[1, 2]] == list(r)  # column-major order of find(): (1,2) (1,3) (2,3)
    # mt19937ar known answer: genrand_res53 after init_genrand(5489) starts 0.8147236863931789 (MATLAB's rand default)
    assert mt19937ar_doubles(5489, 1)[0] == 0.8147236863931789


def test_host_sat_matches_the_oracle():
    from oracle import oracle
    from pdmpc.prioritizer import intersect_sat

    rng = np.random.default_rng(5)
    n_hit = 0
    for _ in range(300):
        polys = []
        for _ in range(2):
            k = int(rng.integers(3, 7))
            ang = np.sort(rng.random(k) * 2 * np.pi)
            c = rng.random(2) * 3
            r = 0.3 + rng.random()
            polys.append(np.stack([c[0] + r * np.cos(ang), c[1] + r * np.sin(ang)]))
        got = intersect_sat(*polys)
        assert got == bool(oracle.intersect_sat(*polys))
        n_hit += got
    assert 30 < n_hit < 270


def test_random_and_fca_prioritizers():
    from pdmpc.controller import directed_coupling_from_priorities, kahn
    from pdmpc.prioritizer import calculate_yaw, fca_priorities, random_priorities

    p = random_priorities(9, 3)
    assert sorted(p) == list(range(1, 10)) and p == random_priorities(9, 3) and p != random_priorities(9, 4)
    yaw = calculate_yaw(np.array([[0.0, 0], [1, 0], [1, 1], [0, 1]]))
    assert np.allclose(yaw, [0, np.pi / 4, 3 * np.pi / 4, np.pi])
    # two vehicles crossing at the origin, a third far away: the crossing pair collects collisions and goes first
    Hp = 5
    t = np.linspace(-0.4, 0.4, Hp)
    refs = [np.stack([t, 0 * t], axis=1), np.stack([0 * t, t], axis=1), np.stack([t + 5, 0 * t + 5], axis=1)]
    A = np.ones((3, 3)) - np.eye(3)
    prio, collisions = fca_priorities(A, refs, 0.22, 0.10, 0.01)
    assert collisions[2] == 0 and collisions[0] == collisions[1] > 0
    assert prio == [1, 2, 3]
    refs = [refs[2], refs[0], refs[1]]
    prio, collisions = fca_priorities(A, refs, 0.22, 0.10, 0.01)
    assert prio == [2, 3, 1]  # positions of the descending sort, as the reference passes them on
    d = directed_coupling_from_priorities(A, prio)
    assert kahn(d).max() == 3


def test_controller_cuts_to_max_num_CLs():
    """With options.max_num_CLs the step has at most that many levels; cut couplings turn into literal obstacles from the
    predecessor's previous plan (PrioritizedController.m:409-447), so they only exist from the second step on."""
    from pdmpc.config import Config, ScenarioType
    from pdmpc.controller import PrioritizedSequentialController
    from pdmpc.mpa import get_mpa
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    results = {}
    for bound in (99, 2):
        options = Config(scenario_type=ScenarioType.commonroad, amount=12, Hp=5, max_num_CLs=bound)
        mpa = get_mpa(options)
        sc = commonroad_scenario(options, seed=1)
        ctl = PrioritizedSequentialController(options, sc, mpa, _oracle_level(options, mpa), coupling="distance", boundary_provider=boundary_provider(sc))
        n_levels = []
        for _ in range(3):
            ctl.step()
            n_levels.append(int(ctl.last_levels.max()))
        results[bound] = n_levels
    assert max(results[2]) <= 2
    assert max(results[99]) > 2  # otherwise the scenario does not exercise the cutter


def test_fallback_spreads_to_coupled_vehicles():
    """HighLevelController.handle_others_fallback / PrioritizedController.check_others_fallback
    (HighLevelController.m:449-463, PrioritizedController.m:623-676): when a moving vehicle's search is exhausted it takes
    its shifted previous plan, and so does every vehicle it reaches in the coupling graph without the fallback vehicle's
    own outgoing sequential edges (its successors planned against the fallback already)."""
    from oracle import oracle
    from pdmpc.scenario import circle_scenario

    options = Config(scenario_type=ScenarioType.circle, amount=4, Hp=5, max_nodes=1 << 15)
    mpa = get_mpa(options)
    force = {"on": False}

    def plan(iters):
        infos, _, _ = oracle.plan_batch(options, mpa, iters)
        if force["on"] and ctl_box[0].current_level_members == [1]:
            infos[0].is_exhausted = True  # vehicle 2 (moving): needs the fallback
        return infos

    ctl_box = [None]

    class Ctl(PrioritizedSequentialController):
        def step(self, plan_step=None):
            return super().step(plan_step)

    ctl = Ctl(options, circle_scenario(options), mpa, None, coupling="full")
    ctl_box[0] = ctl
    # drive level by level so the stub knows which vehicle it plans
    orig_iter_for = ctl._iter_for
    ctl.current_level_members = None

    def plan_level(iters):
        return plan(iters)

    ctl.plan_level = plan_level
    real_step = PrioritizedSequentialController.step

    def tracked_iter_for(i, directed, directed_seq, device_handoff=False):
        ctl.current_level_members = [i]  # full coupling + constant priorities: one vehicle per level
        return orig_iter_for(i, directed, directed_seq, device_handoff)

    ctl._iter_for = tracked_iter_for
    for _ in range(3):
        ctl.step()
    prev = [ctl.info_old[i] for i in range(4)]
    assert all(mpa.trims[int(p.predicted_trims[0]) - 1].speed > 0 for p in prev)  # everybody is moving
    force["on"] = True
    infos = ctl.step()
    # vehicle 2 fell back while planning; 1 is reached over the edge 1-2, 3 and 4 over their edges to 1
    assert infos[1].needs_fallback and not any(infos[i].needs_fallback for i in (0, 2, 3))
    for i in range(4):
        assert np.array_equal(infos[i].y_predicted[:, :-1], prev[i].y_predicted[:, 1:]), i
        assert np.array_equal(infos[i].y_predicted[:, -1], prev[i].y_predicted[:, -1]), i
        assert np.array_equal(infos[i].predicted_trims[:-1], prev[i].predicted_trims[1:]), i


def test_exploration_permutations_follow_the_reference_stream():
    """PrioritizedExplorativeController.computation_level_permutations (:241-309) with RandStream("mt19937ar", Seed = k) / randi:
    the Python producer, its native twin (pdmpc_exploration_permutations) and a third statement on the oracle's own mt19937ar
    agree; the first n_levels rows form a Latin square whose first row is the identity."""
    from oracle import oracle
    from pdmpc.explorative import MatlabRandStream, computation_level_permutations, native_computation_level_permutations

    # the stream itself: numpy's RandomState against the oracle's restatement of mt19937ar / genrand_res53
    for seed in (1, 7, 41):
        want = oracle.mt19937_doubles(seed, 50)
        rs = MatlabRandStream(seed)
        assert [rs.rand() for _ in range(50)] == list(want)
    # known answer of the published generator: rng(0) -> rand = 0.8147..., randi(10) = 9 (seed 0 = default seed 5489)
    assert abs(MatlabRandStream(0).rand() - 0.8147236863931789) < 1e-16 and MatlabRandStream(0).randi(10) == 9
    for n_levels, n_perm, seed in ((5, 5, 3), (13, 64, 21), (12, 12, 8), (1, 4, 2), (14, 64, 40)):
        a = computation_level_permutations(n_levels, n_perm, seed)
        b = native_computation_level_permutations(n_levels, n_perm, seed)
        assert np.array_equal(a, b), (n_levels, n_perm, seed)
        assert list(a[0]) == list(range(1, n_levels + 1))
        sq = a[: min(n_perm, n_levels)]
        for col in range(n_levels):
            assert len(set(sq[:, col])) == len(sq), "a level twice in column %d" % col
        for r in a:
            assert sorted(r) == list(range(1, n_levels + 1))
