"""Weighing and cutting of the directed coupling graph (PrioritizedController.group,
PrioritizedController.m:375-389).

The reference bounds the number of computation levels of a time step (`options.max_num_CLs`, Config.m:28): couplings
are weighed, and only as many of them as fit into `max_num_CLs` levels stay *sequential* (the successor waits for the
predecessor's plan of this step); the rest become *parallel* couplings (the successor uses the predecessor's plan of
the previous step, shifted by one, PrioritizedController.m:409-447).  Fewer, wider levels are exactly what fills a
GPU: a level is one batch of independent vehicle workgroups.

    ConstantWeigher   weight/ConstantWeigher.m:15-17
    DistanceWeigher   weight/DistanceWeigher.m:12-39
    RandomWeigher     weight/RandomWeigher.m:13-21 (mt19937ar seeded with the time step, one draw per edge)
    GreedyCutter      cut/GreedyCutter.m:5-86
"""
import math

import numpy as np


def _edges_column_major(M):
    """[row, col] = find(M): non-zero entries in MATLAB's column-major order (0-based pairs)."""
    M = np.asarray(M)
    cols, rows = np.nonzero(M.T)
    return list(zip(rows.tolist(), cols.tolist()))


def constant_weight(directed_coupling):
    return np.asarray(directed_coupling, dtype=np.float64) * 0.5


def distance_weight(directed_coupling, x0, max_mpa_speed, dt_seconds, Hp):
    """1 - distance / (2 v_max dt Hp) per coupled pair: the closer, the heavier."""
    W = np.asarray(directed_coupling, dtype=np.float64).copy()
    max_distance = 2 * max_mpa_speed * dt_seconds * Hp
    for a, b in _edges_column_major(directed_coupling):
        dx = x0[a][0] - x0[b][0]
        dy = x0[a][1] - x0[b][1]
        W[a, b] = 1 - math.sqrt(dx * dx + dy * dy) / max_distance  # norm() of a 2-vector
    return W


def mt19937ar_doubles(seed, n):
    """rand(RandStream("mt19937ar", Seed = seed), 1, n): init_genrand(seed) + genrand_res53, which is exactly numpy's
    legacy RandomState (tests/test_oracle_golden.py pins the oracle's generator to the same stream)."""
    return np.random.RandomState(int(seed)).random_sample(int(n))


def random_weight(directed_coupling, time_step):
    W = np.asarray(directed_coupling, dtype=np.float64).copy()
    edges = _edges_column_major(directed_coupling)
    r = mt19937ar_doubles(time_step, len(edges))
    for (a, b), w in zip(edges, r):
        W[a, b] = w
    return W


def kahn_levels(A):
    from .controller import kahn

    return kahn(A)


def greedy_cut(weighted_coupling, max_num_CLs):
    """GreedyCutter.cut -> sequential directed coupling (bool [n x n]).  Edges are visited by descending weight
    (stable: equal weights keep column-major order, MATLAB's sort is stable) and made sequential when that does not
    push the level count above max_num_CLs."""
    M = np.asarray(weighted_coupling, dtype=np.float64)
    n = M.shape[0]
    seq = np.zeros((n, n), dtype=bool)
    if max_num_CLs == 1:  # :8-11
        return seq
    edges = _edges_column_major(M)
    order = sorted(range(len(edges)), key=lambda e: -M[edges[e]])  # sorted() is stable
    levels = kahn_levels(seq)
    for e in order:
        a, b = edges[e]
        if levels[a] < levels[b]:  # :64-69 the edge already points down the level order
            seq[a, b] = True
            continue
        trial = seq.copy()  # :71-82 would moving b below a keep the level count within the bound?
        trial[a, b] = True
        new_levels = kahn_levels(trial)
        if new_levels.max() <= max_num_CLs:
            seq = trial
            levels = new_levels
    return seq
