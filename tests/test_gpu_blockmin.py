"""The block-min open list (csrc/blockmin_queue.hpp) against the reference's std::priority_queue on command scripts.

With pairwise distinct keys the pop sequence of any exact priority queue equals the reference's; with duplicate keys
the device queue must raise its tie flag whenever the minimum it pops is not unique (the search then falls back to the
libstdc++-faithful heap, tests/test_gpu_heap.py).
"""
import numpy as np
import pytest

from pdmpc.backend import Handle
from pdmpc.config import Config

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import oracle

    return oracle


@pytest.fixture(scope="module")
def handle():
    h = Handle(Config(Hp=5, max_vehicles=2, max_nodes=1024))
    yield h
    h.close()


def search_like_script(seed, n_pops, max_children=12, p_expand=0.45):
    """pop, then with probability p_expand push a batch of children with keys a bit above the popped key."""
    rng = np.random.default_rng(seed)
    ops, keys = [0], [0.0]
    level = 1.0
    for _ in range(n_pops):
        ops.append(1)
        keys.append(0.0)
        if rng.random() < p_expand:
            c = int(rng.integers(1, max_children + 1))
            level += rng.random() * 0.01
            ops += [0] * c
            keys += list(level + rng.random(c) * 3.0)
    return np.array(ops, dtype=np.int32), np.array(keys)


def reference_pops(ops, keys):
    ids = np.zeros(len(ops), dtype=np.int32)
    ids[ops == 0] = np.arange(1, int((ops == 0).sum()) + 1)
    return _oracle().pq_script(ops, ids, keys)


@pytest.mark.parametrize("seed,n_pops,ring", [(0, 3000, 8192), (1, 3000, 64), (2, 40000, 8192), (3, 40000, 1024), (4, 9000, 256)])
def test_distinct_keys_match_std_priority_queue(handle, seed, n_pops, ring):
    ops, keys = search_like_script(seed, n_pops)
    assert len(np.unique(keys[ops == 0])) == int((ops == 0).sum())
    got, tie, cyc_pop, cyc_push = handle.blockmin_script(ops, keys, ring_entries=ring)
    assert not tie
    assert np.array_equal(got, reference_pops(ops, keys))
    print("seed %d: %d pushes, ring %d: cycles per pop %.0f, per pushed node %.0f" % (seed, int((ops == 0).sum()), ring, cyc_pop, cyc_push))


def test_fill_then_drain_is_sorted_and_empty_pops_return_minus_one(handle):
    rng = np.random.default_rng(5)
    n = 20000
    keys = rng.permutation(n).astype(np.float64) * 0.25
    ops = np.concatenate([np.zeros(n, dtype=np.int32), np.ones(n + 3, dtype=np.int32)])
    k2 = np.concatenate([keys, np.zeros(n + 3)])
    got, tie, cyc_pop, _ = handle.blockmin_script(ops, k2, ring_entries=2048)
    print("drain of %d entries, ring 2048: cycles per pop %.0f" % (n, cyc_pop))
    assert not tie
    assert np.array_equal(got[:n], np.argsort(keys) + 1)
    assert got[n:].tolist() == [-1, -1, -1]  # mex.cpp:87-93
    m = 4000
    ops = np.concatenate([np.zeros(m, dtype=np.int32), np.ones(m, dtype=np.int32)])
    got, tie, cyc_pop, _ = handle.blockmin_script(ops, np.concatenate([keys[:m], np.zeros(m)]), ring_entries=8192)
    print("drain of %d entries (one group, all in LDS): cycles per pop %.0f" % (m, cyc_pop))
    assert np.array_equal(got, np.argsort(keys[:m]) + 1)


@pytest.mark.parametrize("where", ["same_block", "other_block", "other_group"])
def test_tied_minimum_raises_the_flag(handle, where):
    n = 9000
    keys = 10.0 + np.arange(n, dtype=np.float64)
    other = {"same_block": 130, "other_block": 700, "other_group": 8500}[where]
    keys[129] = 1.5
    keys[other] = 1.5
    ops = np.concatenate([np.zeros(n, dtype=np.int32), np.ones(1, dtype=np.int32)])
    _, tie, _, _ = handle.blockmin_script(ops, np.concatenate([keys, [0.0]]), ring_entries=8192)
    assert tie
    # a tie that is NOT at the minimum is harmless
    keys[129] = 5000.25
    keys[other] = 5000.25
    got, tie, _, _ = handle.blockmin_script(ops, np.concatenate([keys, [0.0]]), ring_entries=8192)
    assert not tie and got.tolist() == [1]
