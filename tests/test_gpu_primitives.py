"""The collision primitives on the device, directly: the reference's own known-answer vectors
(tests/unittests/hlc/intersect_unittest.m:8-54) and 10^4 random polygon / polyline pairs per primitive against the oracle —
touching, collinear, NaN-separated and zero-length edges included.  Goes through pdmpc_debug_edge_check, which runs the
device functions the search kernels inline (csrc/edge_checks.hpp), one wavefront per case."""
import json
import os

import numpy as np
import pytest

from pdmpc.backend import Handle

import problems

pytestmark = pytest.mark.gpu

KA = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_known_answers.json")))


@pytest.fixture(scope="module")
def handle():
    options = problems.make_options("interx", Hp=6)
    options.max_vehicles = 4
    options.max_nodes = 1 << 12
    h = Handle(options)
    yield h
    h.close()


def test_reference_intersect_sat_vectors_on_the_device(handle):
    a, b, want = [], [], []
    for case in KA["intersect_sat"]:
        s1 = np.array(case["shape1"], dtype=np.float64)
        s2 = s1 + np.array(case["shift"], dtype=np.float64).reshape(2, 1)
        a += [s1, s2]
        b += [s2, s1]
        want += [case["expected"]] * 2
    assert handle.edge_check(1, a, b).tolist() == want


def test_reference_intersect_lanelets_vectors_on_the_device(handle):
    """intersect_lanelets.m:1-22 = intersect_sat against every right / left boundary segment (2-point polygons)."""
    rows = np.array(KA["lanelet_1_rows_rx_ry_lx_ly_cx_cy"], dtype=np.float64)
    for case in KA["intersect_lanelets"]:
        shape = np.array(case["shape"], dtype=np.float64)
        segs = []
        for i in range(rows.shape[0] - 1):
            segs.append(np.array([[rows[i, 0], rows[i + 1, 0]], [rows[i, 1], rows[i + 1, 1]]]))
            segs.append(np.array([[rows[i, 2], rows[i + 1, 2]], [rows[i, 3], rows[i + 1, 3]]]))
        got = handle.edge_check(1, [shape] * len(segs), segs)
        assert bool(got.any()) == case["expected"]


def _random_shape(rng, convex):
    """A vehicle-area-like closed polygon with 5-7 columns (generate_maneuver.m:74-101) somewhere near the origin."""
    n = int(rng.integers(4, 7))
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    rad = rng.uniform(0.1, 0.4, n) if not convex else np.full(n, rng.uniform(0.1, 0.4))
    c = rng.uniform(-0.5, 0.5, 2)
    p = np.stack([c[0] + rad * np.cos(ang), c[1] + rad * np.sin(ang)])
    return np.concatenate([p, p[:, :1]], axis=1)  # closed: the last column repeats the first


def _random_soup(rng, special):
    """A NaN-separated obstacle soup as vectorize_all_obstacles.m builds it; `special` adds degenerate geometry."""
    parts = []
    for _ in range(int(rng.integers(1, 5))):
        poly = _random_shape(rng, convex=False) + rng.uniform(-0.6, 0.6, (2, 1))
        if special == 1:  # a zero-length edge in the middle
            poly = np.insert(poly, 2, poly[:, 2], axis=1)
        parts += [poly, np.full((2, 1), np.nan)]
    return np.concatenate(parts, axis=1)


@pytest.mark.parametrize("special", [0, 1, 2, 3])
def test_interx_random_pairs(handle, special):
    from oracle import oracle

    rng = np.random.default_rng(100 + special)
    a, b = [], []
    for _ in range(2500):
        s = _random_shape(rng, convex=False)
        o = _random_soup(rng, special)
        if special == 2:  # touching: an obstacle vertex exactly on a shape vertex, a shared collinear edge (strict < 0 must not fire)
            o[:, 0] = s[:, 1]
            o[:, 1] = s[:, 2]
        if special == 3:  # lattice coordinates: products that are exactly zero
            s = np.round(s * 4) / 4
            o = np.round(o * 4) / 4
        a.append(s)
        b.append(o)
    got = handle.edge_check(0, a, b)
    want = np.array([oracle.interx(x, y) for x, y in zip(a, b)])
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
    assert 0.05 < want.mean() < 0.95  # both outcomes occur


@pytest.mark.parametrize("special", [0, 2, 3, 4])
def test_interx_large_soups(handle, special):
    """Soups long enough (more than PDMPC_CULL_MIN segments) for the bounding-box cull in front of InterX pass 1: the cull is
    exact, so the results equal the oracle's bit for bit, including obstacles that are collinear with a shape edge far away
    (where rounding noise alone decides InterX.m:63-76)."""
    from oracle import oracle

    rng = np.random.default_rng(400 + special)
    a, b = [], []
    for _ in range(300):
        s = _random_shape(rng, convex=False)
        parts = []
        n_poly = int(rng.integers(60, 140))
        spread = rng.uniform(3.0, 12.0)
        for q in range(n_poly):
            poly = _random_shape(rng, convex=False) + rng.uniform(-spread, spread, (2, 1))
            if special == 4 and q % 3 == 0:  # a copy of the shape pushed along one of its own edges: exactly collinear in real arithmetic
                e = int(rng.integers(0, s.shape[1] - 1))
                d = s[:, e + 1] - s[:, e]
                poly = s + (d * rng.uniform(6.0, 30.0))[:, None]
            parts += [poly, np.full((2, 1), np.nan)]
        o = np.concatenate(parts, axis=1)[:, :1000]
        if special == 2:
            o[:, 0] = s[:, 1]
            o[:, 1] = s[:, 2]
        if special == 3:
            s = np.round(s * 4) / 4
            o = np.round(o * 4) / 4
        assert o.shape[1] > 300
        a.append(s)
        b.append(o)
    got = handle.edge_check(0, a, b)
    want = np.array([oracle.interx(x, y) for x, y in zip(a, b)])
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
    assert 0.02 < want.mean() < 0.98


@pytest.mark.parametrize("special", [0, 3])
def test_intersect_sat_random_pairs(handle, special):
    from oracle import oracle

    rng = np.random.default_rng(200 + special)
    a, b = [], []
    for _ in range(5000):
        s = _random_shape(rng, convex=True)
        o = _random_shape(rng, convex=True) + rng.uniform(-0.5, 0.5, (2, 1))
        if special == 3:
            s = np.round(s * 4) / 4
            o = np.round(o * 4) / 4
        a.append(s)
        b.append(o)
    got = handle.edge_check(1, a, b)
    want = np.array([oracle.intersect_sat(x, y) for x, y in zip(a, b)])
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
    assert 0.05 < want.mean() < 0.95


def test_lanelet_boundary_random(handle):
    from oracle import oracle

    rng = np.random.default_rng(300)
    a, b, want = [], [], []
    for _ in range(4000):
        left, right, _ = problems.corridor(rng, x0=-1.0, x1=1.5, half_width=rng.uniform(0.1, 0.5), n=int(rng.integers(4, 30)), wobble=rng.uniform(0, 0.2))
        s = _random_shape(rng, convex=True)
        nan = np.full((2, 1), np.nan)
        a.append(s)
        b.append(np.concatenate([left, nan, right, nan], axis=1))
        want.append(oracle.intersect_lanelet_boundary(s, left, right))
    got = handle.edge_check(2, a, b)
    assert np.array_equal(got, np.array(want)), np.flatnonzero(got != np.array(want))[:10]
    assert 0.05 < np.mean(want) < 0.95
