"""Priority strategies that turn the undirected coupling graph of a time step into the directed coupling the level
loop needs.

    constant priorities    ConstantPrioritizer.m:14-20 + Prioritizer.directed_coupling_from_priorities (Prioritizer.m:64-77):
                           pdmpc.controller.directed_coupling_from_priorities
    graph colouring        ColoringPrioritizer.m:11-27: colour the undirected graph (saturation-degree order with
                           largest-degree tie break, :65-89), one computation level per colour (:31-63), levels ordered so
                           that the level of the vertex with the most incoming edges comes first (:91-131), edges
                           directed from the earlier level to the later one (Prioritizer.direct_coupling, Prioritizer.m:36-62)

Few colours = few computation levels = short dependency chains: the number of levels of a colouring is bounded by the
maximum degree + 1, whereas constant priorities can chain all vehicles of a connected component.
"""
import numpy as np


def _vertex_sdo_ldo(A, color, degree):
    """ColoringPrioritizer.m:65-89 (0-based): next vertex = most distinct neighbour colours; among equals the LAST one
    visited whose degree is strictly larger than the current pick's (the reference's two independent ifs)."""
    best, idx = -1, -1
    for i in np.flatnonzero(color == 0):
        d = len(set(int(c) for c in color[A[i] == 1] if c != 0))
        if d > best:
            best, idx = d, i
        if d == best and degree[i] > degree[idx]:
            idx = i
    return idx


def topological_coloring(adjacency):
    """ColoringPrioritizer.m:31-63 -> (colors 1-based per vertex, level matrix L [n_colors x n])."""
    A = (np.asarray(adjacency) != 0).astype(np.int64)
    np.fill_diagonal(A, 0)
    n = A.shape[0]
    degree = A.sum(axis=0)
    color = np.zeros(n, dtype=np.int64)
    color[degree == 0] = 1  # :45
    while not np.all(color != 0):
        v = _vertex_sdo_ldo(A, color, degree)
        neighbour = set(int(c) for c in color[A[v] == 1])
        color[v] = next(c for c in range(1, n + 1) if c not in neighbour)  # setdiff(col, neighbor_col)(1)
    used = np.unique(color)
    L = np.zeros((len(used), n), dtype=np.int64)
    for i, c in enumerate(used):
        L[i, color == c] = 1
    assert int(L.sum()) == n
    return color, L


def order_topo(L, adjacency):
    """ColoringPrioritizer.m:91-131: order of the levels (0-based rows of L).  Levels whose vertices have no edges at all
    never enter the reference's order vector; they are appended here in their natural order (they couple with nobody,
    so where they plan does not matter)."""
    C = (np.asarray(adjacency) != 0).astype(np.int64)
    deg = C.sum(axis=0).astype(np.int64)
    members = [np.flatnonzero(L[g]) for g in range(L.shape[0])]
    order = []
    if deg.sum() == 0:
        return list(range(L.shape[0]))
    while deg.sum() != 0:
        max_idx = int(np.argmax(deg))  # first index of the maximum (strict > in the reference's scan)
        lvl = next(j for j, m in enumerate(members) if max_idx in m)
        order.append(lvl)
        deg[members[lvl]] = 0
    order += [g for g in range(L.shape[0]) if g not in order]
    return order


def coloring_directed_coupling(adjacency):
    """ColoringPrioritizer.prioritize (:11-27) -> (directed coupling bool [n x n], level of each vertex, 1-based)."""
    A = (np.asarray(adjacency) != 0)
    _, L = topological_coloring(A)
    L = L[order_topo(L, A)]
    level = np.argmax(L, axis=0) + 1
    d = A.copy()
    np.fill_diagonal(d, False)
    d[level[:, None] > level[None, :]] = False  # Prioritizer.m:52-55: an edge from a later level to an earlier one goes
    return d, level


def random_priorities(n, time_step):
    """RandomPrioritizer.m:15-25: a permutation drawn from mt19937ar seeded with the time step.  MATLAB's `randperm`
    algorithm is not part of the reference (parity unpinned); this draws a Fisher-Yates shuffle from the
    same 53-bit stream, so runs are reproducible per time step as in the reference."""
    from .grouping import mt19937ar_doubles

    r = mt19937ar_doubles(time_step, n)
    p = list(range(1, n + 1))
    for i in range(n - 1, 0, -1):
        j = int(r[n - 1 - i] * (i + 1))
        p[i], p[j] = p[j], p[i]
    return p


def intersect_sat(shape1, shape2):
    """intersect_sat.m:1-42 for the host-side collision assessment below (2 x V arrays, open polygons)."""

    def separated(a, b):
        closed = np.concatenate([a, a[:, :1]], axis=1)
        edges = np.diff(closed, axis=1)
        normals = np.stack([-edges[1], edges[0]])
        normals = normals / np.sqrt(normals[0] ** 2 + normals[1] ** 2)
        pa = normals.T @ a
        pb = normals.T @ b
        return bool(np.any((pa.min(axis=1) - pb.max(axis=1) > 0) | (pb.min(axis=1) - pa.max(axis=1) > 0)))

    s1 = np.asarray(shape1, dtype=np.float64)
    s2 = np.asarray(shape2, dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):  # a zero-length edge gives a NaN axis; NaN > 0 is false
        return not (separated(s1, s2) or separated(s2, s1))


def calculate_yaw(path):
    """utility/calculate_yaw.m: heading of every point of a (n, 2) path (central differences, one-sided at the ends)."""
    path = np.asarray(path, dtype=np.float64)
    yaw = np.zeros(len(path))
    d = path[2:] - path[:-2]
    yaw[1:-1] = np.arctan2(d[:, 1], d[:, 0])
    yaw[0] = np.arctan2(path[1, 1] - path[0, 1], path[1, 0] - path[0, 0])
    yaw[-1] = np.arctan2(path[-1, 1] - path[-2, 1], path[-1, 0] - path[-2, 0])
    return yaw


def fca_priorities(adjacency, reference_points, length, width, offset, obstacles=(), dynamic_obstacle_area=()):
    """FcaPrioritizer.m:13-92 (future collision assessment): count, per vehicle, the steps at which its footprint on the
    reference trajectory overlaps an obstacle or the footprint of a coupled vehicle; more collisions = earlier.
    reference_points: per vehicle an (Hp, 2) array.  Returns the reference's `current_priorities` vector — the
    *positions* of the descending sort, which the reference feeds to directed_coupling_from_priorities as they are."""
    A = np.asarray(adjacency) != 0
    n = A.shape[0]
    Hp = len(reference_points[0])
    collisions = np.zeros(n)
    xl = np.array([-1, -1, 1, 1]) * (length / 2 + offset)
    yl = np.array([-1, 1, 1, -1]) * (width / 2 + offset)
    yaws = [calculate_yaw(reference_points[v]) for v in range(n)]

    def footprint(v, s):
        c, si = np.cos(yaws[v][s]), np.sin(yaws[v][s])
        x0, y0 = reference_points[v][s]
        return np.stack([c * xl - si * yl + x0, si * xl + c * yl + y0])

    for a in range(n - 1):
        later = [b for b in np.flatnonzero(A[a]) if b > a]
        for s in range(Hp):
            shape_a = footprint(a, s)
            for o in obstacles:
                if intersect_sat(shape_a, o):
                    collisions[a] += 1
            for row in dynamic_obstacle_area:
                if intersect_sat(shape_a, row[s]):
                    collisions[a] += 1
            for b in later:
                if intersect_sat(shape_a, footprint(b, s)):
                    collisions[a] += 1
                    collisions[b] += 1
    order = sorted(range(n), key=lambda v: -collisions[v])  # sort(..., 'descend') is stable
    return [v + 1 for v in order], collisions
