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
