"""Simultaneous computation of multiple prioritizations (BASELINE config 4).

Restates the host side of the reference's explorative controller
(hlc/controller/prioritized/PrioritizedExplorativeController.m):

    computation_level_permutations   :241-309  Latin-square-like permutations of the computation levels; the first
                                               row keeps the current prioritization; random draws from the reference's
                                               own stream: RandStream("mt19937ar", Seed = time step) / randi
    one plan per permutation         :25-91    every vehicle plans once per permutation
    solution cost per sub-graph      :94-144   sum over the vehicles of a weakly connected sub-graph of the cost-to-come
                                               of the final node, tree.get_cost(tree_path(end))
    choose_solution                  :146-176  per sub-graph the permutation with the smallest cost after round(., 8)

In the reference the permutations are planned one after the other by every vehicle process.  Here all K instances are
flattened into ONE batch: instance p contributes its 20 vehicles with predecessor lists inside the instance; slots are
ordered by (level, instance) so that the early levels of every instance are dispatched first.  Instances are independent
coupling-graph components, so on several GPUs they shard without any collective (pdmpc.distributed.partition_components).
"""
import numpy as np

from .controller import kahn, directed_coupling_from_priorities
from .distributed import weak_components


class MatlabRandStream:
    """RandStream("mt19937ar", Seed = k) as far as the explorative controller uses it (PrioritizedExplorativeController.m:249,283-286):
    `randi(stream, n)` = floor(n * rand(stream)) + 1 with rand = the 53-bit double of two 32-bit draws (genrand_res53; numpy's
    RandomState is the same generator, pinned against the repo's own mt19937ar in tests/test_host.py).  MATLAB maps Seed = 0 to
    the generator's default seed 5489."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(int(seed) if int(seed) != 0 else 5489)

    def rand(self):
        return float(self.rs.random_sample())

    def randi(self, n):
        return int(np.floor(n * self.rand())) + 1


def computation_level_permutations(n_levels, n_perm, seed):
    """(n_perm, n_levels) array, row 0 = identity.  Rows 1..n_levels-1 follow PrioritizedExplorativeController.m:241-309: no
    column repeats a level (a Latin rectangle built 'fewest possibilities first' with random choices, a row that runs into a dead
    end is drawn again), the random choices from the reference's own stream, RandStream("mt19937ar", Seed = time step) / randi
    (:249, :283-286).  The reference explores exactly n_levels permutations; rows beyond that (n_perm > n_levels, BASELINE config
    C5 asks for 64) are this framework's extension: Fisher-Yates shuffles drawn from the same stream."""
    rng = MatlabRandStream(seed)
    rows = [list(range(1, n_levels + 1))]
    while len(rows) < min(n_perm, n_levels):
        allowed = np.ones((n_levels, n_levels), dtype=bool)  # is_level_allowed(level, class)
        for col in range(n_levels):
            for r in rows:
                allowed[r[col] - 1, col] = False
        perm = [0] * n_levels
        ok = True
        for _ in range(n_levels):
            counts = allowed.sum(axis=0)
            col = int(np.argmin(counts))  # [n_possibilities, i_cell] = min(sum(is_level_allowed, 1)): the first minimum
            if counts[col] == 0:
                ok = False
                break
            choices = np.nonzero(allowed[:, col])[0]  # find(is_level_allowed(:, i_cell)), ascending
            lvl = int(choices[rng.randi(len(choices)) - 1])
            perm[col] = lvl + 1
            allowed[lvl, :] = False
            allowed[:, col] = True  # never the column with the fewest possibilities again
        if ok:
            rows.append(perm)
    while len(rows) < n_perm:
        perm = list(range(1, n_levels + 1))
        for i in range(n_levels - 1, 0, -1):  # Fisher-Yates from the back
            j = rng.randi(i + 1) - 1
            perm[i], perm[j] = perm[j], perm[i]
        rows.append(perm)
    return np.array(rows[:n_perm], dtype=np.int64)


def native_computation_level_permutations(n_levels, n_perm, seed):
    """The same table from libpdmpc_hip.so (pdmpc_exploration_permutations, csrc/step_controller.cpp): the native twin."""
    import ctypes as C

    from . import backend

    L = backend.load_library()
    L.pdmpc_exploration_permutations.argtypes = [C.c_int32, C.c_int32, C.c_uint32, C.POINTER(C.c_int32)]
    out = np.zeros((n_perm, n_levels), dtype=np.int32)
    rc = L.pdmpc_exploration_permutations(n_levels, n_perm, int(seed), out.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise backend.BackendError("pdmpc_exploration_permutations failed: %d" % rc)
    return out.astype(np.int64)


def build_exploration_batch(ctl, n_perm, seed):
    """One flattened step problem holding `n_perm` prioritizations of the controller's current traffic state.
    Returns the problem (slots ordered by (level, instance)) with extra keys `instance` and `vehicle` per slot."""
    base = ctl.build_step_problem()  # refreshes the traffic state and the adjacency
    # the computation levels of the controller's own prioritization: kahn of the sequential coupling the step was just built with
    # (PrioritizedExplorativeController.m prepare_permutation :42-58 permutes kahn(iter.directed_coupling_sequential))
    levels0 = np.asarray(ctl.last_levels)
    n_levels = int(levels0.max())
    perms = computation_level_permutations(n_levels, n_perm, seed)
    parts = []
    directed0, seq0 = np.array(ctl.last_directed) != 0, np.array(base["directed_seq"]) != 0
    for p in range(n_perm):
        # prepare_permutation (:42-77): i11changem(levels, 1:n, permutation) -- a vehicle of (old) level L gets the POSITION of L in
        # the permutation as its new level; every coupling i -> j of the base prioritization that the new levels invert is swapped
        # in all coupling matrices (a sequential coupling stays sequential, a parallel one parallel and as directed)
        where = {int(lvl): j + 1 for j, lvl in enumerate(perms[p])}
        prio = [where[int(levels0[v])] for v in range(ctl.n)]
        if p == 0:
            prob = base
        else:
            directed, seq = directed0.copy(), seq0.copy()
            for i, j in zip(*np.nonzero(directed0)):
                if prio[i] > prio[j]:
                    directed[i, j], directed[j, i] = False, True
                    if seq0[i, j]:
                        seq[i, j], seq[j, i] = False, True
            prob = ctl.build_step_problem(refresh=False, couplings=(directed, seq))
        parts.append(prob)
    flat = []
    for p, prob in enumerate(parts):
        for s in range(len(prob["iters"])):
            flat.append((prob["levels"][s], p, s))
    flat.sort()
    slot_of = {(p, s): i for i, (_, p, s) in enumerate(flat)}
    out = {"order": [], "iters": [], "preds": [], "fallback": [], "levels": [], "instance": [], "vehicle": []}
    for lvl, p, s in flat:
        prob = parts[p]
        out["order"].append(prob["order"][s])
        out["iters"].append(prob["iters"][s])
        out["preds"].append([slot_of[(p, q)] for q in prob["preds"][s]])
        out["fallback"].append(prob["fallback"][s])
        out["levels"].append(lvl)
        out["instance"].append(p)
        out["vehicle"].append(prob["order"][s])
    lv = np.array(out["levels"])
    out["level_sizes"] = [int(np.sum(lv == l)) for l in range(1, int(lv.max()) + 1)]
    out["n_instances"] = n_perm
    out["adjacency"] = np.array(ctl.last_adjacency)
    out["graph_coupling"] = np.array(base["directed_seq"])  # the sub-graphs of the cost choice: conncomp(directed_coupling_sequential) of the base prioritization (:94-112)
    out["directed_seq"] = [prob["directed_seq"] for prob in parts]  # per instance: what a vehicle that goes on with it inherits
    return out


def choose_solution(batch, records, Hp):
    """PrioritizedExplorativeController.m:94-176: per weakly connected sub-graph of the coupling graph, the instance with the
    smallest summed cost-to-come of the final nodes after round(., 8).  Returns {sub-graph label: chosen instance} and the
    cost table (instances x sub-graphs).  A vehicle whose search was exhausted makes its instance infinitely expensive."""
    adj = batch["graph_coupling"]
    n = adj.shape[0]
    labels = weak_components([[j for j in range(n) if adj[i, j] or adj[j, i]] for i in range(n)])
    graphs = sorted(set(labels))
    K = batch["n_instances"]
    cost = np.zeros((K, len(graphs)))
    for slot, (p, v) in enumerate(zip(batch["instance"], batch["vehicle"])):
        rec = records[slot]
        c = float(rec["path_nodes"][Hp][4]) if int(rec["status"]) == 0 else np.inf
        cost[p, graphs.index(labels[v])] += c
    cost = np.round(cost, 8)
    return {g: int(np.argmin(cost[:, gi])) for gi, g in enumerate(graphs)}, cost


def explore_step(ctl, plan_batch, n_perm):
    """One explorative time step of the Python controller (PrioritizedExplorativeController.m:25-176; twin of
    pdmpc_controller_explore_step): the step's prioritizations as one batch (seed = time step, :249), `plan_batch(batch)` -> records
    in slot order, the choice per sub-graph, and every vehicle goes on with the plan and the couplings of its sub-graph's choice
    (obj.info / obj.iter = ..._array_tmp{chosen_solution}, :157-158).  Returns (batch, records, chosen instance per vehicle)."""
    from .iteration_data import info_from_record

    Hp = ctl.options.Hp
    kept = {}

    def plan_step(prob):
        batch = build_exploration_batch(ctl, n_perm, seed=ctl.k)
        records = plan_batch(batch)
        chosen_of_graph, _ = choose_solution(batch, records, Hp)
        adj = batch["graph_coupling"]
        labels = weak_components([[j for j in range(ctl.n) if adj[i, j] or adj[j, i]] for i in range(ctl.n)])
        chosen = [chosen_of_graph[labels[v]] for v in range(ctl.n)]
        slot = {(p, v): s for s, (p, v) in enumerate(zip(batch["instance"], batch["vehicle"]))}
        seq = np.zeros_like(batch["directed_seq"][0])
        for v in range(ctl.n):
            seq[v, :] = batch["directed_seq"][chosen[v]][v, :]
        ctl.last_directed_seq = seq
        kept.update(batch=batch, records=records, chosen=chosen)
        return [info_from_record(records[slot[(chosen[v], v)]], Hp) for v in prob["order"]]

    ctl.step(plan_step=plan_step)
    return kept["batch"], kept["records"], kept["chosen"]
