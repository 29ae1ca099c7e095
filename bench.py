#!/usr/bin/env python3
"""bench.py — MPC steps/s of the HIP graph-search backend.

Workload at N = 1 (default, BASELINE config C2): 20 vehicles on the CPM-lab road network, horizon 8, InterX constraint
checker, constant priorities, distance coupling; one "step" = every vehicle plans once, all computation levels, with hand-off
of solved areas to successors (one kernel launch per step, dependencies resolved on the device).  Inputs are recorded from the
framework's own closed-loop simulation (the first 20 steps are dropped, as the reference's evaluation does,
eval/eval_phd/eval_phd.m:41-49), packed into HBM before the timed region, and replayed: the planner is deterministic, so a
replayed step does exactly the work of the closed-loop step.

The measurement validates itself (outside the timed region):
  * `parity_checked` / `parity_mismatches`: the CPU oracle plans every recorded step for `cpu_baseline` anyway; its records are
    compared byte for byte (status, ids along the path, pop count, tree size, trajectory, areas) with the records the GPU
    produced for the same bank.  A mismatch makes the run exit non-zero.
  * `bad_status_plans_in_timed_region` (counted on the device: no truncated, overflowed or timed-out search can hide in the timed
    region) and `replay_checked` / `replay_mismatches`: after the timed loop every bank is launched once more and its records are
    compared with the recording's.

Next to `value` (inputs resident in HBM), never as `value`:
  * `value_host_inclusive`: the native closed loop (pdmpc_controller_run): host step logic + pack + H2D + launch + D2H + apply;
  * `value_run_optimizer_literal`: the recorded steps the way an UNMODIFIED reference controller would drive the backend — one
    pdmpc_plan_batch(h, 1, ...) per vehicle in kahn order, hand-over on the host (pdmpc_plan_step_literal).

N > 1 (`--gpus N`, launched by torch.distributed.run) defaults to north_star's scaling workload: C4 (512 vehicles, Hp 10,
colouring levels) in STRONG scaling — coupling-graph components sharded over the ranks, one launch per rank and step, one RCCL
all-gather of the result records (`--shard levels`: every level block-partitioned, one all-gather per level).
`--workload c3` (128 vehicles, 2-level DAG; 1 -> 4 GPUs) and `--workload c5` (64 prioritizations, instances sharded, no
collective on the data path) likewise.  `--weak` is the old mode: N independent C2 networks.  The default N = 1 line carries
`scaling_reference`: the C4 rate on one GPU, the base of the strong-scaling curve.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "p-dmpc_amd")]

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, MI355X_MICROARCH.md
PROFILE_ROUND = "r06"


def workload_defaults(args):
    """BASELINE.json configs as named: C3 = 128 vehicles, Hp 8, colouring priorities cut to a 2-level coupling DAG (max_num_CLs = 2,
    Config.m:28; the cut couplings become previous-trajectory obstacles, PrioritizedController.m:409-447); C4 = 512 vehicles, Hp 10,
    graph-colouring levels (ColoringPrioritizer.m:31-89 + kahn.m); C2 / C5 = constant priorities, no cut."""
    if args.workload == "c3":
        args.vehicles, args.hp = 128, 8
        defaults = ("coloring", 2)
    elif args.workload == "c4":
        args.vehicles, args.hp = 512, 10
        defaults = ("coloring", 99)
    else:
        defaults = ("constant", 99)
    if args.priorities is None:
        args.priorities = defaults[0]
    if args.max_levels is None:
        args.max_levels = defaults[1]
    if args.workload != "c2":
        args.record = min(args.record, 8)
        args.skip = min(args.skip, 4)
    if args.max_nodes <= 0:
        args.max_nodes = (1 << 17) if args.workload == "c2" else (1 << 16)


def build_world(args, seed_offset):
    from pdmpc.config import Config, MpaType, ScenarioType
    from pdmpc.controller import PrioritizedSequentialController
    from pdmpc.mpa import get_mpa
    from pdmpc.road_network import boundary_provider, commonroad_scenario

    tiles = max(1, (args.vehicles + 19) // 20)
    options = Config(
        scenario_type=ScenarioType.commonroad,
        amount=args.vehicles,
        Hp=args.hp,
        mpa_type=MpaType[args.mpa],
        max_vehicles=max(args.vehicles * (args.instances if args.workload == "c5" else 1), 32),
        max_nodes=args.max_nodes,
        max_num_CLs=args.max_levels,
    )
    mpa = get_mpa(options)
    scenario = commonroad_scenario(options, seed=args.seed + seed_offset, tiles=tiles)
    ctl = PrioritizedSequentialController(options, scenario, mpa, None, coupling="distance", boundary_provider=boundary_provider(scenario),
                                         priority_strategy=args.priorities)
    return options, mpa, ctl


def record_steps(options, mpa, ctl, optimizer, n_skip, n_record, explore_instances=0):
    """Closed loop with the GPU planner (untimed); returns the recorded step problems.  With `explore_instances` every
    recorded problem is the flattened batch of that many prioritizations of the step's traffic state (config c5); the
    closed loop itself advances with the controller's own prioritization."""
    problems = []
    last = {}  # vehicle -> nodes its search popped in the step before (what a closed-loop caller knows when it packs the next step)

    def plan_step(prob):
        problems.append(prob)
        # expected work per slot = the work of the same vehicle's search in the PREVIOUS time step (pdmpc_set_step_weights: the
        # dispatch order of the launch; never this step's own pops)
        prob["prev_pops"] = [last.get(v, 0) + 1 for v in prob["order"]]
        infos = optimizer.run_optimizer_step(prob, mpa)
        prob["pops"] = [int(i.n_popped) for i in infos]  # the work each vehicle's search took (weights of the multi-GPU partition)
        for v, p in zip(prob["order"], prob["pops"]):
            last[v] = p
        return infos

    batches = []
    for k in range(n_skip + n_record):
        if explore_instances and k >= n_skip:
            from pdmpc.explorative import build_exploration_batch

            batch = build_exploration_batch(ctl, explore_instances, seed=ctl.k + 1)
            batch["prev_pops"] = [last.get(v, 0) + 1 for v in batch["vehicle"]]
            batches.append(batch)
        ctl.step(plan_step=plan_step)
    return batches if explore_instances else problems[n_skip:]


PARITY_FIELDS = ("status", "n_expanded", "n_popped", "n_hp", "tree_path", "predicted_trims", "shape_cols", "y_predicted", "shapes", "path_nodes")


def count_record_mismatches(a, b):
    """Vehicles whose result records differ in any bit of any field the ABI defines (floating point compared as raw IEEE bits)."""
    bad = np.zeros(len(a), dtype=bool)
    for name in PARITY_FIELDS:
        x, y = np.ascontiguousarray(a[name]), np.ascontiguousarray(b[name])
        if x.dtype.kind == "f":
            x, y = x.view(np.uint64), y.view(np.uint64)
        bad |= (x != y).reshape(len(a), -1).any(axis=1)
    return int(bad.sum())


def cpu_baseline(options, mpa, problems, gpu_records, budget_s):
    """The CPU oracle on the same recorded steps: all vehicles of a level concurrently on min(level, cores) threads of a pool
    that persists across levels and steps (the stand-in for ComputationMode.parallel_threads), whole
    level loop in C++ (oracle_plan_step).  Bounded by `budget_s`.  Its records are the parity check of the measured work."""
    import copy

    from oracle import oracle
    from oracle import packing

    cores = os.cpu_count() or 1
    unbounded = copy.copy(options)
    unbounded.max_nodes = 1 << 30  # the reference's tree is unbounded (Tree.m:54-70)
    mpa_struct, keep = packing.pack_mpa(mpa)
    oracle.plan_step_native(unbounded, mpa, problems[0], n_threads=cores, mpa_struct=mpa_struct)  # (starts the pool: not timed)
    ms_total, thr_total, n_done, mismatches, plans, n_timed, plans_timed, pops_timed = 0.0, 0.0, 0, 0, 0, 0, 0, 0
    t0 = time.time()
    for prob, gpu in zip(problems, gpu_records):
        # every recorded step is planned by the oracle and compared (the parity check of the measured work); the baseline's rate is
        # taken from the steps inside the time budget
        recs, ms, thr = oracle.plan_step_native(unbounded, mpa, prob, n_threads=cores, mpa_struct=mpa_struct)
        if n_timed == 0 or time.time() - t0 <= budget_s:
            ms_total += ms
            thr_total += thr * ms
            n_timed += 1
            plans_timed += len(recs)
            pops_timed += int(recs["n_popped"].sum())
        n_done += 1
        plans += len(recs)
        mismatches += count_record_mismatches(gpu, recs)
    del keep
    return {
        "value": n_timed / (ms_total / 1e3) if ms_total > 0 else None,
        "unit": "MPC steps/s",
        "cores": cores,
        "threads_used_mean": thr_total / ms_total if ms_total > 0 else None,  # time-weighted: a level of one vehicle uses one thread
        "kind": "port",
        "sample": "%d recorded steps of the same workload (%d plans), C++ oracle, whole level loop in C++ (kahn order, hand-over of solved areas on the host), "
        "min(level size, %d) threads of a persistent pool per level" % (n_timed, plans_timed, cores),
        "ms_per_step": ms_total / max(n_timed, 1),
        # what a pop of the reference costs a host thread: busy thread-time / pops (the level chain starves the pool: threads_used_mean)
        "us_per_pop_per_thread": 1e3 * thr_total / pops_timed if pops_timed else None,
    }, n_done, plans, mismatches


def measure_replay(h, problems, steps, warmup, one_step, dist, torch, reset_extra=None, host_gate=None):
    for i in range(warmup):
        one_step(i)
    h.reset_stats()
    if reset_extra is not None:
        reset_extra()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    lat = []
    t_begin = time.perf_counter()
    for i in range(steps):
        t0 = time.perf_counter()
        one_step(i)
        lat.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_begin
    if host_gate is not None:
        # (the group path: rank 0 plans on every device; the other ranks wait HERE, on the host, until it is through — a collective
        # entered early would sit on their GPUs as a spinning RCCL kernel next to rank 0's launches for the whole timed region)
        host_gate()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        elapsed = float(t.item())
    return elapsed, lat


def pack_banks(h, problems, first_bank=0, use_weights=True):
    """Keeps every recorded step resident in HBM (one bank each); returns per-bank records, algorithmic bytes, pops, nodes."""
    recs_per_bank, bytes_per_bank, pops_per_bank, nodes_per_bank = [], [], [], []
    t_grow = 0.0
    for b, prob in enumerate(problems):
        h.select_bank(first_bank + b)
        fb = [f if f is not None else [] for f in prob["fallback"]]
        # (the work of the step before as expected work: the launch fills its slots by priority, pdmpc_set_step_weights; the
        # one-process-per-GPU twin exchanges raw slots between ranks and keeps level order)
        h.pack_step(prob["iters"], prob["preds"], fb, weights=prob.get("prev_pops") if use_weights else None)
        while True:
            h.launch()
            recs = h.fetch(len(prob["iters"]))
            if not (recs["status"] == 2).any():
                break
            # the reference's tree is unbounded (Tree.m:54-70): never measure truncated searches -- double the arenas, plan again
            tg = time.perf_counter()
            h.grow_arena(2 * h.arena_nodes()[0])
            t_grow += time.perf_counter() - tg
        st = h.stats()
        recs_per_bank.append(recs.copy())
        bytes_per_bank.append(st["algorithmic_bytes"])
        pops_per_bank.append(st["nodes_popped"])
        nodes_per_bank.append(st["nodes_generated"])
    return recs_per_bank, bytes_per_bank, pops_per_bank, nodes_per_bank, t_grow


def scaling_reference(args_in, local_rank, torch):
    """The C4 rate on this one GPU (few recorded steps, replayed): the N = 1 point of the strong-scaling curve `--gpus N > 1` reports."""
    import copy

    from pdmpc.optimizer import GraphSearchHip

    a = copy.copy(args_in)
    a.workload, a.priorities, a.max_levels, a.max_nodes = "c4", None, None, 0
    a.record, a.skip = 8, 4  # what `--workload c4` records (workload_defaults): value(N) / this value is a speed-up on the same steps
    workload_defaults(a)
    options, mpa, ctl = build_world(a, 0)
    options.device = local_rank
    opt = GraphSearchHip(options)
    opt._ensure_mpa(mpa)
    h = opt.handle
    h.allow_overflow = True
    problems = record_steps(options, mpa, ctl, opt, a.skip, a.record)
    recs, _, _, _, _ = pack_banks(h, problems)
    S = len(problems)

    def one_step(i):
        h.select_bank(i % S)
        h.launch()
        h.synchronize()

    elapsed, _ = measure_replay(h, problems, 2 * S, S, one_step, None, torch)
    bad = 0
    for b in range(S):
        h.select_bank(b)
        h.launch()
        bad += count_record_mismatches(h.fetch(len(problems[b]["iters"])), recs[b])
    h.close()
    return {"workload": "c4", "n_gpus": 1, "value": 2 * S / elapsed, "unit": "MPC steps/s", "ms_per_step": 1e3 * elapsed / (2 * S),
            "steps": 2 * S, "recorded_steps": S, "replay_mismatches": bad,
            "what": "C4 (512 vehicles, Hp 10, colouring levels) replayed from HBM on one GPU: divide the value of a `--gpus N` line (default workload c4, strong scaling) by this"}


# The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner when a communicator is created):
# file descriptor 1 is pointed at stderr for the whole run and the JSON line goes to a duplicate of the real stdout.
JSON_FD = 1


def main():
    global JSON_FD
    sys.stdout.flush()
    JSON_FD = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--vehicles", type=int, default=20)
    ap.add_argument("--hp", type=int, default=8)
    ap.add_argument("--mpa", default="single_speed", choices=["single_speed", "triple_speed", "realistic"])
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--record", type=int, default=20, help="distinct recorded time steps kept resident in HBM")
    ap.add_argument("--skip", type=int, default=20, help="closed-loop steps dropped before recording")
    ap.add_argument("--max-nodes", type=int, default=0, help="initial per-vehicle arena (0: 1<<17 for c2, 1<<16 otherwise); grows while recording if a search needs more")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skips the oracle (and with it the parity check of the measured steps)")
    ap.add_argument("--no-host-inclusive", action="store_true", help="skip the native closed loop behind `value_host_inclusive` and the literal per-vehicle loop (profiling runs: the timed replay is then the last thing launched)")
    ap.add_argument("--no-scaling-reference", action="store_true", help="N = 1, default workload: skip the C4 single-GPU rate")
    ap.add_argument("--workload", default=None, choices=["c2", "c3", "c4", "c5"], help="default: c2 on one GPU, c4 (strong scaling) on several")
    ap.add_argument("--weak", action="store_true", help="N > 1: one independent C2 network per GPU (weak scaling, no collective) instead of a sharded workload")
    ap.add_argument("--priorities", default=None, choices=["constant", "coloring", "random", "fca"],
                    help="priority strategy of the host driver: vehicle index (ConstantPrioritizer.m), graph colouring "
                    "(ColoringPrioritizer.m), random per step (RandomPrioritizer.m), future collision assessment (FcaPrioritizer.m)")
    ap.add_argument("--max-levels", type=int, default=None,
                    help="options.max_num_CLs (Config.m:28): couplings that do not fit into this many computation levels are cut "
                    "(GreedyCutter.m) and handled as parallel couplings")
    ap.add_argument("--instances", type=int, default=64, help="c5: simultaneous prioritizations per time step")
    ap.add_argument("--multi", default="group", choices=["group", "dist"],
                    help="N > 1, sharded workloads: `group` = the C ABI's own multi-GPU path (pdmpc_group_*: ONE process drives all N devices, RCCL through "
                    "ncclCommInitAll; rank 0 plans, the other ranks of the launcher only take part in the barriers) with a fall-back to `dist` if the group "
                    "cannot be created; `dist` = one process per GPU through torch.distributed (pdmpc.distributed)")
    ap.add_argument("--shard", default="components", choices=["components", "levels"],
                    help="multi-GPU mode of c3/c4: whole coupling-graph components per rank (one speculative launch per rank and step, one "
                    "all-gather of results) or block-partitioned levels (one all-gather per level).  (pdmpc.distributed.plan_step_hybrid -- whole components, a dominating one split by level -- is a library call: the tiled benchmark maps have no dominating component)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    default_workload = args.workload is None
    if args.workload is None:
        args.workload = "c2" if (world == 1 or args.weak) else "c4"
    workload_defaults(args)
    weak = args.workload == "c2"  # every rank its own network
    sharded = not weak
    explore = args.workload == "c5"

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP backend has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("PDMPC_FORCE_DIST") == "1":  # the env switch lets a 1-GPU box exercise the RCCL path
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from pdmpc.optimizer import GraphSearchHip

    options, mpa, ctl = build_world(args, rank if weak else 0)
    options.device = local_rank
    optimizer = GraphSearchHip(options)
    optimizer._ensure_mpa(mpa)
    h = optimizer.handle
    problems = record_steps(options, mpa, ctl, optimizer, args.skip, args.record, args.instances if explore else 0)
    S = len(problems)
    parts = None
    full_problems = problems
    # ---- N > 1 through the C ABI's group (what a MATLAB / C caller has): rank 0 creates one handle per device, packs the recorded steps
    # into group banks (whole coupling-graph components per device by LPT on the recorded pops, or every level sharded) and plans them;
    # the other ranks idle between the barriers.  If the group cannot be set up on this node, every rank falls back to torch.distributed.
    grp = None
    want_group = sharded and args.multi == "group" and (world > 1 or os.environ.get("PDMPC_FORCE_GROUP") == "1")  # (the env switch: a group of ONE device on a 1-GPU box)
    if want_group:
        from pdmpc import backend

        ok = [0]
        if rank == 0:
            # (PDMPC_GROUP_LOGICAL=R on a 1-GPU box: R logical ranks on this GPU — the multi-rank protocol with peer copies for RCCL)
            n_logical = int(os.environ.get("PDMPC_GROUP_LOGICAL", "0"))
            gmode = backend.SHARD_LEVELS if args.shard == "levels" else backend.SHARD_COMPONENTS
            fb0 = [f if f is not None else [] for f in problems[0]["fallback"]]
            want0 = h.plan_step(problems[0]["iters"], problems[0]["preds"], fb0)  # (what the group must return for the first step)
            # RCCL's all-gather first (ncclCommInitAll over the devices of this process); if the group cannot be created with it, fails, or
            # returns other records than the single launch, the same exchange as peer copies ordered by events; then the twin
            for coll in ((backend.COLLECTIVE_COPY,) if n_logical > 1 else (backend.COLLECTIVE_AUTO, backend.COLLECTIVE_COPY)):
                try:
                    grp = (backend.Group(options, n_devices=n_logical, devices=[local_rank] * n_logical, collective=coll) if n_logical > 1
                           else backend.Group(options, n_devices=world, collective=coll))
                    grp.upload_mpa(mpa)
                    grp.grow_arena(h.arena_nodes()[0])
                    for b, prob in enumerate(problems):
                        grp.pack_step(b, prob["iters"], prob["preds"], [f if f is not None else [] for f in prob["fallback"]],
                                      weights=prob.get("prev_pops"), mode=gmode)  # (LPT and dispatch order on the work of the step BEFORE)
                    grp.launch(0)
                    bad0 = count_record_mismatches(grp.fetch(0, len(problems[0]["iters"])), want0)
                    if bad0:
                        raise RuntimeError("%d records of the first step differ from the single launch's" % bad0)
                    ok = [1]
                    break
                except Exception as e:  # noqa: BLE001 (whatever went wrong: the other path is still there)
                    sys.stderr.write("bench.py: pdmpc_group with collective %d not usable (%s)\n" % (coll, e))
                    try:
                        if grp is not None:
                            grp.close()
                    except Exception:  # noqa: BLE001
                        pass
                    grp = None
            if grp is None:
                sys.stderr.write("bench.py: falling back to torch.distributed (one process per GPU)\n")
        # (the library leaves the current device as it found it; torch's collectives are bound to cuda:local_rank, so make sure anyway)
        torch.cuda.set_device(local_rank)
        if dist is not None:
            dist.broadcast_object_list(ok, src=0)
        if not ok[0]:
            args.multi = "dist"
    use_group = want_group and args.multi == "group"
    if not use_group and ((sharded and dist is not None and args.shard == "components" and not explore) or (explore and dist is not None)):
        from pdmpc.distributed import shard_problems

        # every rank recorded the same closed loop; now it keeps only what is assigned to it: whole coupling-graph components by
        # longest processing time on the work the searches took in the closed loop (pops + 1) -- C5: the prioritization instances,
        # which are independent until the final cost comparison (pdmpc.distributed.shard_problems, also driven by the gloo tests)
        parts, problems = shard_problems(full_problems, world, rank, explore=explore)
    # keep every recorded step resident in HBM (one bank each) and collect its algorithmic bytes
    h.allow_overflow = True  # statuses are checked below, per bank
    t_host = time.perf_counter()
    bank_recs, bytes_per_bank, pops_per_bank, nodes_per_bank, t_grow = pack_banks(h, problems, use_weights=dist is None)
    status_counts = {"ok": 0, "exhausted": 0, "arena_overflow": 0, "error": 0}
    for recs in bank_recs:
        status_counts["ok"] += int((recs["status"] == 0).sum())
        status_counts["exhausted"] += int((recs["status"] == 1).sum())
        status_counts["arena_overflow"] += int((recs["status"] == 2).sum())
        status_counts["error"] += int((recs["status"] < 0).sum())
    if use_group and grp is not None:
        # the group's devices get arenas as large as the single handle needed for these steps (the resident path does not plan a step
        # again when a search outgrows its arena: pdmpc_group_plan_step does), and every bank is planned once more in them
        grp.grow_arena(h.arena_nodes()[0])
        for b in range(S):
            grp.launch(b)
    # a bank recorded before the arenas grew replays in the grown arenas: same searches, none of them truncated
    lds_bytes = h.stats()["lds_bytes"]
    host_buffer_ms = 1e3 * (time.perf_counter() - t_host - t_grow) / max(S, 1)  # pack (host buffers -> HBM) + launch + fetch + stats

    planner = None
    gather_bufs = None
    if use_group:
        pass
    elif sharded and dist is not None and args.shard == "levels" and not explore:
        from pdmpc.distributed import HipRangePlanner, plan_step_sharded

        planner = HipRangePlanner(optimizer, mpa, torch.device("cuda", local_rank))
    elif parts is not None:
        from pdmpc.distributed import REC_BYTES

        per = max(max(len(q) for q in pb) for pb in parts)
        gather_bufs = (
            torch.zeros(max(per, 1) * REC_BYTES, dtype=torch.uint8, device="cuda"),
            torch.zeros(max(per, 1) * REC_BYTES * world, dtype=torch.uint8, device="cuda"),
        )
    ext_stream = None
    if gather_bufs is not None:
        ext_stream = torch.cuda.ExternalStream(h.stream_ptr(), device=torch.device("cuda", local_rank))

    def one_step(i):
        if use_group:
            if grp is not None:
                grp.launch(i % S)  # launches, all-gathers and imports on every device's stream, one wait (csrc/group.cpp)
            return
        if planner is not None:
            plan_step_sharded(problems[i % S], planner, dist, rank, world, resident_bank=i % S, fetch=False)
            h.synchronize()
            return
        h.select_bank(i % S)
        h.launch()
        if gather_bufs is not None:
            # end-of-step exchange: every rank receives the records of all components (one RCCL all-gather over xGMI), enqueued on
            # the handle's own stream behind the launch: no host synchronisation between kernel and collective
            h.export_results_async(0, len(problems[i % S]["iters"]), gather_bufs[0].data_ptr())
            with torch.cuda.stream(ext_stream):
                dist.all_gather_into_tensor(gather_bufs[1], gather_bufs[0])
        h.synchronize()

    host_gate = None
    if use_group and dist is not None:
        from torch.distributed.distributed_c10d import _get_default_store

        store, gate_key = _get_default_store(), "pdmpc_bench_group_done"
        host_gate = (lambda: store.set(gate_key, "1")) if rank == 0 else (lambda: store.wait([gate_key]))
    elapsed, lat = measure_replay(h, problems, args.steps, args.warmup, one_step, dist, torch, reset_extra=grp.reset_stats if grp is not None else None, host_gate=host_gate)
    # the group: counts summed over the ranks, kernel time of the rank whose kernels ran longest (backend.Group.stats_all); the
    # roofline below is PER DEVICE (the step's algorithmic bytes / ranks, over that time) against one device's peak
    st = grp.stats_all() if (use_group and grp is not None) else h.stats()
    n_ranks_planning = grp.n_devices if (use_group and grp is not None) else 1
    torch.cuda.set_device(local_rank)
    kernel_ms = st["kernel_ms"]
    n_launch = st["n_launches"]
    alg_bytes = sum(bytes_per_bank[i % S] for i in range(args.steps)) / n_ranks_planning
    pops = sum(pops_per_bank[i % S] for i in range(args.steps))
    nodes = sum(nodes_per_bank[i % S] for i in range(args.steps))
    achieved = (alg_bytes / max(n_launch, 1)) / ((kernel_ms / max(n_launch, 1)) * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    n_search = max(len(p["iters"]) for p in problems)
    # (api.cpp: launch_range -- helper workgroups on the CUs a launch of at most one search per CU leaves idle, 96 for up to two per CU)
    wg_per_launch = n_search + (256 - n_search if n_search <= 254 else 200)  # (the searches and the helper workgroups of the launch: api.cpp, launch_range)

    # ---- the timed launches did the recorded work.  Their records are not fetched (no copies in the timed region), so: the device
    # counts every plan that ended with anything but OK / EXHAUSTED (overflow, time-out) since the reset in front of the timed loop,
    # and every bank is launched once more, in the arenas and under the conditions of the timed loop, and compared with the recording.
    bad_status_timed = st["bad_status_plans"]
    replay_mismatches, replay_checked = 0, 0
    if use_group:
        if grp is not None:
            for b in range(S):
                grp.launch(b)
                after = grp.fetch(b, len(problems[b]["iters"]))
                replay_mismatches += count_record_mismatches(after, bank_recs[b])
                replay_checked += len(after)
    elif planner is None:
        for b in range(S):
            h.select_bank(b)
            h.launch()
            after = h.fetch(len(problems[b]["iters"]))
            replay_mismatches += count_record_mismatches(after, bank_recs[b])
            replay_checked += len(after)

    # ---- the closed loop a caller of the boundary sees: the native step controller (csrc/step_controller.cpp) drives the same
    # scenario, every step = build the step problem on the host + pack (H2D) + one launch + fetch (D2H) + apply, no replay and
    # no interpreter on the path.  Reported next to `value`, never as `value`.
    host_inclusive = None
    literal = None
    if not explore and dist is None and not args.no_host_inclusive:
        from pdmpc.native_controller import NativeController
        from pdmpc.road_network import commonroad_scenario

        tiles = max(1, (args.vehicles + 19) // 20)
        nat = NativeController(options, commonroad_scenario(options, seed=args.seed + (rank if weak else 0), tiles=tiles), mpa, h, coupling="distance",
                               priority_strategy=args.priorities)
        # the SAME closed-loop steps the resident replay cycles through (skip + 1 .. skip + S), as often as the timed loop's step count
        # asks for: a fresh controller per pass (the closed loop cannot be rewound)
        reps = max(1, args.steps // max(S, 1))
        ms_all, sums = [], None
        for rep in range(reps):
            if rep:
                nat.close()
                nat = NativeController(options, commonroad_scenario(options, seed=args.seed + (rank if weak else 0), tiles=tiles), mpa, h, coupling="distance",
                                       priority_strategy=args.priorities)
            nat.run(args.skip)
            nat.timing_mean(reset=True)
            ms_all.append(nat.run(S))
            part = nat.timing_mean()
            sums = part if sums is None else {k: sums[k] + part[k] for k in part}
        ms = np.concatenate(ms_all)
        breakdown = {k: v / reps for k, v in sums.items()}
        host_inclusive = {
            "value": 1e3 / float(np.mean(ms)),
            "unit": "MPC steps/s",
            "ms_per_step": float(np.mean(ms)),
            "p50_latency_ms": float(np.median(ms)),
            "p99_latency_ms": float(np.sort(ms)[min(len(ms) - 1, int(0.99 * len(ms)))]),
            "breakdown_ms": breakdown,
            "what": "closed loop through the C ABI (pdmpc_controller_run): host step logic in C++ + pack + H2D + one launch + D2H + apply per step, closed-loop steps %d..%d (the resident replay's window), %d passes" % (args.skip + 1, args.skip + S, reps),
        }
        nat.close()
        # ---- the same recorded steps, one run_optimizer call per vehicle (what GraphSearchHip.m gives an unmodified controller)
        n_lit = min(S, max(1, args.steps))
        h.select_bank(S)  # a scratch bank: the recorded banks stay as they are
        marshalled = [h.step_args(p["iters"], p["preds"], [f if f is not None else [] for f in p["fallback"]]) for p in problems[:n_lit]]
        h.plan_step_literal(marshalled[0])  # warm-up
        lit_ms, lit_bad = [], 0
        for b in range(n_lit):
            t0 = time.perf_counter()
            recs = h.plan_step_literal(marshalled[b])
            lit_ms.append(1e3 * (time.perf_counter() - t0))
            lit_bad += count_record_mismatches(recs, bank_recs[b])
        literal = {
            "value": 1e3 / float(np.mean(lit_ms)),
            "unit": "MPC steps/s",
            "ms_per_step": float(np.mean(lit_ms)),
            "p50_latency_ms": float(np.median(lit_ms)),
            "calls_per_step": args.vehicles,
            "record_mismatches_vs_single_launch": lit_bad,
            "what": "%d recorded steps, each as %d sequential pdmpc_plan_batch(h, 1, ...) calls in kahn order with the hand-over of solved areas on the host "
            "(pdmpc_plan_step_literal: pack + H2D + launch + D2H per vehicle; marshalling from Python objects excluded)" % (n_lit, args.vehicles),
        }
        replay_mismatches += lit_bad
    if explore and dist is None and not args.no_host_inclusive:
        # config c5 through the C ABI: the native explorative step (pdmpc_controller_explore_step) builds the prioritizations of the
        # step, flattens them, plans them with one launch, chooses per sub-graph and goes on with the chosen plans
        from pdmpc.native_controller import NativeController
        from pdmpc.road_network import commonroad_scenario

        nat = NativeController(options, commonroad_scenario(options, seed=args.seed, tiles=max(1, (args.vehicles + 19) // 20)), mpa, h, coupling="distance",
                               priority_strategy=args.priorities)
        n_x = S  # (exactly the recorded window: the steps the resident replay cycles through)
        # (a) on the SAME steps as the resident replay: every step builds, plans and chooses among the prioritizations, and goes on with the
        # controller's own one — the closed loop the recorded batches come from (pdmpc_controller_explore_follow_own); a fresh controller
        # per pass over the window
        reps = max(1, min(args.steps, 40) // max(S, 1))
        ms_all, sums = [], None
        for rep in range(reps):
            if rep:
                nat.close()
                nat = NativeController(options, commonroad_scenario(options, seed=args.seed, tiles=max(1, (args.vehicles + 19) // 20)), mpa, h, coupling="distance",
                                       priority_strategy=args.priorities)
            nat.run(args.skip)
            nat.explore_follow_own(True)
            nat.timing_mean(reset=True)
            ms_all.append(nat.explore_run(args.instances, n_x))
            part = nat.timing_mean()
            sums = part if sums is None else {k: sums[k] + part[k] for k in part}
        ms = np.concatenate(ms_all)
        breakdown = {k: v / reps for k, v in sums.items()}
        host_inclusive = {
            "value": 1e3 / float(np.mean(ms)),
            "unit": "MPC steps/s",
            "ms_per_step": float(np.mean(ms)),
            "p50_latency_ms": float(np.median(ms)),
            "p99_latency_ms": float(np.sort(ms)[min(len(ms) - 1, int(0.99 * len(ms)))]),
            "breakdown_ms": breakdown,
            "what": "through the C ABI (pdmpc_controller_explore_run) on the steps of the resident replay: per step the %d prioritizations built and flattened in C++ + pack + H2D + "
            "one launch + read-back of status / final cost of every plan + choice per sub-graph + the records of the applied plans; the loop goes on with the controller's own "
            "prioritization, closed-loop steps %d..%d, %d passes" % (args.instances, args.skip + 1, args.skip + n_x, reps),
        }
        # (b) the explorative closed loop proper (the chosen plans are applied: other traffic, heavier steps among them)
        nat.explore_follow_own(False)
        nat.timing_mean(reset=True)
        n_x2 = min(args.steps, 40)
        ms2 = nat.explore_run(args.instances, n_x2)
        host_inclusive["explorative_closed_loop"] = {
            "value": 1e3 / float(np.mean(ms2)),
            "ms_per_step": float(np.mean(ms2)),
            "p50_latency_ms": float(np.median(ms2)),
            "breakdown_ms": nat.timing_mean(),
            "what": "the same call with the chosen plans applied, the %d steps that follow (other traffic; heavier steps among them)" % n_x2,
        }
        nat.close()
    scal_ref = None
    if world == 1 and dist is None and default_workload and not args.no_scaling_reference:
        h.close()
        scal_ref = scaling_reference(args, local_rank, torch)
    if rank == 0:
        traffic, tsrc = None, None
        tpath = os.path.join(ROOT, "profiles", "%s_pmc_traffic_%s.json" % (PROFILE_ROUND, args.workload))
        if os.path.exists(tpath) and world == 1:
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
                tsrc = "profiles/%s: FETCH_SIZE + WRITE_SIZE of the timed launches of this workload, search kernel + the helper kernel next to it (separate rocprofv3 --pmc passes, tools/collect_profiles.sh)" % os.path.basename(tpath)
            except Exception:
                traffic = None
        mode = ("path = group (pdmpc_group_* of the C ABI, ONE process drives %d ranks, exchange = %s): " % (grp.n_devices if grp is not None else world, grp.collective if grp is not None else "-") + ("every level sharded over the devices, one all-gather per level" if args.shard == "levels" else
                "coupling-graph components sharded over the devices, one launch per device and step, one all-gather of results")) if use_group else "path = dist (one process per GPU, torch.distributed): levels sharded over ranks with one all-gather per level" if planner is not None else (
            "path = dist (one process per GPU, torch.distributed): " + ("prioritization instances sharded over ranks" if explore else "coupling-graph components sharded over ranks") + ", one launch per rank and step, one all-gather of results" if gather_bufs is not None else "one launch per step")
        out = {
            "metric": "MPC steps/sec (whole node) + p50 per-step plan latency, N vehicles H=8",
            "value": (1 if sharded else world) * args.steps / elapsed,
            "unit": "MPC steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "p50_latency_ms": 1e3 * statistics.median(lat),
            "p99_latency_ms": 1e3 * sorted(lat)[min(len(lat) - 1, int(0.99 * len(lat)))],
            "max_latency_ms": 1e3 * max(lat),
            "max_latency_step": int(max(range(len(lat)), key=lambda i: lat[i])),
            "step_latencies_ms": [round(1e3 * x, 3) for x in lat] if len(lat) <= 40 else None,
            "host_buffer_ms_per_step": host_buffer_ms,  # PCIe-inclusive path incl. Python marshalling (never `value`)
            # `value` is the contract's number: inputs resident in HBM when the timed region starts.  What a caller of the C ABI gets per
            # step -- host step logic + pack + H2D + launch + D2H + apply (pdmpc_controller_run) -- is value_host_inclusive, on the same
            # closed-loop window; DESIGN.md section 7 quotes both.
            "value_resident": (1 if sharded else world) * args.steps / elapsed,
            "value_host_inclusive": host_inclusive["value"] if host_inclusive else None,
            "host_inclusive": host_inclusive,
            "value_run_optimizer_literal": literal["value"] if literal else None,
            "run_optimizer_literal": literal,
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d vehicles on the CPM-lab road network (labmap fixture%s), Hp %d, InterX checker, %s MPA, "
                "distance coupling, %s priorities, %s; %d recorded closed-loop steps replayed from HBM%s"
                % (args.workload.upper() + (" (%d prioritizations of each step flattened into one batch)" % args.instances if explore else ""),
                   args.vehicles, ", tiled" if sharded and not explore else "", args.hp, args.mpa, args.priorities, mode, S,
                   "" if sharded else "; per GPU one independent network"),
                "multi_path": ("group/" + grp.collective if (use_group and grp is not None) else "dist") if (use_group or planner is not None or gather_bufs is not None) else None,
                "vehicles": args.vehicles,
                "Hp": args.hp,
                "mpa": args.mpa,
                "levels_per_step": statistics.mean(len(p["level_sizes"]) for p in full_problems),
                "max_num_CLs": args.max_levels,
                "seed": args.seed,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": tsrc,
                "kernel": "pdmpc_bulk_kernel",
                "kernel_ms_avg": kernel_ms / max(n_launch, 1),
                "algorithmic_bytes_per_launch": alg_bytes / max(n_launch, 1),
                "launches": n_launch,
                "ranks": n_ranks_planning,  # > 1 (the group): achieved / bytes / kernel time are per device, the slowest device's time
                "lds_bytes_per_workgroup": lds_bytes,
                "open_list": "unordered near (LDS) / mid / far (HBM) lists, bulk-synchronous rounds of the smallest keys; equal keys: replay through the libstdc++-faithful binary heap",
            },
            # The compute side next to the HBM side: the (area segment, obstacle segment) pairs the reference's InterX forms for the edges the
            # kernel evaluated (InterX.m:63-76: ~10 flops per pair) against the vector-f64 peak of the CUs that hold a
            # workgroup of the launch (78.6 TFLOP/s for 256 CUs = half the FP32 vector peak of MI355X_MICROARCH.md).
            "roofline_compute": {
                "bound": "valu_f64",
                "achieved": 10.0 * st["segment_pair_tests"] / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else None,
                "unit": "TFLOP/s",
                "peak_chip": 78.6,
                "workgroups_per_launch": wg_per_launch,
                "peak_of_occupied_cus": 78.6 * min(wg_per_launch, 256) / 256.0,
                "frac_of_occupied_cus": (10.0 * st["segment_pair_tests"] / (kernel_ms * 1e-3) / 1e12) / (78.6 * min(wg_per_launch, 256) / 256.0) if kernel_ms > 0 else None,
                "segment_pair_tests_per_launch": st["segment_pair_tests"] / max(n_launch, 1),
                "flops_per_pair": 10,
            },
            "us_per_pop": {
                "gpu_whole_step": 1e6 * elapsed / pops if pops else None,  # step time / pops of the reference in it (all searches side by side)
                "gpu_kernel": 1e3 * kernel_ms / pops if pops else None,
                "cpu_per_thread": None,  # filled from cpu_baseline below
            },
            # every plan of the recorded steps by outcome; arena_overflow and error must be 0 (the reference's tree is unbounded, Tree.m:54-70)
            "status_counts": status_counts,
            "bad_status_plans_in_timed_region": bad_status_timed,  # device-side count: must be 0
            "replay_checked": replay_checked,  # plans planned once more after the timed loop and compared with the recording's records
            "replay_mismatches": replay_mismatches,
            "safe_replans": st["safe_replans"],
            "arena_nodes_per_vehicle": None if scal_ref is not None else h.arena_nodes()[0],
            "counters": {
                "nodes_popped_per_s": pops / elapsed,
                "nodes_generated_per_s": nodes / elapsed,
                "nodes_popped_per_step": pops / args.steps,
                "edge_checks_per_s": st["edge_checks"] / elapsed,
                "segment_pair_tests_per_s": st["segment_pair_tests"] / elapsed,
                "speculation_arrivals_per_step": st["speculation_arrivals"] / args.steps,
                "nodes_processed_per_step": st["nodes_processed"] / args.steps,  # edges evaluated; nodes_popped of them are the reference's pops
                "rounds_per_step": st["rounds"] / args.steps,
                "shared_rounds_per_step": st["shared_rounds"] / args.steps,  # rounds whose edge checks helper workgroups took part in
                "helper_checked_per_step": st["helper_checked"] / args.steps,
                "queue_fallbacks_per_step": st["queue_fallbacks"] / args.steps,
            },
            "scaling_reference": scal_ref,
        }
        parity_mismatches = 0
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], n_steps_checked, n_plans, parity_mismatches = cpu_baseline(options, mpa, full_problems, bank_recs, args.cpu_budget_s)
            out["us_per_pop"]["cpu_per_thread"] = out["cpu_baseline"].get("us_per_pop_per_thread")
            out["parity_checked"] = n_steps_checked == S
            out["parity_steps_checked"] = n_steps_checked
            out["parity_plans_checked"] = n_plans
            out["parity_mismatches"] = parity_mismatches
        sys.stdout.flush()
        os.write(JSON_FD, (json.dumps(out) + "\n").encode())  # the ONE line of the contract, on the real stdout
        bad = []
        if status_counts["arena_overflow"] or status_counts["error"]:
            bad.append("%d plans overflowed their arena, %d carried an error status" % (status_counts["arena_overflow"], status_counts["error"]))
        if bad_status_timed:
            bad.append("%d plans of the timed region ended with an error or overflow status" % bad_status_timed)
        if replay_mismatches:
            bad.append("%d plans of the timed replay / the literal loop differ from the recording" % replay_mismatches)
        if parity_mismatches:
            bad.append("%d plans differ from the CPU oracle's" % parity_mismatches)
        if scal_ref is not None and scal_ref["replay_mismatches"]:
            bad.append("scaling reference: %d plans differ between recording and replay" % scal_ref["replay_mismatches"])
        if bad:
            raise SystemExit("bench.py: " + "; ".join(bad) + " -- the measurement is void")
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
