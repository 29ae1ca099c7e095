#!/usr/bin/env python3
"""bench.py — MPC steps/s of the HIP graph-search backend on BASELINE config 1 (C2).

Workload (N = 1): 20 vehicles on the CPM-lab road network, horizon 8, InterX constraint checker,
constant priorities, distance coupling; one "step" = every vehicle plans once, all computation levels,
with hand-off of solved areas to successors (one kernel launch per step, dependencies resolved on
the device).  Inputs are recorded from the framework's own closed-loop simulation (the first 20 steps are
dropped, as the reference's evaluation does, eval/eval_phd/eval_phd.m:41-49), packed into HBM before the
timed region, and replayed: the planner is deterministic, so a replayed step does exactly the work of the
closed-loop step.

N > 1 (`--gpus N`, launched by torch.distributed.run): weak scaling — every rank plans its own independent
20-vehicle road network (vehicles of different networks are not coupled, so the data path has no collective);
value = network-steps per second summed over ranks.

`--workload c3` (128 vehicles, Hp 8) and `--workload c4` (512 vehicles, Hp 10) are BASELINE configs 2 and 3 on a
tiled map: at N = 1 one launch per step, at N > 1 every computation level is block-partitioned over the ranks and the
solved areas are exchanged with one RCCL all-gather per level (strong scaling; see pdmpc/distributed.py).

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


def build_world(args, rank):
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
        max_num_CLs=getattr(args, "max_levels", 99),
    )
    mpa = get_mpa(options)
    scenario = commonroad_scenario(options, seed=args.seed + rank, tiles=tiles)
    ctl = PrioritizedSequentialController(options, scenario, mpa, None, coupling="distance", boundary_provider=boundary_provider(scenario),
                                         priority_strategy=getattr(args, "priorities", "constant"))
    return options, mpa, ctl


def record_steps(options, mpa, ctl, optimizer, n_skip, n_record, explore_instances=0):
    """Closed loop with the GPU planner (untimed); returns the recorded step problems.  With `explore_instances` every
    recorded problem is the flattened batch of that many prioritizations of the step's traffic state (config c5); the
    closed loop itself advances with the controller's own prioritization."""
    problems = []

    def plan_step(prob):
        problems.append(prob)
        infos = optimizer.run_optimizer_step(prob, mpa)
        prob["pops"] = [int(i.n_popped) for i in infos]  # the work each vehicle's search took (weights of the multi-GPU partition)
        return infos

    batches = []
    for k in range(n_skip + n_record):
        if explore_instances and k >= n_skip:
            from pdmpc.explorative import build_exploration_batch

            batches.append(build_exploration_batch(ctl, explore_instances, seed=ctl.k + 1))
        ctl.step(plan_step=plan_step)
    return batches if explore_instances else problems[n_skip:]


def cpu_baseline(options, mpa, problems, budget_s):
    """The CPU oracle on the same recorded steps: all vehicles of a level concurrently on min(level, cores) threads
    (the stand-in for ComputationMode.parallel_threads).  Bounded by `budget_s`."""
    from oracle import oracle
    from pdmpc import abi

    cores = os.cpu_count() or 1
    mpa_struct, keep = abi.pack_mpa(mpa)
    ms_total, n_done = 0.0, 0
    t0 = time.time()
    for prob in problems:
        _, ms = oracle.plan_step(options, mpa, prob, n_threads=cores, mpa_struct=mpa_struct)
        ms_total += ms
        n_done += 1
        if time.time() - t0 > budget_s:
            break
    del keep
    return {
        "value": n_done / (ms_total / 1e3) if ms_total > 0 else None,
        "unit": "MPC steps/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d recorded steps of the same workload, C++ oracle, levels in kahn order, min(level size, %d) threads per level, planning time only" % (n_done, cores),
        "ms_per_step": ms_total / max(n_done, 1),
    }


def main():
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
    ap.add_argument("--cpu-budget-s", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-inclusive", action="store_true", help="skip the native closed loop behind `value_host_inclusive` (profiling runs: the timed replay is then the last thing launched)")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--priorities", default=None, choices=["constant", "coloring", "random", "fca"],
                    help="priority strategy of the host driver: vehicle index (ConstantPrioritizer.m), graph colouring "
                    "(ColoringPrioritizer.m), random per step (RandomPrioritizer.m), future collision assessment (FcaPrioritizer.m)")
    ap.add_argument("--max-levels", type=int, default=None,
                    help="options.max_num_CLs (Config.m:28): couplings that do not fit into this many computation levels are cut "
                    "(GreedyCutter.m) and handled as parallel couplings")
    ap.add_argument("--instances", type=int, default=64, help="c5: simultaneous prioritizations per time step")
    ap.add_argument("--shard", default="components", choices=["components", "levels"],
                    help="multi-GPU mode of c3/c4: whole coupling-graph components per rank (one speculative launch per rank and step, one "
                    "all-gather of results) or block-partitioned levels (one all-gather per level)")
    args = ap.parse_args()
    # BASELINE.json configs as named: C3 = 128 vehicles, Hp 8, colouring priorities cut to a 2-level coupling DAG
    # (max_num_CLs = 2, Config.m:28; the cut couplings become previous-trajectory obstacles, PrioritizedController.m:409-447);
    # C4 = 512 vehicles, Hp 10, graph-colouring levels (ColoringPrioritizer.m:31-89 + kahn.m); C2 / C5 = constant priorities, no cut
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
    sharded = args.workload != "c2"
    if sharded:
        args.record = min(args.record, 8)
        args.skip = min(args.skip, 4)
    if args.max_nodes <= 0:
        args.max_nodes = (1 << 17) if args.workload == "c2" else (1 << 16)
    explore = args.workload == "c5"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP backend has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("PDMPC_FORCE_DIST") == "1":  # the env switch lets a 1-GPU box exercise the RCCL path
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from pdmpc.optimizer import GraphSearchHip

    options, mpa, ctl = build_world(args, 0 if sharded else rank)
    options.device = local_rank
    optimizer = GraphSearchHip(options)
    optimizer._ensure_mpa(mpa)
    h = optimizer.handle
    problems = record_steps(options, mpa, ctl, optimizer, args.skip, args.record, args.instances if explore else 0)
    S = len(problems)
    parts = None
    full_problems = problems
    if sharded and dist is not None and args.shard == "components":
        from pdmpc.distributed import partition_components, sub_problem

        # every rank recorded the same closed loop; now it keeps only the components assigned to it
        # longest-processing-time assignment by the work the searches took in the closed loop (pops + 1), not by vehicle count
        parts = [partition_components(p["preds"], world, weights=[w + 1 for w in p["pops"]] if "pops" in p else None) for p in full_problems]
        problems = [sub_problem(p, parts[b][rank]) for b, p in enumerate(full_problems)]
    # keep every recorded step resident in HBM (one bank each) and collect its algorithmic bytes
    bytes_per_bank, pops_per_bank, nodes_per_bank = [], [], []
    status_counts = {"ok": 0, "exhausted": 0, "arena_overflow": 0, "error": 0}
    h.allow_overflow = True  # counted below (and grown away), not raised
    t_host = time.perf_counter()
    t_grow = 0.0
    for b, prob in enumerate(problems):
        h.select_bank(b)
        fb = [f if f is not None else [] for f in prob["fallback"]]
        h.pack_step(prob["iters"], prob["preds"], fb)
        while True:
            h.launch()
            recs = h.fetch(len(prob["iters"]))
            if not (recs["status"] == 2).any():
                break
            # the reference's tree is unbounded (Tree.m:54-70): never measure truncated searches -- double the arenas, plan again
            tg = time.perf_counter()
            h.grow_arena(2 * h.arena_nodes()[0])
            t_grow += time.perf_counter() - tg
        status_counts["ok"] += int((recs["status"] == 0).sum())
        status_counts["exhausted"] += int((recs["status"] == 1).sum())
        status_counts["error"] += int((recs["status"] < 0).sum())
        st = h.stats()
        bytes_per_bank.append(st["algorithmic_bytes"])
        pops_per_bank.append(st["nodes_popped"])
        nodes_per_bank.append(st["nodes_generated"])
    # a bank recorded before the arenas grew replays in the grown arenas: same searches, none of them truncated
    lds_bytes = h.stats()["lds_bytes"]
    host_buffer_ms = 1e3 * (time.perf_counter() - t_host - t_grow) / max(S, 1)  # pack (host buffers -> HBM) + launch + fetch + stats

    planner = None
    gather_bufs = None
    if sharded and dist is not None and args.shard == "levels":
        from pdmpc.distributed import HipRangePlanner, plan_step_sharded

        planner = HipRangePlanner(optimizer, mpa, torch.device("cuda", local_rank))
    elif parts is not None:
        from pdmpc.distributed import REC_BYTES

        per = max(max(len(q) for q in pb) for pb in parts)
        gather_bufs = (
            torch.zeros(max(per, 1) * REC_BYTES, dtype=torch.uint8, device="cuda"),
            torch.zeros(max(per, 1) * REC_BYTES * world, dtype=torch.uint8, device="cuda"),
        )

    def one_step(i):
        if planner is not None:
            plan_step_sharded(problems[i % S], planner, dist, rank, world, resident_bank=i % S, fetch=False)
            h.synchronize()
            return
        h.select_bank(i % S)
        h.launch()
        if gather_bufs is not None:
            # end-of-step exchange: every rank receives the records of all components (one RCCL all-gather over xGMI)
            h.export_results(0, len(problems[i % S]["iters"]), gather_bufs[0].data_ptr())  # waits for the kernel
            dist.all_gather_into_tensor(gather_bufs[1], gather_bufs[0])
            torch.cuda.current_stream().synchronize()
            return
        h.synchronize()

    for i in range(args.warmup):
        one_step(i)
    h.reset_stats()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    lat = []
    t_begin = time.perf_counter()
    for i in range(args.steps):
        t0 = time.perf_counter()
        one_step(i)
        lat.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_begin
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        elapsed = float(t.item())
    st = h.stats()
    kernel_ms = st["kernel_ms"]
    n_launch = st["n_launches"]
    alg_bytes = sum(bytes_per_bank[i % S] for i in range(args.steps))
    pops = sum(pops_per_bank[i % S] for i in range(args.steps))
    nodes = sum(nodes_per_bank[i % S] for i in range(args.steps))
    achieved = (alg_bytes / max(n_launch, 1)) / ((kernel_ms / max(n_launch, 1)) * 1e-3) / 1e9 if kernel_ms > 0 else 0.0

    # ---- the closed loop a caller of the boundary sees: the native step controller (csrc/step_controller.cpp) drives the same
    # scenario, every step = build the step problem on the host + pack (H2D) + one launch + fetch (D2H) + apply, no replay and
    # no interpreter on the path.  Reported next to `value`, never as `value`.
    host_inclusive = None
    if not explore and dist is None and not args.no_host_inclusive:
        from pdmpc.native_controller import NativeController
        from pdmpc.road_network import commonroad_scenario

        tiles = max(1, (args.vehicles + 19) // 20)
        nat = NativeController(options, commonroad_scenario(options, seed=args.seed + (0 if sharded else rank), tiles=tiles), mpa, h, coupling="distance",
                               priority_strategy=args.priorities)
        nat.run(args.skip)
        ms = nat.run(args.steps)
        host_inclusive = {
            "value": 1e3 / float(np.mean(ms)),
            "unit": "MPC steps/s",
            "ms_per_step": float(np.mean(ms)),
            "p50_latency_ms": float(np.median(ms)),
            "p99_latency_ms": float(np.sort(ms)[min(len(ms) - 1, int(0.99 * len(ms)))]),
            "what": "closed loop through the C ABI (pdmpc_controller_run): host step logic in C++ + pack + H2D + one launch + D2H + apply per step, %d steps after %d" % (args.steps, args.skip),
        }
        nat.close()
    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json" if args.workload == "c2" else "none")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
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
            "host_buffer_ms_per_step": host_buffer_ms,  # PCIe-inclusive path incl. Python marshalling (never `value`)
            "value_host_inclusive": host_inclusive["value"] if host_inclusive else None,
            "host_inclusive": host_inclusive,
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d vehicles on the CPM-lab road network (labmap fixture%s), Hp %d, InterX checker, %s MPA, "
                "distance coupling, %s priorities, %s; %d recorded closed-loop steps replayed from HBM%s"
                % (args.workload.upper() + (" (%d prioritizations of each step flattened into one batch)" % args.instances if explore else ""),
                   args.vehicles, ", tiled" if sharded and not explore else "", args.hp, args.mpa, args.priorities,
                   "levels sharded over ranks with one all-gather per level" if planner is not None else ("coupling-graph components sharded over ranks, one launch per rank and step, one all-gather of results" if gather_bufs is not None else "one launch per step"), S,
                   "" if sharded else "; per GPU one independent network"),
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
                "traffic_source": "profiles/r02_pmc_traffic.json: FETCH_SIZE + WRITE_SIZE of the timed launches of the default command, search kernel + the helper kernel next to it (separate rocprofv3 --pmc passes, tools/collect_profiles.sh)" if traffic is not None else None,
                "kernel": "pdmpc_frontier_kernel" if st["kernel"] == 1 else "pdmpc_search_kernel",
                "kernel_ms_avg": kernel_ms / max(n_launch, 1),
                "algorithmic_bytes_per_launch": alg_bytes / max(n_launch, 1),
                "launches": n_launch,
                "lds_bytes_per_workgroup": lds_bytes,
                "open_list": "unordered near / far lists, rounds of the smallest keys (frontier kernel)" if st["kernel"] == 1 else ("block-min queue, %d keys in LDS" % st["queue_ring_entries"] if st["queue_mode"] == 1 else "binary heap"),
            },
            # every plan of the recorded steps by outcome; arena_overflow must be 0 (the reference's tree is unbounded, Tree.m:54-70)
            "status_counts": status_counts,
            "arena_nodes_per_vehicle": h.arena_nodes()[0],
            "counters": {
                "nodes_popped_per_s": pops / elapsed,
                "nodes_generated_per_s": nodes / elapsed,
                "nodes_popped_per_step": pops / args.steps,
                "edge_checks_per_s": st["edge_checks"] / elapsed,
                "segment_pair_tests_per_s": st["segment_pair_tests"] / elapsed,
                "speculation_arrivals_per_step": st["speculation_arrivals"] / args.steps,
                "speculation_restarts_per_step": st["speculation_restarts"] / args.steps,
                "speculation_wasted_pops_per_step": st["speculation_wasted_pops"] / args.steps,
                "nodes_processed_per_step": st["nodes_processed"] / args.steps,  # frontier kernel: edges evaluated; nodes_popped of them are the reference's pops
                "rounds_per_step": st["rounds"] / args.steps,
                "shared_rounds_per_step": st["shared_rounds"] / args.steps,  # rounds whose edge checks helper workgroups took part in
                "helper_checked_per_step": st["helper_checked"] / args.steps,
                "entries_dropped_per_step": st["entries_dropped"] / args.steps,
                "dropped_counted_as_pops_per_step": st["dropped_counted_as_pops"] / args.steps,
                "queue_fallbacks_per_step": st["queue_fallbacks"] / args.steps,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(options, mpa, full_problems, args.cpu_budget_s)
        print(json.dumps(out))
        if status_counts["arena_overflow"] or status_counts["error"]:
            raise SystemExit("bench.py: %d plans overflowed their arena, %d carried an error status -- the measurement is void" % (status_counts["arena_overflow"], status_counts["error"]))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
