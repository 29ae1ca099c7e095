"""Loader and thin object wrapper for libpdmpc_hip.so (the C ABI of include/pdmpc.h).

There is deliberately NO CPU fallback: if the shared library is missing or no gfx950 device is
present, construction raises.  The library is built in-tree by `__graft_entry__.build()` /
`make -C p-dmpc_amd/csrc`.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PDMPC_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libpdmpc_hip.so")

_LIB = None

EXPORTS = [
    "pdmpc_create",
    "pdmpc_destroy",
    "pdmpc_upload_mpa",
    "pdmpc_get_config",
    "pdmpc_plan_batch",
    "pdmpc_plan_batch_sampled",
    "pdmpc_pack_batch",
    "pdmpc_launch_packed",
    "pdmpc_launch_range",
    "pdmpc_begin_step",
    "pdmpc_select_bank",
    "pdmpc_reset_stats",
    "pdmpc_fetch_results",
    "pdmpc_synchronize",
    "pdmpc_set_safe_launch",
    "pdmpc_pack_step",
    "pdmpc_plan_step",
    "pdmpc_plan_step_literal",
    "pdmpc_set_arena_limit",
    "pdmpc_grow_arena",
    "pdmpc_arena_nodes",
    "pdmpc_result_device_buffer",
    "pdmpc_import_results",
    "pdmpc_export_results",
    "pdmpc_export_results_async",
    "pdmpc_stream",
    "pdmpc_get_last_stats",
    "pdmpc_group_create",
    "pdmpc_group_create_ex",
    "pdmpc_set_step_weights",
    "pdmpc_set_device_share",
    "pdmpc_last_call_timing",
    "pdmpc_plan_step_lean",
    "pdmpc_fetch_records_at",
    "pdmpc_controller_last_timing",
    "pdmpc_controller_timing_sum",
    "pdmpc_controller_explore_follow_own",
    "pdmpc_group_collective",
    "pdmpc_group_destroy",
    "pdmpc_group_size",
    "pdmpc_group_handle",
    "pdmpc_group_grow_arena",
    "pdmpc_group_upload_mpa",
    "pdmpc_group_plan_step",
    "pdmpc_group_pack_step",
    "pdmpc_group_launch",
    "pdmpc_group_fetch",
    "pdmpc_group_partition",
    "pdmpc_group_last_timing",
    "pdmpc_debug_heap_script",
    "pdmpc_debug_pop_trace",
    "pdmpc_debug_tree",
    "pdmpc_debug_raw_tree",
    "pdmpc_debug_edge_check",
    "pdmpc_debug_progress",
    "pdmpc_debug_counters",
    "pdmpc_controller_create",
    "pdmpc_controller_destroy",
    "pdmpc_controller_step",
    "pdmpc_controller_run",
    "pdmpc_controller_build_step",
    "pdmpc_controller_apply",
    "pdmpc_controller_problem",
    "pdmpc_controller_state",
    "pdmpc_controller_records",
    "pdmpc_controller_last_error",
    "pdmpc_exploration_permutations",
    "pdmpc_controller_explore_build",
    "pdmpc_controller_explore_problem",
    "pdmpc_controller_explore_choose",
    "pdmpc_controller_explore_step",
    "pdmpc_controller_explore_run",
    "pdmpc_controller_explore_result",
    "pdmpc_last_error",
    "pdmpc_version",
]


class BackendError(RuntimeError):
    pass


def load_library(path=None):
    """dlopen the HIP backend and declare every prototype of include/pdmpc.h."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or LIB_PATH
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64; if this library pulled in the system
    # copy first, torch.cuda would later find "no HIP GPUs".  Importing torch first makes both share torch's copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise BackendError(
            "HIP backend %s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % p
        )
    L = C.CDLL(p)
    H = C.c_void_p
    L.pdmpc_create.argtypes = [C.POINTER(abi.Config), C.POINTER(H)]
    L.pdmpc_destroy.argtypes = [H]
    L.pdmpc_upload_mpa.argtypes = [H, C.POINTER(abi.Mpa)]
    L.pdmpc_get_config.argtypes = [H, C.POINTER(abi.Config), C.POINTER(C.c_int32)]
    L.pdmpc_plan_batch.argtypes = [H, C.c_int32, C.POINTER(abi.VehicleIn), C.POINTER(abi.VehicleOut)]
    L.pdmpc_pack_batch.argtypes = [H, C.c_int32, C.POINTER(abi.VehicleIn)]
    L.pdmpc_launch_packed.argtypes = [H]
    L.pdmpc_launch_range.argtypes = [H, C.c_int32, C.c_int32]
    L.pdmpc_begin_step.argtypes = [H]
    L.pdmpc_select_bank.argtypes = [H, C.c_int32]
    L.pdmpc_reset_stats.argtypes = [H]
    L.pdmpc_fetch_results.argtypes = [H, C.c_int32, C.POINTER(abi.VehicleOut)]
    L.pdmpc_synchronize.argtypes = [H]
    L.pdmpc_set_safe_launch.argtypes = [H, C.c_int32]
    L.pdmpc_pack_step.argtypes = [H, C.c_int32, C.POINTER(abi.VehicleIn), abi.c_int32_p, abi.c_int32_p, C.POINTER(abi.PolygonSet)]
    L.pdmpc_plan_step.argtypes = [H, C.c_int32, C.POINTER(abi.VehicleIn), abi.c_int32_p, abi.c_int32_p, C.POINTER(abi.PolygonSet), C.POINTER(abi.VehicleOut)]
    L.pdmpc_plan_step_literal.argtypes = L.pdmpc_plan_step.argtypes
    L.pdmpc_set_arena_limit.argtypes = [H, C.c_int32]
    L.pdmpc_grow_arena.argtypes = [H, C.c_int32]
    L.pdmpc_arena_nodes.argtypes = [H, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    L.pdmpc_result_device_buffer.argtypes = [H, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.pdmpc_import_results.argtypes = [H, C.c_int32, C.c_int32, C.c_void_p]
    L.pdmpc_export_results.argtypes = [H, C.c_int32, C.c_int32, C.c_void_p]
    L.pdmpc_export_results_async.argtypes = [H, C.c_int32, C.c_int32, C.c_void_p]
    L.pdmpc_stream.argtypes = [H, C.POINTER(C.c_void_p)]
    L.pdmpc_plan_batch_sampled.argtypes = [H, C.c_int32, C.POINTER(abi.VehicleIn), C.POINTER(C.c_uint32), C.POINTER(abi.VehicleOut)]
    L.pdmpc_get_last_stats.argtypes = [H, C.POINTER(abi.Stats)]
    L.pdmpc_debug_heap_script.argtypes = [H, C.c_int32, abi.c_int32_p, abi.c_int32_p, abi.c_double_p, C.c_int32, abi.c_int32_p, abi.c_int32_p, abi.c_double_p, abi.c_double_p]
    L.pdmpc_debug_pop_trace.argtypes = [H, C.c_int32, C.c_int32, abi.c_int32_p, abi.c_int32_p]
    L.pdmpc_debug_tree.argtypes = [H, C.c_int32, C.c_int32] + [abi.c_double_p] * 5 + [abi.c_int32_p] * 4
    L.pdmpc_debug_raw_tree.argtypes = [H, C.c_int32, C.c_int32] + [abi.c_double_p] * 5 + [abi.c_int32_p] * 3 + [abi.c_double_p, abi.c_uint8_p, abi.c_int32_p]
    L.pdmpc_debug_edge_check.argtypes = [H, C.c_int32, C.c_int32, abi.c_int32_p, abi.c_double_p, abi.c_double_p, abi.c_int32_p, abi.c_double_p, abi.c_double_p, abi.c_int32_p]
    L.pdmpc_debug_progress.argtypes = [H, C.c_int32, C.POINTER(C.c_uint32)]
    L.pdmpc_last_error.restype = C.c_char_p
    L.pdmpc_version.restype = C.c_char_p
    for name in EXPORTS:
        if name.startswith("pdmpc_controller_"):
            continue  # declared by pdmpc.native_controller
        if name not in ("pdmpc_last_error", "pdmpc_version"):
            getattr(L, name).restype = C.c_int
    if path is None:
        _LIB = L
    return L


def _check(L, rc, what):
    if rc != 0:
        msg = L.pdmpc_last_error()
        raise BackendError("%s failed with status %d: %s" % (what, rc, msg.decode() if msg else ""))


SHARD_AUTO, SHARD_COMPONENTS, SHARD_LEVELS = 0, 1, 2
COLLECTIVE_AUTO, COLLECTIVE_RCCL, COLLECTIVE_COPY = 0, 1, 2  # include/pdmpc.h: how the ranks of a group exchange their records


def group_partition(preds, world, mode=SHARD_AUTO, weights=None):
    """pdmpc_group_partition (no GPU needed): per vehicle the device of its whole component (-1: planned by levels over all
    devices), its level (1-based, 0 for whole components) and the device of its block within that level (-1)."""
    L = load_library()
    n = len(preds)
    off = np.zeros(n + 1, dtype=np.int32)
    for i, p in enumerate(preds):
        off[i + 1] = off[i] + len(p)
    idx = np.array([j for p in preds for j in p] + [0], dtype=np.int32)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
    rank_of, level_of, block = (np.zeros(max(n, 1), dtype=np.int32) for _ in range(3))
    L.pdmpc_group_partition.argtypes = [C.c_int32, abi.c_int32_p, abi.c_int32_p, abi.c_double_p, C.c_int32, C.c_int32, abi.c_int32_p, abi.c_int32_p, abi.c_int32_p]
    _check(L, L.pdmpc_group_partition(n, off.ctypes.data_as(abi.c_int32_p), idx.ctypes.data_as(abi.c_int32_p), None if w is None else w.ctypes.data_as(abi.c_double_p), world, mode,
                                      rank_of.ctypes.data_as(abi.c_int32_p), level_of.ctypes.data_as(abi.c_int32_p), block.ctypes.data_as(abi.c_int32_p)), "pdmpc_group_partition")
    return rank_of[:n], level_of[:n], block[:n]


class Group:
    """One pdmpc_group: a handle per GPU of this process, bound by an RCCL communicator; plan_step plans a time step over them
    (include/pdmpc.h: pdmpc_group_*)."""

    def __init__(self, options, n_devices=1, devices=None, checker=None, collective=COLLECTIVE_AUTO):
        """devices: HIP ordinals of the ranks (None: 0 .. n_devices - 1).  A device listed more than once makes LOGICAL ranks that share
        a GPU (collective = peer copies instead of RCCL): how the multi-rank protocol is exercised on a 1-GPU box."""
        self.L = load_library()
        self.options = options
        self.Hp = options.Hp
        if checker is None:
            checker = abi.CHECK_INTERX if options.are_any_obstacles_non_convex else abi.CHECK_SAT
        self.cfg = abi.Config(Hp=options.Hp, checker=checker, dt_seconds=options.dt_seconds, device=0, max_nodes=options.max_nodes, max_vehicles=options.max_vehicles,
                              trace_pops=0)
        self.g = C.c_void_p()
        devs = None if devices is None else (C.c_int32 * n_devices)(*devices)
        self.L.pdmpc_group_create_ex.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        _check(self.L, self.L.pdmpc_group_create_ex(C.byref(self.cfg), n_devices, devs, collective, C.byref(self.g)), "pdmpc_group_create_ex")
        self.n_devices = n_devices
        c = C.c_int32(0)
        self.L.pdmpc_group_collective.argtypes = [C.c_void_p, C.c_void_p]
        _check(self.L, self.L.pdmpc_group_collective(self.g, C.byref(c)), "pdmpc_group_collective")
        self.collective = {COLLECTIVE_RCCL: "rccl", COLLECTIVE_COPY: "copy"}[c.value]
        self._mpa_keep = None

    def close(self):
        if self.g:
            self.L.pdmpc_group_destroy(self.g)
            self.g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_mpa(self, mpa):
        s, keep = abi.pack_mpa(mpa)
        self.L.pdmpc_group_upload_mpa.argtypes = [C.c_void_p, C.c_void_p]
        _check(self.L, self.L.pdmpc_group_upload_mpa(self.g, C.byref(s)), "pdmpc_group_upload_mpa")
        self._mpa_keep = keep

    def plan_step(self, iters, predecessors, fallback_shapes=None, weights=None, mode=SHARD_AUTO):
        n = len(iters)
        arr, off, idx, fb, keep = Handle._step_args(self, iters, predecessors, fallback_shapes)
        out = abi.out_array(n)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        self.L.pdmpc_group_plan_step.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, abi.c_int32_p, abi.c_int32_p, C.c_void_p, abi.c_double_p, C.c_int32, C.c_void_p]
        _check(self.L, self.L.pdmpc_group_plan_step(self.g, n, arr, off.ctypes.data_as(abi.c_int32_p), idx.ctypes.data_as(abi.c_int32_p), fb,
                                                    None if w is None else w.ctypes.data_as(abi.c_double_p), mode, abi.out_ptr(out)), "pdmpc_group_plan_step")
        del keep
        return out[:n]

    def pack_step(self, bank, iters, predecessors, fallback_shapes=None, weights=None, mode=SHARD_AUTO):
        """Make a step resident on the devices (group bank `bank`); launch(bank) plans it, fetch(bank, n) reads the records."""
        n = len(iters)
        arr, off, idx, fb, keep = Handle._step_args(self, iters, predecessors, fallback_shapes)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        self.L.pdmpc_group_pack_step.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, abi.c_int32_p, abi.c_int32_p, C.c_void_p, abi.c_double_p, C.c_int32]
        _check(self.L, self.L.pdmpc_group_pack_step(self.g, bank, n, arr, off.ctypes.data_as(abi.c_int32_p), idx.ctypes.data_as(abi.c_int32_p), fb,
                                                    None if w is None else w.ctypes.data_as(abi.c_double_p), mode), "pdmpc_group_pack_step")
        del keep

    def grow_arena(self, max_nodes):
        """Arenas of at least max_nodes nodes per vehicle on every device (the resident path does not grow them by itself)."""
        self.L.pdmpc_group_grow_arena.argtypes = [C.c_void_p, C.c_int32]
        _check(self.L, self.L.pdmpc_group_grow_arena(self.g, int(max_nodes)), "pdmpc_group_grow_arena")

    def launch(self, bank):
        self.L.pdmpc_group_launch.argtypes = [C.c_void_p, C.c_int32]
        _check(self.L, self.L.pdmpc_group_launch(self.g, bank), "pdmpc_group_launch")

    def fetch(self, bank, n):
        out = abi.out_array(n)
        self.L.pdmpc_group_fetch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        _check(self.L, self.L.pdmpc_group_fetch(self.g, bank, n, abi.out_ptr(out)), "pdmpc_group_fetch")
        return out[:n]

    def timing(self):
        t = (C.c_double * 6)()
        self.L.pdmpc_group_last_timing.argtypes = [C.c_void_p, C.c_void_p]
        _check(self.L, self.L.pdmpc_group_last_timing(self.g, t), "pdmpc_group_last_timing")
        return dict(zip(("total", "partition", "pack", "enqueue", "wait", "read_back"), t))

    def reset_stats(self):
        for r in range(self.n_devices):
            h = C.c_void_p()
            self.L.pdmpc_group_handle.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
            _check(self.L, self.L.pdmpc_group_handle(self.g, r, C.byref(h)), "pdmpc_group_handle")
            _check(self.L, self.L.pdmpc_reset_stats(h), "pdmpc_reset_stats")

    def stats_all(self):
        """Statistics over the whole group: counts summed over the ranks, kernel_ms / n_launches of the rank whose kernels ran longest
        (the ranks' launches run side by side), per-rank values under "per_rank"."""
        per = [self.stats(r) for r in range(self.n_devices)]
        tot = dict(per[max(range(len(per)), key=lambda r: per[r]["kernel_ms"])])
        for k in ("edge_checks", "segment_pair_tests", "nodes_processed", "rounds", "shared_rounds", "helper_checked", "bad_status_plans", "queue_fallbacks",
                  "speculation_arrivals", "safe_replans"):
            if k in tot:
                tot[k] = sum(p[k] for p in per)
        tot["per_rank"] = per
        return tot

    def stats(self, rank=0):
        h = C.c_void_p()
        self.L.pdmpc_group_handle.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        _check(self.L, self.L.pdmpc_group_handle(self.g, rank, C.byref(h)), "pdmpc_group_handle")
        st = abi.Stats()
        _check(self.L, self.L.pdmpc_get_last_stats(h, C.byref(st)), "pdmpc_get_last_stats")
        return {k: getattr(st, k) for k, _ in abi.Stats._fields_}


class Handle:
    """One pdmpc_handle: bound to one GPU, owns the uploaded MPA and all device buffers."""

    def __init__(self, options, checker=None):
        self.L = load_library()
        self.options = options
        self.Hp = options.Hp
        if checker is None:
            checker = abi.CHECK_INTERX if options.are_any_obstacles_non_convex else abi.CHECK_SAT
        self.cfg = abi.Config(
            Hp=options.Hp,
            checker=checker,
            dt_seconds=options.dt_seconds,
            device=options.device,
            max_nodes=options.max_nodes,
            max_vehicles=options.max_vehicles,
            trace_pops=options.trace_pops,
        )
        self.allow_overflow = False  # tests of the overflow status itself switch this on
        self.h = C.c_void_p()
        _check(self.L, self.L.pdmpc_create(C.byref(self.cfg), C.byref(self.h)), "pdmpc_create")
        self._mpa_keep = None

    def close(self):
        if self.h:
            self.L.pdmpc_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_mpa(self, mpa):
        s, keep = abi.pack_mpa(mpa)
        _check(self.L, self.L.pdmpc_upload_mpa(self.h, C.byref(s)), "pdmpc_upload_mpa")
        self._mpa_keep = keep

    def plan_batch(self, iters):
        """list[VehicleIter] -> numpy records (abi.VEHICLE_OUT_DTYPE)."""
        n = len(iters)
        arr, keep = abi.pack_vehicles(iters, self.Hp)
        out = abi.out_array(n)
        _check(self.L, self.L.pdmpc_plan_batch(self.h, n, arr, abi.out_ptr(out)), "pdmpc_plan_batch")
        del keep
        return self._checked(out[:n])

    def plan_batch_sampled(self, iters, seeds):
        """The sampled optimizer for one computation level; seeds[i] = time_step + vehicle_index (MonteCarloTreeSearch.m:32)."""
        n = len(iters)
        arr, keep = abi.pack_vehicles(iters, self.Hp)
        out = abi.out_array(n)
        sd = (C.c_uint32 * max(n, 1))(*[int(s) for s in seeds])
        _check(self.L, self.L.pdmpc_plan_batch_sampled(self.h, n, arr, sd, abi.out_ptr(out)), "pdmpc_plan_batch_sampled")
        del keep
        return out[:n]

    # ---- device-resident path ----
    def pack_batch(self, iters):
        arr, keep = abi.pack_vehicles(iters, self.Hp)
        _check(self.L, self.L.pdmpc_pack_batch(self.h, len(iters), arr), "pdmpc_pack_batch")
        del keep

    def set_step_weights(self, weights):
        """Expected work per vehicle of the next packed step (pdmpc_set_step_weights): its searches go out by priority, not slot order."""
        w = np.ascontiguousarray(weights, dtype=np.float64)
        self.L.pdmpc_set_step_weights.argtypes = [C.c_void_p, C.c_int32, abi.c_double_p]
        _check(self.L, self.L.pdmpc_set_step_weights(self.h, len(w), w.ctypes.data_as(abi.c_double_p)), "pdmpc_set_step_weights")

    def pack_step(self, iters, predecessors, fallback_shapes=None, weights=None):
        """predecessors: list (per vehicle) of lists of 0-based vehicle indices in this batch."""
        n = len(iters)
        arr, off, idx, fb, keep = self._step_args(iters, predecessors, fallback_shapes)
        if weights is not None:
            self.set_step_weights(weights)
        _check(
            self.L,
            self.L.pdmpc_pack_step(self.h, n, arr, off.ctypes.data_as(abi.c_int32_p), idx.ctypes.data_as(abi.c_int32_p), fb),
            "pdmpc_pack_step",
        )
        del keep

    def _step_args(self, iters, predecessors, fallback_shapes):
        n = len(iters)
        arr, keep = abi.pack_vehicles(iters, self.Hp)
        off = np.zeros(n + 1, dtype=np.int32)
        for i, p in enumerate(predecessors):
            off[i + 1] = off[i] + len(p)
        idx = np.array([j for p in predecessors for j in p] + [0], dtype=np.int32)
        fb = None
        if fallback_shapes is not None:
            fb = (abi.PolygonSet * n)()
            for i, shapes in enumerate(fallback_shapes):
                fb[i] = abi.pack_polygon_set(list(shapes), keep)
        return arr, off, idx, fb, keep

    def plan_step(self, iters, predecessors, fallback_shapes=None, weights=None):
        """A whole time step in one call (pdmpc_plan_step): pack + launch + fetch, arenas grow if a search needs it."""
        n = len(iters)
        arr, off, idx, fb, keep = self._step_args(iters, predecessors, fallback_shapes)
        out = abi.out_array(n)
        if weights is not None:
            self.set_step_weights(weights)
        _check(
            self.L,
            self.L.pdmpc_plan_step(self.h, n, arr, off.ctypes.data_as(abi.c_int32_p), idx.ctypes.data_as(abi.c_int32_p), fb, abi.out_ptr(out)),
            "pdmpc_plan_step",
        )
        del keep
        return self._checked(out[:n])

    def step_args(self, iters, predecessors, fallback_shapes=None):
        """The marshalled arguments of pdmpc_plan_step / pdmpc_plan_step_literal, reusable across calls (bench.py times the calls,
        not the Python marshalling): (n, vehicle array, pred_offset, pred_index, fallback sets, keep-alive)."""
        arr, off, idx, fb, keep = self._step_args(iters, predecessors, fallback_shapes)
        return len(iters), arr, off, idx, fb, keep

    def plan_step_literal(self, args):
        """One pdmpc_plan_batch(h, 1, ...) per vehicle in slot (= kahn) order with the hand-over on the host: what GraphSearchHip.m
        gives an unmodified controller (pdmpc_plan_step_literal).  args = step_args(...)."""
        n, arr, off, idx, fb, _ = args
        out = abi.out_array(n)
        _check(
            self.L,
            self.L.pdmpc_plan_step_literal(self.h, n, arr, off.ctypes.data_as(abi.c_int32_p), idx.ctypes.data_as(abi.c_int32_p), fb, abi.out_ptr(out)),
            "pdmpc_plan_step_literal",
        )
        return self._checked(out[:n])

    def _checked(self, recs):
        """Only PDMPC_EXHAUSTED is a planning result (info.is_exhausted); anything else is an error of this backend."""
        st = np.asarray(recs["status"])
        if ((st != abi.OK) & (st != abi.EXHAUSTED)).any() and not self.allow_overflow:
            bad = int(st[(st != abi.OK) & (st != abi.EXHAUSTED)][0])
            if bad == abi.ARENA_OVERFLOW:
                raise BackendError("a search outgrew its arena (%d nodes per vehicle) and the arena limit forbids growing it" % self.arena_nodes()[0])
            raise BackendError("device-side error status %d in a result record (predecessor wait timed out?)" % bad)
        return recs

    def set_arena_limit(self, max_nodes_limit):
        _check(self.L, self.L.pdmpc_set_arena_limit(self.h, int(max_nodes_limit)), "pdmpc_set_arena_limit")

    def grow_arena(self, max_nodes):
        _check(self.L, self.L.pdmpc_grow_arena(self.h, int(max_nodes)), "pdmpc_grow_arena")

    def arena_nodes(self):
        n, r = C.c_int32(), C.c_int64()
        _check(self.L, self.L.pdmpc_arena_nodes(self.h, C.byref(n), C.byref(r)), "pdmpc_arena_nodes")
        return n.value, r.value

    def launch(self):
        _check(self.L, self.L.pdmpc_launch_packed(self.h), "pdmpc_launch_packed")

    def launch_range(self, first, count):
        _check(self.L, self.L.pdmpc_launch_range(self.h, first, count), "pdmpc_launch_range")

    def begin_step(self):
        _check(self.L, self.L.pdmpc_begin_step(self.h), "pdmpc_begin_step")

    def select_bank(self, bank):
        _check(self.L, self.L.pdmpc_select_bank(self.h, bank), "pdmpc_select_bank")

    def reset_stats(self):
        _check(self.L, self.L.pdmpc_reset_stats(self.h), "pdmpc_reset_stats")

    def synchronize(self):
        _check(self.L, self.L.pdmpc_synchronize(self.h), "pdmpc_synchronize")

    def set_safe_launch(self, on):
        """Every launch in slices that are resident as a whole (forward progress without any assumption on the dispatch order)."""
        _check(self.L, self.L.pdmpc_set_safe_launch(self.h, 1 if on else 0), "pdmpc_set_safe_launch")

    def fetch(self, n):
        out = abi.out_array(n)
        _check(self.L, self.L.pdmpc_fetch_results(self.h, n, abi.out_ptr(out)), "pdmpc_fetch_results")
        return self._checked(out[:n])

    def result_device_buffer(self):
        p = C.c_void_p()
        nb = C.c_size_t()
        _check(self.L, self.L.pdmpc_result_device_buffer(self.h, C.byref(p), C.byref(nb)), "pdmpc_result_device_buffer")
        return p.value, nb.value

    def import_results(self, first, n, dev_ptr):
        _check(self.L, self.L.pdmpc_import_results(self.h, first, n, C.c_void_p(dev_ptr)), "pdmpc_import_results")

    def export_results(self, first, n, dev_ptr):
        _check(self.L, self.L.pdmpc_export_results(self.h, first, n, C.c_void_p(dev_ptr)), "pdmpc_export_results")

    def export_results_async(self, first, n, dev_ptr):
        _check(self.L, self.L.pdmpc_export_results_async(self.h, first, n, C.c_void_p(dev_ptr)), "pdmpc_export_results_async")

    def stream_ptr(self):
        p = C.c_void_p()
        _check(self.L, self.L.pdmpc_stream(self.h, C.byref(p)), "pdmpc_stream")
        return p.value or 0

    def stats(self):
        s = abi.Stats()
        _check(self.L, self.L.pdmpc_get_last_stats(self.h, C.byref(s)), "pdmpc_get_last_stats")
        return {name: getattr(s, name) for name, _ in abi.Stats._fields_}

    def heap_script(self, ops, ids, keys, lds_entries=4096):
        """Run a push/pop script on the device open list -> (popped ids, cycles per pop, cycles per push)."""
        ops = np.ascontiguousarray(ops, dtype=np.int32)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        keys = np.ascontiguousarray(keys, dtype=np.float64)
        out = np.zeros(max(len(ops), 1), dtype=np.int32)
        n = C.c_int32()
        cp, cq = C.c_double(), C.c_double()
        _check(
            self.L,
            self.L.pdmpc_debug_heap_script(self.h, len(ops), ops.ctypes.data_as(abi.c_int32_p), ids.ctypes.data_as(abi.c_int32_p), keys.ctypes.data_as(abi.c_double_p),
                                           lds_entries, out.ctypes.data_as(abi.c_int32_p), C.byref(n), C.byref(cp), C.byref(cq)),
            "pdmpc_debug_heap_script",
        )
        return out[: n.value].copy(), cp.value, cq.value

    def pop_trace(self, vehicle, capacity=1 << 16):
        ids = np.zeros(capacity, dtype=np.int32)
        n = C.c_int32()
        _check(self.L, self.L.pdmpc_debug_pop_trace(self.h, vehicle, capacity, ids.ctypes.data_as(abi.c_int32_p), C.byref(n)), "pdmpc_debug_pop_trace")
        return ids[: min(n.value, capacity)].copy()

    def raw_tree(self, vehicle, capacity=1 << 20):
        """The arena as the kernel left it: node arrays + key + validity (see pdmpc_debug_raw_tree)."""
        f = {k: np.zeros(capacity) for k in ("x", "y", "yaw", "g", "h", "key")}
        i = {k: np.zeros(capacity, dtype=np.int32) for k in ("trim", "k", "parent")}
        val = np.zeros(capacity, dtype=np.uint8)
        n = C.c_int32()
        args = [f[k].ctypes.data_as(abi.c_double_p) for k in ("x", "y", "yaw", "g", "h")] + [i[k].ctypes.data_as(abi.c_int32_p) for k in ("trim", "k", "parent")]
        args += [f["key"].ctypes.data_as(abi.c_double_p), val.ctypes.data_as(abi.c_uint8_p)]
        _check(self.L, self.L.pdmpc_debug_raw_tree(self.h, vehicle, capacity, *args, C.byref(n)), "pdmpc_debug_raw_tree")
        nn = min(n.value, capacity)
        d = {k: v[:nn].copy() for k, v in f.items()}
        d.update({k: v[:nn].copy() for k, v in i.items()})
        d["validity"] = val[:nn].copy()
        return d

    def edge_check(self, mode, a_list, b_list):
        """Run a collision primitive on the device for len(a_list) cases: mode 0 InterX, 1 intersect_sat, 2 intersect_lanelet_boundary
        (b = [left, NaN, right, NaN]).  a_list / b_list: lists of (2, n) arrays.  Returns a bool array."""
        n = len(a_list)
        def flat(lst):
            off = np.zeros(n + 1, dtype=np.int32)
            for i, p in enumerate(lst):
                off[i + 1] = off[i] + np.asarray(p).shape[1]
            x = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.float64)[0] for p in lst] + [np.zeros(1)]))
            y = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.float64)[1] for p in lst] + [np.zeros(1)]))
            return off, x, y
        ao, ax, ay = flat(a_list)
        bo, bx, by = flat(b_list)
        hit = np.zeros(max(n, 1), dtype=np.int32)
        _check(
            self.L,
            self.L.pdmpc_debug_edge_check(self.h, mode, n, ao.ctypes.data_as(abi.c_int32_p), ax.ctypes.data_as(abi.c_double_p), ay.ctypes.data_as(abi.c_double_p),
                                          bo.ctypes.data_as(abi.c_int32_p), bx.ctypes.data_as(abi.c_double_p), by.ctypes.data_as(abi.c_double_p), hit.ctypes.data_as(abi.c_int32_p)),
            "pdmpc_debug_edge_check",
        )
        return hit[:n] != 0

    def debug_counters(self):
        out = (C.c_uint64 * 16)()
        self.L.pdmpc_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
        _check(self.L, self.L.pdmpc_debug_counters(self.h, out), "pdmpc_debug_counters")
        return list(out)

    def progress(self, vehicle):
        w = (C.c_uint32 * 32)()
        _check(self.L, self.L.pdmpc_debug_progress(self.h, vehicle, w), "pdmpc_debug_progress")
        return list(w)

    def tree(self, vehicle, capacity=1 << 16):
        f = {k: np.zeros(capacity) for k in ("x", "y", "yaw", "g", "h")}
        i = {k: np.zeros(capacity, dtype=np.int32) for k in ("trim", "k", "parent")}
        n = C.c_int32()
        args = [f[k].ctypes.data_as(abi.c_double_p) for k in ("x", "y", "yaw", "g", "h")] + [i[k].ctypes.data_as(abi.c_int32_p) for k in ("trim", "k", "parent")]
        _check(self.L, self.L.pdmpc_debug_tree(self.h, vehicle, capacity, *args, C.byref(n)), "pdmpc_debug_tree")
        nn = min(n.value, capacity)
        d = {k: v[:nn].copy() for k, v in f.items()}
        d.update({k: v[:nn].copy() for k, v in i.items()})
        return d
