/* pdmpc.h — C ABI of libpdmpc_hip.so: the MI355X (gfx950) backend for p-dmpc's
 * per-vehicle graph-search trajectory optimizer.
 *
 * Drop-in boundary.  The reference selects its optimizer in
 * hlc/optimizer/OptimizerInterface.m:19-34 (`get_optimizer`) and calls it through
 *     info = run_optimizer(obj, veh_index, iter, mpa, options, time_step)   (OptimizerInterface.m:14)
 * from hlc/controller/prioritized/PrioritizedController.m:335-341.  A MATLAB MEX shim (or a
 * ctypes binding) marshals the 1-vehicle `iter` slice, the `mpa` tables and
 * `options` into the plain structs below and calls pdmpc_plan_batch(); one vehicle per call is the
 * literal `run_optimizer`, n vehicles per call is one computation level of
 * hlc/controller/prioritized/PrioritizedSequentialController.m:83-91.
 *
 * Conventions
 *  - every entry point returns an int status (PDMPC_OK == 0); no exceptions cross the ABI;
 *  - all floating point is IEEE double, all indices that mirror MATLAB values are 1-based
 *    (trim indices, tree node ids, tree_path) exactly as the reference stores them;
 *  - the caller owns every host buffer; the library owns all device memory inside the handle;
 *  - a handle is bound to one GPU and is not re-entrant (one call in flight per handle), which is
 *    the reference's threading model (one blocking optimizer per controller process, main.m:43-60).
 *  - polygons are stored as the reference stores them: 2 x V column lists [x; y] whose last column
 *    repeats the first (closed), see generate_maneuver.m:46 and vectorize_all_obstacles.m:68-75.
 */
#ifndef PDMPC_H
#define PDMPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDMPC_HP_MAX 16 /* largest prediction horizon Hp accepted (BASELINE configs use 5..10) */
#define PDMPC_VMAX 8    /* columns reserved per maneuver area (reference: 5, 6 or 7; generate_maneuver.m:74-101) */

/* status codes (function results and pdmpc_vehicle_out.status) */
enum {
    PDMPC_OK = 0,
    PDMPC_EXHAUSTED = 1,        /* per vehicle: open list ran empty == info.is_exhausted (GraphSearch.m:57-61) */
    PDMPC_ARENA_OVERFLOW = 2,   /* per vehicle: search tree outgrew its arena and could not be grown further (see pdmpc_set_arena_limit);
                                   NOT an exhaustion: the reference tree is unbounded (Tree.m:54-70) and would have kept searching */
    PDMPC_ERR_INVALID = -1,     /* bad argument / inconsistent sizes */
    PDMPC_ERR_NO_DEVICE = -2,   /* no gfx950 device, or the HIP code object failed to load */
    PDMPC_ERR_HIP = -3,         /* a HIP runtime call failed; see pdmpc_last_error() */
    PDMPC_ERR_CAPACITY = -4,    /* problem does not fit the per-vehicle LDS/HBM budget of this handle */
    PDMPC_ERR_NO_MPA = -5       /* pdmpc_plan_* before pdmpc_upload_mpa */
};

/* constraint checker, chosen by OptimizerInterface.set_constraint_checker (OptimizerInterface.m:36-46)
 * from Config.are_any_obstacles_non_convex (config/Config.m:71-87) */
enum {
    PDMPC_CHECK_SAT = 0,   /* are_constraints_satisfied_sat.m  (circle scenario / convex areas) */
    PDMPC_CHECK_INTERX = 1 /* are_constraints_satisfied_interx.m + vectorize_all_obstacles.m (road networks) */
};

typedef struct pdmpc_handle pdmpc_handle;

/* options.* fields read on the hot path (config/Config.m:32-33,47) plus backend sizing */
typedef struct {
    int32_t Hp;           /* options.Hp */
    int32_t checker;      /* PDMPC_CHECK_* */
    double dt_seconds;    /* options.dt_seconds (expand_node.m:70) */
    int32_t device;       /* HIP device ordinal */
    int32_t max_nodes;    /* per-vehicle tree capacity in HBM (0 = default 32768) */
    int32_t max_vehicles; /* largest batch this handle will see (0 = default 256) */
    int32_t trace_pops;   /* unused (kept for the layout): pdmpc_debug_pop_trace reconstructs the pop sequence from the tree */
} pdmpc_config;

/* One maneuver mpa.maneuvers{i,j} (generate_maneuver.m:25-66).  area* are [2][PDMPC_VMAX]
 * row-major (row 0 = x, row 1 = y), the first n_cols columns valid. */
typedef struct {
    double dx, dy, dyaw;
    int32_t n_cols;
    int32_t _pad;
    double area[2][PDMPC_VMAX];
    double area_without_offset[2][PDMPC_VMAX];
    double area_large_offset[2][PDMPC_VMAX];
} pdmpc_maneuver;

/* The MPA tables the search reads (MotionPrimitiveAutomaton.m:5-9). */
typedef struct {
    int32_t n_trims;               /* length(mpa.trims) */
    int32_t Hp;                    /* size(transition_matrix_single, 3) */
    const uint8_t* transition;     /* [Hp][n_trims][n_trims]: transition[k][i][j] = transition_matrix_single(i+1, j+1, k+1) */
    const int32_t* maneuver_index; /* [n_trims][n_trims]: index into maneuvers[] or -1 where maneuvers{i,j} is empty */
    int32_t n_maneuvers;
    const pdmpc_maneuver* maneuvers;
} pdmpc_mpa;

/* A list of polygons/polylines: polygon p = columns offset[p] .. offset[p+1]-1 of (x, y). */
typedef struct {
    int32_t n_polygons;
    const int32_t* offset; /* [n_polygons + 1] */
    const double* x;
    const double* y;
} pdmpc_polygon_set;

/* The 1-vehicle IterationData slice (hlc/controller/common/IterationData.m:4-33, filter :95-114)
 * as PrioritizedController.plan hands it to run_optimizer (PrioritizedController.m:297-341). */
typedef struct {
    double x0, y0, yaw0;  /* iter.x0(1, 1:3) */
    int32_t trim0;        /* iter.trim_indices (1-based) */
    int32_t n_left;       /* columns of predicted_lanelet_boundary{1,1} (0 = empty, circle scenario) */
    int32_t n_right;      /* columns of predicted_lanelet_boundary{1,2} */
    int32_t _pad;
    const double* ref_x;  /* [Hp] iter.reference_trajectory_points(1, :, 1) */
    const double* ref_y;  /* [Hp] iter.reference_trajectory_points(1, :, 2) */
    const double* v_ref;  /* [Hp] iter.v_ref(1, :) */
    const double* left_x; /* left boundary polyline */
    const double* left_y;
    const double* right_x;
    const double* right_y;
    pdmpc_polygon_set obstacles;          /* iter.obstacles: n_s polygons */
    pdmpc_polygon_set dynamic_obstacles;  /* iter.dynamic_obstacle_area: n_d x Hp cell, polygon index = i*Hp + (k-1) */
    pdmpc_polygon_set hdv_reachable_sets; /* iter.hdv_reachable_sets(adjacent, :): n_h x Hp, same indexing (InterX only) */
} pdmpc_vehicle_in;

/* ControlResultsInfo for one vehicle (hlc/controller/common/ControlResultsInfo.m:5-17) in the
 * fixed-stride form create_control_results_info_from_mex expects (OptimizerInterface.m:63-101). */
typedef struct {
    int32_t status;      /* PDMPC_OK / PDMPC_EXHAUSTED / PDMPC_ARENA_OVERFLOW */
    int32_t n_expanded;  /* info.n_expanded == tree size (GraphSearch.m:58,89) */
    int32_t n_popped;    /* nodes taken from the open list, incl. rejected ones */
    int32_t n_hp;        /* Hp this record was written for */
    int32_t tree_path[PDMPC_HP_MAX + 1];   /* info.tree_path, 1-based node ids (GraphSearch.m:84) */
    int32_t predicted_trims[PDMPC_HP_MAX]; /* info.predicted_trims (GraphSearch.m:86) */
    int32_t shape_cols[PDMPC_HP_MAX];      /* columns of info.shapes{1,k} */
    int32_t _pad;
    double y_predicted[PDMPC_HP_MAX][3];   /* info.y_predicted(:, k, 1) = [x; y; yaw] (return_path_to.m:14-23); NaN if exhausted */
    double shapes[PDMPC_HP_MAX][2][PDMPC_VMAX]; /* info.shapes{1,k} (return_path_area.m:1-8) */
    double path_nodes[PDMPC_HP_MAX + 1][8]; /* rows in NodeInfo order x,y,yaw,trim,g,h,k,exactEval (NodeInfo.m:4-13) */
} pdmpc_vehicle_out;

/* counters of the last pdmpc_plan_* call, summed over the batch (feeds the roofline formula) */
typedef struct {
    int64_t n_vehicles;
    int64_t nodes_popped;
    int64_t nodes_generated;  /* children created (tree size minus roots) */
    int64_t obstacle_columns; /* sum over plans and steps of the obstacle-soup columns + boundary columns */
    int64_t algorithmic_bytes;/* SURVEY.md 8(d) formula evaluated for the call */
    double kernel_ms;         /* duration of the search kernel measured with HIP events on the launch stream */
    int64_t lds_bytes;        /* dynamic LDS per workgroup used by the launch */
    int64_t lds_nodes;        /* tree nodes resident in LDS per vehicle */
    int64_t n_launches;       /* kernel launches since the last pdmpc_pack_* (kernel_ms is their sum) */
    int64_t queue_fallbacks;  /* searches since pdmpc_create / pdmpc_reset_stats that met equal keys where the pop order decides and ended
                                 on the replay of their tree through the libstdc++-faithful binary heap (priority_queue_interface_mex.cpp:19-31) */
    int64_t speculation_arrivals; /* verifications of a running search's tree against areas of predecessors that finished meanwhile (same period) */
    int64_t edge_checks;       /* eval_edge_exact evaluations since pdmpc_create / pdmpc_reset_stats (incl. the ones the reference never makes) */
    int64_t segment_pair_tests;/* (area segment, obstacle segment) pairs those checks stand for: sum of (V-1)(M-1) per soup,
                                  InterX.m:63-76 (InterX checker only) */
    int64_t kernel;            /* kernel of the last launch: 2 the graph search (bulk-synchronous rounds), 3 the sampled optimizer */
    int64_t nodes_processed;   /* nodes whose edge was evaluated (same period; the reference pops nodes_popped of them, the rest is what
                                  the parallel rounds overshoot) */
    int64_t rounds;            /* rounds (select a batch of the smallest open keys, process it) */
    int64_t shared_rounds;     /* ... of which shared with helper workgroups (CUs the launch left idle; same period) */
    int64_t helper_checked;    /* ... and the nodes whose edges those helpers evaluated (part of nodes_processed) */
    int64_t safe_replans;      /* calls since pdmpc_create that were planned a second time in resident slices because a search of an
                                  oversubscribed launch gave up waiting for a predecessor (see pdmpc_set_safe_launch) */
    int64_t bad_status_plans;  /* plans since pdmpc_create / pdmpc_reset_stats whose record carries neither PDMPC_OK nor PDMPC_EXHAUSTED
                                  (arena overflow, predecessor time-out), counted on the device: also covers launches nobody fetched */
} pdmpc_stats;

/* Tuning.  The graph search's knobs (round sizes, helper workgroups, tiles, the open set's lists) and its A/B and test switches have
 * measured defaults; ONE environment variable, read once in pdmpc_create, overrides them for benchmarking and tests:
 *     PDMPC_TUNING="key=value,key=value,..."
 * keys: round0 round ramp ready share_min own_div tile mid_min mid_fill (rounds and lists), tentative fast_arrival speculate helpers
 * helpers_oversub helpers_first seat_nodes waves (A/B switches), force_tie reverse_dispatch spin_limit (testing), debug_tail debug_lds debug_host debug_progress
 * (diagnostics); csrc/api.cpp: struct Tuning documents each.  No setting changes a result; an unknown key fails pdmpc_create. */

/* ---- life cycle (replaces GraphSearch() construction in OptimizerInterface.get_optimizer, :26-27,
 *      and the MEX instance table of priority_queue_interface_mex.cpp:48-53,111) ---- */
int pdmpc_create(const pdmpc_config* config, pdmpc_handle** out_handle);
int pdmpc_destroy(pdmpc_handle* handle);

/* the configuration the handle runs with (defaults filled in, max_nodes = current arena size) and whether the MPA is uploaded */
int pdmpc_get_config(pdmpc_handle* handle, pdmpc_config* config, int32_t* mpa_uploaded);

/* uploads mpa.maneuvers / mpa.transition_matrix_single once (MotionPrimitiveAutomaton.m:5-9) */
int pdmpc_upload_mpa(pdmpc_handle* handle, const pdmpc_mpa* mpa);

/* Plans n independent vehicles (one computation level).  Blocking.  With n == 1 this is
 * GraphSearch.run_optimizer (GraphSearch.m:14-17). */
int pdmpc_plan_batch(pdmpc_handle* handle, int32_t n_vehicles, const pdmpc_vehicle_in* in,
                     pdmpc_vehicle_out* out);

/* Arena growth.  The reference's search tree is unbounded (Tree.m:54-70: add_nodes appends); this backend keeps the tree of
 * every vehicle in a fixed HBM arena of config.max_nodes nodes.  pdmpc_plan_batch / pdmpc_plan_step therefore re-plan a
 * call in which some search outgrew its arena with arenas twice as large (the searches are deterministic: vehicles that
 * fitted produce the same records again), repeatedly, until every search fits.  PDMPC_ARENA_OVERFLOW is only ever
 * returned when the limit set here is reached (max_nodes_limit == current size: growth off; 0: up to what HBM holds). */
int pdmpc_set_arena_limit(pdmpc_handle* handle, int32_t max_nodes_limit);
/* explicit growth for the device-resident path below (launch / fetch do not grow by themselves) */
int pdmpc_grow_arena(pdmpc_handle* handle, int32_t max_nodes);
/* current per-vehicle arena size and how often a call had to be re-planned with larger arenas since pdmpc_create */
int pdmpc_arena_nodes(pdmpc_handle* handle, int32_t* max_nodes, int64_t* regrows);

/* A whole time step in one call (PrioritizedSequentialController.controller, :77-94): pdmpc_pack_step + launch + fetch,
 * with arena growth as above.  Arguments as for pdmpc_pack_step (below). */
int pdmpc_plan_step(pdmpc_handle* handle, int32_t n_vehicles, const pdmpc_vehicle_in* in, const int32_t* pred_offset,
                    const int32_t* pred_index, const pdmpc_polygon_set* fallback_shapes, pdmpc_vehicle_out* out);

/* The same step the way an UNMODIFIED reference controller drives this backend: one pdmpc_plan_batch of ONE vehicle per
 * run_optimizer call (PrioritizedController.m:335-341) in kahn order (PrioritizedSequentialController.m:77-94), the predecessors'
 * solved areas handed over on the host as dynamic obstacles (PrioritizedController.m:476-491).  Arguments and records as for
 * pdmpc_plan_step (slots in level order).  The slow way to use the library: bench.py reports it as value_run_optimizer_literal. */
int pdmpc_plan_step_literal(pdmpc_handle* handle, int32_t n_vehicles, const pdmpc_vehicle_in* in, const int32_t* pred_offset, const int32_t* pred_index,
                            const pdmpc_polygon_set* fallback_shapes, pdmpc_vehicle_out* out);

/* ---- device-resident path used by the batched host driver and bench.py ----
 * pdmpc_pack_batch flattens host inputs into the handle's device blob (H2D copy, async on the
 * handle's stream); pdmpc_launch_packed runs the search kernel on whatever is packed (no copies);
 * pdmpc_fetch_results copies the result records back.  pdmpc_plan_batch == pack + launch + fetch.
 * A pack writes the batch straight into the selected bank's staging memory: one that fails (invalid input, out of memory) leaves that
 * bank EMPTY — the batch that was in it is gone, a launch on it returns PDMPC_ERR_INVALID until the next successful pack. */
int pdmpc_pack_batch(pdmpc_handle* handle, int32_t n_vehicles, const pdmpc_vehicle_in* in);
/* pdmpc_plan_step for a caller that keeps only a few of the batch's plans (the explorative step: the choice among the prioritizations
 * rests on the cost-to-come of every vehicle's final node, PrioritizedExplorativeController.m:94-112, and only the chosen plans are
 * applied): plans the step like pdmpc_plan_step but copies back status[v] and final_cost[v] = path_nodes[Hp][4] only (12 bytes per
 * vehicle instead of 2.9 KB); the records stay on the device until the next launch and pdmpc_fetch_records_at reads the ones wanted
 * (vehicles: indices into the step as it was handed over). */
int pdmpc_plan_step_lean(pdmpc_handle* handle, int32_t n_vehicles, const pdmpc_vehicle_in* in, const int32_t* pred_offset, const int32_t* pred_index,
                         const pdmpc_polygon_set* fallback_shapes, int32_t* status, double* final_cost);
int pdmpc_fetch_records_at(pdmpc_handle* handle, int32_t count, const int32_t* vehicles, pdmpc_vehicle_out* out);
/* host wall-clock microseconds of the last pdmpc_plan_batch / pdmpc_plan_step on this handle: [0] pack (flatten + queue the
 * host-to-device copy), [1] enqueue the launch, [2] wait for the kernel + copy the records back */
int pdmpc_last_call_timing(pdmpc_handle* handle, double* us3);
int pdmpc_launch_packed(pdmpc_handle* handle);
/* launches only slots [first, first + count) of the packed batch: one computation level, or one GPU's shard of it */
int pdmpc_launch_range(pdmpc_handle* handle, int32_t first, int32_t count);
int pdmpc_fetch_results(pdmpc_handle* handle, int32_t n_vehicles, pdmpc_vehicle_out* out);
int pdmpc_synchronize(pdmpc_handle* handle);
/* Forward progress of launches with more searches than resident workgroups (256 CUs x 1 workgroup; x 2 with the compact kernel).
 * A search spins for predecessors of the same launch.  Every predecessor sits in a lower slot (pdmpc_pack_step puts a batch into a
 * topological order of its coupling DAG — level order, or priority order with pdmpc_set_step_weights — unless it is in one already),
 * so as long as the hardware hands workgroups out IN INDEX ORDER whoever a search waits for was dispatched before it and the launch
 * cannot stall.  THAT ORDER IS AN ASSUMPTION about the dispatcher (observed on gfx950 / ROCm 7, not a documented guarantee).
 * Should a launch stall anyway, the kernel's watchdog (spin limit) ends the waiting searches with status PDMPC_ERR_HIP.
 * Which entry points recover by themselves:
 *   pdmpc_plan_batch, pdmpc_plan_step, pdmpc_plan_step_literal, pdmpc_controller_step / _run / _explore_*  — plan the call again in
 *     slices that are resident as a whole (predecessors in the same or an earlier slice: no assumption left), counted in
 *     pdmpc_stats.safe_replans;
 *   pdmpc_launch_packed, pdmpc_launch_range, pdmpc_group_launch, pdmpc_group_plan_step (the resident paths)  — do NOT: the records
 *     carry the error status (pdmpc_stats.bad_status_plans counts them on the device) and the caller decides.
 * pdmpc_set_safe_launch(on != 0) makes every launch of this handle, resident paths included, use the resident slices from the start.
 * Tested against the adversarial order with PDMPC_TUNING=reverse_dispatch=1 (tests/test_gpu_step.py). */
int pdmpc_set_safe_launch(pdmpc_handle* handle, int32_t on);
/* n_handles handles of this process launch on the handle's device side by side (pdmpc_group_create_ex sets it for logical ranks that
 * share a GPU): the helper workgroups of a launch are sized for 1 / n_handles of the device's idle CUs and none of them is dispatched
 * in front of the searches — four launches with half a chip's worth of helpers in front of each would fill the device with helpers
 * that wait for searches which cannot start. */
int pdmpc_set_device_share(pdmpc_handle* handle, int32_t n_handles);
/* starts a new time step for launches issued with pdmpc_launch_range: results of earlier steps stop
 * satisfying predecessor waits (pdmpc_launch_packed does this implicitly) */
int pdmpc_begin_step(pdmpc_handle* handle);
/* several packed batches can stay resident in HBM side by side; pack/launch/fetch act on the selected
 * bank (default 0).  bench.py keeps one bank per recorded time step so the timed region has no copies. */
int pdmpc_select_bank(pdmpc_handle* handle, int32_t bank);
/* forget the kernel timings accumulated so far (pdmpc_get_last_stats sums launches since the last reset/pack) */
int pdmpc_reset_stats(pdmpc_handle* handle);

/* Step-level planning (PrioritizedSequentialController.controller, :77-94): all vehicles of a
 * time step in ONE launch.  pred_offset/pred_index (CSR over vehicles, 0-based vehicle indices of
 * this batch) list each vehicle's sequential predecessors; their solved info.shapes(1,:) are appended
 * on the device to the vehicle's dynamic obstacles (PrioritizedController.m:476-491) before it plans.
 * fallback_shapes (may be NULL) gives, per vehicle, the Hp areas published when its search is
 * exhausted (PrioritizedController.m:568-616,678-718): [n][Hp] pdmpc_polygon_set-style via offsets. */
/* Expected work per vehicle of the NEXT packed step (n = its vehicle count, the caller's vehicle order; e.g. n_popped of the previous
 * time step), consumed by the next pdmpc_pack_step / pdmpc_plan_step / pdmpc_pack_batch.  With it, a launch of the whole step hands its
 * searches out by priority — the largest expected work among a vehicle and its descendants in the coupling DAG, descending — instead
 * of slot order: still a topological order (whoever a search waits for was dispatched before it), but in a launch of more searches
 * than CUs the heavy searches of late computation levels start with the launch instead of behind the finished searches that wait for
 * their predecessors.  Slots, records and results are the same bit for bit.  Optional: without it searches go out in slot order. */
int pdmpc_set_step_weights(pdmpc_handle* handle, int32_t n_vehicles, const double* weights);
int pdmpc_pack_step(pdmpc_handle* handle, int32_t n_vehicles, const pdmpc_vehicle_in* in,
                    const int32_t* pred_offset, const int32_t* pred_index,
                    const pdmpc_polygon_set* fallback_shapes);

/* raw device pointer + byte size of the packed result records (pdmpc_vehicle_out[n]) for exchange
 * between GPUs (RCCL all-gather of solved areas).  This entry point, pdmpc_import_results and
 * pdmpc_export_results(_async) address SLOTS: they return PDMPC_ERR_INVALID for a batch that pdmpc_pack_step had to put into level
 * order itself (slots then are not the caller's vehicles); pack in level order -- predecessors in lower slots -- to use them. */
int pdmpc_result_device_buffer(pdmpc_handle* handle, void** dev_ptr, size_t* nbytes);
/* make result records produced elsewhere (e.g. gathered from another GPU into dev_ptr) visible as
 * predecessor outputs: copies n records into slots [first, first+n) of the handle's result buffer */
int pdmpc_import_results(pdmpc_handle* handle, int32_t first, int32_t n, const void* dev_records);
/* the other direction: copies the records of slots [first, first+n) into the caller's DEVICE buffer (the send
 * buffer of the all-gather) and waits for the copy, so the buffer can be handed to another stream */
int pdmpc_export_results(pdmpc_handle* handle, int32_t first, int32_t n, void* dev_records);

/* the same without the wait: the copy is only ordered on the handle's stream (see pdmpc_stream) */
int pdmpc_export_results_async(pdmpc_handle* handle, int32_t first, int32_t n, void* dev_records);
/* The HIP stream (hipStream_t) every launch and copy of this handle is enqueued on.  A caller that exchanges records
 * between GPUs enqueues its collective on THIS stream (e.g. torch.cuda.ExternalStream around it), so export -> all-gather
 * -> import -> next launch are ordered by the stream itself, with no host synchronisation and no cross-stream race. */
int pdmpc_stream(pdmpc_handle* handle, void** hip_stream);

int pdmpc_get_last_stats(pdmpc_handle* handle, pdmpc_stats* stats);

/* ---- debug / parity instrumentation (no reference counterpart: the reference keeps the whole
 *      Tree in info.tree, Tree.m:3-13; these calls read it back from HBM) ---- */
/* node ids popped by vehicle v in order; returns count in *n */
int pdmpc_debug_pop_trace(pdmpc_handle* handle, int32_t vehicle, int32_t capacity, int32_t* ids, int32_t* n);
/* the search tree of vehicle v: arrays of length capacity, *n receives tree size */
int pdmpc_debug_tree(pdmpc_handle* handle, int32_t vehicle, int32_t capacity, double* x, double* y,
                     double* yaw, double* g, double* h, int32_t* trim, int32_t* k, int32_t* parent,
                     int32_t* n);

/* The collision primitives on given polygons, n_cases at once (one wavefront each, the device functions the search kernels
 * inline): mode 0 = InterX(a, b) (graph_search/InterX.m:48-103, isReturnPoints = false; b may hold NaN separators),
 * mode 1 = intersect_sat(a, b) (graph_search/intersect_sat.m:1-42), mode 2 = intersect_lanelet_boundary(a, [left, NaN, right, NaN])
 * (optimizer/common/intersect_lanelet_boundary.m:1-56).  Case c: a = columns a_off[c] .. a_off[c+1]-1 of (a_x, a_y), at most
 * PDMPC_VMAX; b likewise, at most 1024 columns.  hit[c] = 1 if the reference function returns true. */
int pdmpc_debug_edge_check(pdmpc_handle* handle, int32_t mode, int32_t n_cases, const int32_t* a_off, const double* a_x, const double* a_y,
                           const int32_t* b_off, const double* b_x, const double* b_y, int32_t* hit);

/* the arena of vehicle v as the kernel left it (the frontier kernel's own creation order, incl. nodes the reference never
 * creates), with every node's open-list key and validity byte (0 never evaluated, 1 collision-free, 2 colliding) */
int pdmpc_debug_raw_tree(pdmpc_handle* handle, int32_t vehicle, int32_t capacity, double* x, double* y, double* yaw, double* g,
                         double* h, int32_t* trim, int32_t* k, int32_t* parent, double* key, uint8_t* validity, int32_t* n);

/* the device's work counters since pdmpc_create / pdmpc_reset_stats, raw: [0..6] as pdmpc_stats reports them, [8..12] with
 * PDMPC_TUNING=debug_tail=1 the helper workgroups' time in 100 MHz ticks (idle, claim -> soup, records, checks, verdicts + report), [13] tiles */
int pdmpc_debug_counters(pdmpc_handle* handle, uint64_t* out16);

/* live counters of a running frontier launch (needs PDMPC_DEBUG_PROGRESS=1 in the environment; callable from another thread
 * while pdmpc_plan_* blocks): rounds, nodes processed, tree size, near / far entries, flags, best candidate, stage */
int pdmpc_debug_progress(pdmpc_handle* handle, int32_t vehicle, uint32_t* words16);

/* drives the device open list with a command script (op[i] == 0: push (id[i], key[i]); op[i] == 1: pop) the way the
 * reference drives priority_queue_interface_mex (PUSH / POP, .cpp:62-99); popped[] receives the popped ids (-1 on an empty
 * queue).  lds_entries = how many heap entries live in LDS (the rest spills to HBM).  Used by the heap-order unit test. */
int pdmpc_debug_heap_script(pdmpc_handle* handle, int32_t n, const int32_t* op, const int32_t* id, const double* key,
                            int32_t lds_entries, int32_t* popped, int32_t* n_popped, double* cycles_per_pop,
                            double* cycles_per_push);

/* ---- the sampled optimizer (replaces MonteCarloTreeSearch.run_optimizer, graph_search/MonteCarloTreeSearch.m:30-35,
 *      selected by OptimizerType.MatlabSampled in OptimizerInterface.get_optimizer, OptimizerInterface.m:29-31) ----
 * One computation level like pdmpc_plan_batch.  seeds[i] = time_step + vehicle_index of vehicle i, the seed of its
 * mt19937ar stream (:32).  Records: status OK / EXHAUSTED, n_expanded = expansions (:209), tree_path = node ids of the
 * chosen descent, path_nodes rows with g = -1 except the cost of the chosen node, h = -1, k = 1..Hp+1 (:223-244). */
int pdmpc_plan_batch_sampled(pdmpc_handle* handle, int32_t n_vehicles, const pdmpc_vehicle_in* in, const uint32_t* seeds,
                             pdmpc_vehicle_out* out);

/* ---- the caller's side of the boundary, natively (csrc/step_controller.cpp) ----
 * One MPC time step of the prioritized sequential controller around pdmpc_plan_step, without any interpreter in the loop:
 * traffic info, coupling, priorities, grouping, computation levels, obstacle assembly, ONE launch, exhaustion handling,
 * fallbacks, plant update (HighLevelController.main_control_loop, hlc/controller/HighLevelController.m:334-373;
 * PrioritizedSequentialController.controller, hlc/controller/prioritized/PrioritizedSequentialController.m:77-94;
 * PrioritizedController.m:297-324,375-389,449-718; plant/Simulation.m:86-100).  p-dmpc_amd/pdmpc/controller.py is the same
 * logic in Python (it also offers the random and FCA prioritizers); the two build bit-identical step problems. */
enum { PDMPC_COUPLING_FULL = 0, PDMPC_COUPLING_DISTANCE = 1, PDMPC_COUPLING_NONE = 2 };          /* Coupler.m:31-32, DistanceCoupler.m:15-50 */
enum { PDMPC_PRIORITY_CONSTANT = 0, PDMPC_PRIORITY_COLORING = 1 };                             /* ConstantPrioritizer.m, ColoringPrioritizer.m */
enum { PDMPC_WEIGHT_DISTANCE = 0, PDMPC_WEIGHT_CONSTANT = 1 };                                 /* weight/DistanceWeigher.m, ConstantWeigher.m */
enum { PDMPC_SUCCESSOR_NONE = 0, PDMPC_SUCCESSOR_AREA_OF_STANDSTILL = 1, PDMPC_SUCCESSOR_AREA_OF_PREVIOUS_TRAJECTORY = 2 }; /* ConstraintFromSuccessor.m */

typedef struct {
    int32_t Hp;                        /* options.Hp */
    int32_t coupling;                  /* PDMPC_COUPLING_* */
    int32_t priority_strategy;         /* PDMPC_PRIORITY_* */
    int32_t weight_strategy;           /* PDMPC_WEIGHT_* (only matters when the coupling DAG is deeper than max_num_CLs) */
    int32_t max_num_CLs;               /* options.max_num_CLs (Config.m:28) */
    int32_t constraint_from_successor; /* PDMPC_SUCCESSOR_* (Config.m:37) */
    double dt_seconds;                 /* options.dt_seconds */
    double offset;                     /* options.offset (Config.m:49) */
    double vehicle_length, vehicle_width; /* scenarios/Vehicle.m:10-11 */
} pdmpc_controller_config;

/* The scenario fields the controller reads (scenarios/Scenario.m, Vehicle.m) plus the trims' speed / steering
 * (MotionPrimitiveAutomaton.trims) and the per-lanelet boundaries of the map (RoadDataCommonRoad.get_lanelet_boundary). */
typedef struct {
    int32_t n_vehicles;
    const double *x_start, *y_start, *yaw_start, *reference_speed; /* [n_vehicles] */
    const int32_t* path_offset;      /* [n_vehicles + 1]: reference path of vehicle v = points path_offset[v] .. path_offset[v+1]-1 */
    const double *path_x, *path_y;
    const int32_t* lanelets_offset;  /* [n_vehicles + 1] or NULL (no lanelets: circle scenario) */
    const int32_t* lanelets_index;   /* 1-based lanelet ids along the vehicle's loop (Vehicle.lanelets_index) */
    const int32_t* points_index;     /* same offsets: 1-based index of the last path point of each of those lanelets */
    const int32_t* is_loop;          /* [n_vehicles] or NULL */
    const double *tile_dx, *tile_dy; /* [n_vehicles] or NULL: translation of the vehicle's copy of the map */
    int32_t n_lanelets;
    const int32_t *left_offset, *right_offset; /* [n_lanelets + 1] */
    const double *left_x, *left_y, *right_x, *right_y;
    pdmpc_polygon_set obstacles;     /* scenario.obstacles */
    int32_t n_trims;
    const double *trim_speed, *trim_steering; /* [n_trims] */
} pdmpc_scenario;

typedef struct pdmpc_controller pdmpc_controller;

/* handle may be NULL for a controller that only builds step problems / applies given records (tests without a GPU) */
int pdmpc_controller_create(pdmpc_handle* handle, const pdmpc_controller_config* cfg, const pdmpc_scenario* scenario, pdmpc_controller** out);
int pdmpc_controller_destroy(pdmpc_controller* c);
/* one whole time step: build the step problem, plan it with ONE launch (pdmpc_plan_step), apply the result */
int pdmpc_controller_step(pdmpc_controller* c);
/* n_steps closed-loop time steps in one call; ms[i] (may be NULL) = wall-clock milliseconds of step i */
int pdmpc_controller_run(pdmpc_controller* c, int32_t n_steps, double* ms);
/* the two host halves on their own: build_step advances the time step counter and leaves the problem readable with
 * pdmpc_controller_problem; apply takes the records of that problem in slot order */
int pdmpc_controller_build_step(pdmpc_controller* c);
int pdmpc_controller_apply(pdmpc_controller* c, const pdmpc_vehicle_out* records);
/* the problem of the last build_step exactly as pdmpc_plan_step receives it (pointers stay valid until the next build_step);
 * order[s] = vehicle (0-based) in slot s, levels[v] = computation level (1-based) of vehicle v */
int pdmpc_controller_problem(pdmpc_controller* c, int32_t* n, const pdmpc_vehicle_in** in, const int32_t** pred_offset, const int32_t** pred_index,
                             const pdmpc_polygon_set** fallback, const int32_t** order, const int32_t** levels);
/* plant state after the last apply (PlantMeasurement), per vehicle; any pointer may be NULL */
int pdmpc_controller_state(pdmpc_controller* c, double* x, double* y, double* yaw, double* speed, double* steering, int32_t* needs_fallback,
                           int32_t* time_step);
const pdmpc_vehicle_out* pdmpc_controller_records(pdmpc_controller* c); /* records of the last pdmpc_controller_step, slot order */
/* The explorative controller's permutations of the computation levels (PrioritizedExplorativeController.m:241-309): n_perm x
 * n_levels, row-major, drawn from RandStream("mt19937ar", Seed = seed) / randi as the reference draws them (:249, :283-286);
 * rows beyond n_levels (the reference stops there) are shuffles from the same stream.  Twin of pdmpc.explorative. */
int pdmpc_exploration_permutations(int32_t n_levels, int32_t n_perm, uint32_t seed, int32_t* out);
/* The explorative time step (PrioritizedExplorativeController.m:25-176): the step's traffic state under n_perm
 * prioritizations — instance 0 the controller's own, instance p the computation levels permuted by row p of the table above —
 * flattened into one batch whose slots are ordered by (level, instance).  explore_build advances the time step like build_step and
 * leaves the batch readable with explore_problem (instance / vehicle / level per slot); explore_choose takes the batch's records:
 * per weakly connected sub-graph the instance with the smallest summed cost-to-come of the final nodes after round(., 8)
 * (:94-176), chosen[v] = the instance vehicle v goes on with, cost = n_perm x n_graphs; explore_step = build + ONE launch +
 * choose + apply of the chosen plans (seed = time step, :249); explore_run = n_steps of them, ms[i] = wall-clock of step i. */
int pdmpc_controller_explore_build(pdmpc_controller* c, int32_t n_perm, uint32_t seed);
int pdmpc_controller_explore_problem(pdmpc_controller* c, int32_t* n_slots, const pdmpc_vehicle_in** in, const int32_t** pred_offset, const int32_t** pred_index,
                                     const pdmpc_polygon_set** fallback, const int32_t** instance, const int32_t** vehicle, const int32_t** level);
int pdmpc_controller_explore_choose(pdmpc_controller* c, const pdmpc_vehicle_out* records, int32_t* chosen, int32_t* n_graphs, double* cost);
int pdmpc_controller_explore_step(pdmpc_controller* c, int32_t n_perm);
int pdmpc_controller_explore_run(pdmpc_controller* c, int32_t n_perm, int32_t n_steps, double* ms);
/* measurement only: on != 0 makes the explorative step build, plan and choose as always but APPLY the plans of the controller's own
 * prioritization, so that the traffic follows pdmpc_controller_step's closed loop (bench.py: the host-inclusive rate on the same steps
 * as the resident replay) */
int pdmpc_controller_explore_follow_own(pdmpc_controller* c, int32_t on);
int pdmpc_controller_explore_result(pdmpc_controller* c, int32_t* chosen, int32_t* n_graphs, const double** cost, const pdmpc_vehicle_out** records);
/* host wall-clock milliseconds of the last pdmpc_controller_step / pdmpc_controller_explore_step, by part: [0] build the step problem(s)
 * on the host, [1] pack, [2] enqueue, [3] wait + read-back, [4] choice among the prioritizations (explorative step), [5] apply */
int pdmpc_controller_last_timing(pdmpc_controller* c, double* ms6);
/* ... and summed over the steps since the last call with reset != 0 (n_steps: how many) */
int pdmpc_controller_timing_sum(pdmpc_controller* c, double* ms6, int64_t* n_steps, int32_t reset);
const char* pdmpc_controller_last_error(void);

/* ---- several GPUs behind the same boundary (csrc/group.cpp) ----
 * The reference's vehicles exchange their solved areas after every computation level: each publishes a Predictions message that every
 * coupled vehicle reads (hlc/communication/PredictionsCommunication.m:34-63, sent from PrioritizedController.publish_predictions,
 * PrioritizedController.m:356-365, read back at :476-491).  A group is that exchange between the GPUs of ONE process: one handle and
 * one stream per device, bound by an RCCL communicator (ncclCommInitAll; librccl is loaded when the first group is created, the
 * single-GPU entry points do not need it).  pdmpc_group_plan_step plans a time step over the group:
 *   PDMPC_SHARD_COMPONENTS  the weakly connected components of the step's coupling graph exchange nothing within the step: every
 *                           device takes whole components (longest processing time first on `weights`), plans them with ONE
 *                           launch, and ONE all-gather of the result records ends the step;
 *   PDMPC_SHARD_LEVELS      every computation level is block-partitioned over the devices; after each level one all-gather of the
 *                           level's records (2.9 KB per vehicle) on the handles' streams, imported on every device as predecessor
 *                           areas of the next level -- the literal image of the per-level Predictions broadcast;
 *   PDMPC_SHARD_AUTO        whole components, except a component that outweighs the mean load per device: that one by levels over
 *                           all devices, the others whole.
 * The prioritization instances of an explorative step (pdmpc_controller_explore_*) are components of their batch: COMPONENTS deals
 * them out.  Arguments as for pdmpc_plan_step (any slot order; records come back in the caller's order); weights (may be NULL: 1
 * each) = expected work per vehicle, e.g. n_popped of the previous step.  Results are those of the single launch, bit for bit. */
enum { PDMPC_SHARD_AUTO = 0, PDMPC_SHARD_COMPONENTS = 1, PDMPC_SHARD_LEVELS = 2 };
typedef struct pdmpc_group pdmpc_group;
/* devices: n_devices HIP device ordinals (NULL: 0 .. n_devices - 1); config as for pdmpc_create (config.device is ignored).
 * The exchange between the ranks sits behind a function table (csrc/group.cpp: struct Collective):
 *   PDMPC_COLLECTIVE_RCCL  ncclAllGather on the handles' streams, one communicator per device (ncclCommInitAll) — distinct devices;
 *   PDMPC_COLLECTIVE_COPY  the same all-gather as peer copies ordered by HIP events on the handles' streams (no library, no host
 *                          wait).  A device may then be listed more than once: such ranks are LOGICAL ranks — a handle, a stream
 *                          and arenas of their own on a shared GPU — and the whole multi-rank protocol (slot remapping, block
 *                          partition of a level, import of the other ranks' blocks) runs on a 1-GPU box;
 *   PDMPC_COLLECTIVE_AUTO  RCCL for distinct devices (PDMPC_GROUP_COLLECTIVE=copy in the environment: peer copies), peer copies
 *                          when a device is listed twice.
 * pdmpc_group_create = pdmpc_group_create_ex(..., PDMPC_COLLECTIVE_AUTO, ...).  Every pdmpc_group_* and pdmpc_* entry point leaves
 * the calling thread's current HIP device as it found it. */
enum { PDMPC_COLLECTIVE_AUTO = 0, PDMPC_COLLECTIVE_RCCL = 1, PDMPC_COLLECTIVE_COPY = 2 };
int pdmpc_group_create(const pdmpc_config* config, int32_t n_devices, const int32_t* devices, pdmpc_group** out_group);
int pdmpc_group_create_ex(const pdmpc_config* config, int32_t n_devices, const int32_t* devices, int32_t collective, pdmpc_group** out_group);
int pdmpc_group_collective(pdmpc_group* group, int32_t* collective); /* which of the two the group uses (RCCL or COPY) */
int pdmpc_group_destroy(pdmpc_group* group);
int pdmpc_group_size(pdmpc_group* group, int32_t* n_devices);
int pdmpc_group_handle(pdmpc_group* group, int32_t rank, pdmpc_handle** handle); /* rank's handle (statistics, debug read-backs) */
/* The arenas of every device at least `max_nodes` nodes per vehicle (pdmpc_grow_arena on each handle).  pdmpc_group_plan_step grows them
 * by itself when a search overflows; a caller of the resident path (pack_step / launch / fetch), which does not plan again, sizes them
 * here — e.g. with what a single handle needed for the same steps — and reads the statuses. */
int pdmpc_group_grow_arena(pdmpc_group* group, int32_t max_nodes);
int pdmpc_group_upload_mpa(pdmpc_group* group, const pdmpc_mpa* mpa);
int pdmpc_group_plan_step(pdmpc_group* group, int32_t n_vehicles, const pdmpc_vehicle_in* in, const int32_t* pred_offset, const int32_t* pred_index,
                          const pdmpc_polygon_set* fallback_shapes, const double* weights, int32_t mode, pdmpc_vehicle_out* out);
/* The three parts of pdmpc_group_plan_step on their own, for steps that stay resident (bench.py's timed replay): pack partitions the
 * step and makes its sub-problems resident on the devices in group bank `bank` (0 .. 999; the handles' own banks 0 .. 2047 stay the
 * caller's), launch plans a packed bank (launches, all-gathers and imports enqueued on the handles' streams, ONE wait at the end; no
 * host-to-device copy), fetch copies the records of the bank launched last to the host (PDMPC_ERR_INVALID for any other bank: the
 * devices hold one step's gathered records).  plan_step = pack(0) + launch(0) + fetch(0). */
int pdmpc_group_pack_step(pdmpc_group* group, int32_t bank, int32_t n_vehicles, const pdmpc_vehicle_in* in, const int32_t* pred_offset, const int32_t* pred_index,
                          const pdmpc_polygon_set* fallback_shapes, const double* weights, int32_t mode);
int pdmpc_group_launch(pdmpc_group* group, int32_t bank);
int pdmpc_group_fetch(pdmpc_group* group, int32_t bank, int32_t n_vehicles, pdmpc_vehicle_out* out);
/* The partition pdmpc_group_plan_step uses, on its own (no GPU needed): rank_of[v] = device that plans vehicle v as part of a whole
 * component, or -1 if v belongs to the component that is planned by levels over all devices (then level_of[v] = its computation
 * level, 1-based, and block_rank[v] = the device of its block within that level; 0 / -1 for the others).  Twin of
 * pdmpc.distributed.partition_components / hybrid_partition / level_partition. */
int pdmpc_group_partition(int32_t n_vehicles, const int32_t* pred_offset, const int32_t* pred_index, const double* weights, int32_t n_devices, int32_t mode,
                          int32_t* rank_of, int32_t* level_of, int32_t* block_rank);
/* wall-clock milliseconds of the last pdmpc_group_plan_step and of its phases: [0] total, [1] partition + sub-problems, [2] pack,
 * [3] launches and collectives enqueued, [4] wait, [5] read-back */
int pdmpc_group_last_timing(pdmpc_group* group, double* ms6);

const char* pdmpc_last_error(void);
const char* pdmpc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PDMPC_H */
