// pdmpc_device.h — HBM data layout shared by the host packer (api.cpp) and the search kernel.
//
// Everything the kernel reads is a flat, pointer-free blob so one hipMemcpyAsync moves a whole batch:
//
//   DevVehicle veh[n]      fixed-stride record per vehicle (pose, reference, offsets into the pools)
//   double2    points[]    every literal polygon vertex of the batch, already in the "NaN-separated
//                          soup" order of vectorize_all_obstacles.m:36-62 (step-major per vehicle), so a
//                          workgroup stages its obstacles with one coalesced, fully pipelined copy
//   int32      pred[]      CSR list of predecessor slots (their solved areas are appended on the device)
//
// MPA tables (uploaded once):
//   uint64  succ_mask[Hp][n][n_words]   bit j of word w set <=> transition_matrix_single(i, 64w+j, k) ~= 0
//   int16   man_index[n][n]            index into the maneuver arrays or -1
//   DevManPose man_pose[T]             dx, dy, dyaw, n_cols
//   double2 man_area[T][3][VMAX]       area, area_without_offset, area_large_offset as (x, y) pairs
#ifndef PDMPC_DEVICE_H
#define PDMPC_DEVICE_H

#include <stdint.h>

#include "../../include/pdmpc.h"

#define PDMPC_WAVE 64
/* Twelve wavefronts per workgroup = three per SIMD.  With sixteen (128 VGPRs) the round-3 kernel spilled 44 VGPRs next to ~320 SGPRs held
 * in VGPR lanes, and in that regime hipcc 7.2 produced code that lost spilled values depending on unrelated source changes (DESIGN.md
 * section 3.4); at twelve no workload is slower.  The search kernels need 139-155 VGPRs today (profiles/r05_resource_usage.txt). */
#define PDMPC_MAX_WAVES 16     /* wavefronts per workgroup of the InterX search kernels (126 VGPRs: four per SIMD) */
#define PDMPC_MAX_WAVES_SAT 12 /* ... of the separating-axis kernel (150 VGPRs: three per SIMD) */
#define PDMPC_MAX_THREADS (PDMPC_WAVE * PDMPC_MAX_WAVES)
#define PDMPC_SH_WORDS 128 /* 32-bit LDS words shared by the waves of a workgroup (state, counters of the search) */

struct DevManPose {
    double dx, dy, dyaw;
    int32_t n_cols;
    int32_t pad;
};

struct DevVehicle {
    double x0, y0, yaw0;
    double ref_x[PDMPC_HP_MAX];
    double ref_y[PDMPC_HP_MAX];
    double v_ref[PDMPC_HP_MAX];
    int32_t trim0;                      // 1-based
    int32_t n_pred;                     // predecessors whose solved areas become dynamic obstacles
    int32_t pred_off;                   // into pred[]
    int32_t ll_off, ll_len;             // lanelet soup [left, NaN, right, NaN] in points[]
    int32_t lit_off[PDMPC_HP_MAX + 1];  // literal vehicle-obstacle soup of step k: points[lit_off[k] .. lit_off[k+1])
    int32_t hdv_off[PDMPC_HP_MAX + 1];  // literal HDV soup of step k
    int32_t fb_off[PDMPC_HP_MAX + 1];   // fallback areas (one polygon per step, no separators); fb_off[0] < 0: none
    int32_t pad;
};

// One search-tree node (Tree.m:3-13 row + what the check of this node's children needs), 64 bytes so a node is four
// 16-byte accesses and the children of an expansion form one contiguous, coalesced store.
struct NodeRec {
    double x, y, yaw, g;  // Tree.x/y/yaw/g
    double cs, sn;        // cos/sin(yaw), filled when the node is checked (expand_node.m:50-51); its children's
                          // edge checks read them (GraphSearch.m:155-156); a goal candidate keeps its path's largest key here
    double h;             // Tree.h
    uint32_t parent;      // Tree.parent (1-based id, 0 for the root)
    uint32_t packed;      // trim (10 bit, 1-based) | k << 10 (5 bit) | maneuver index << 15 (12 bit) | area columns << 27 (4 bit)
};
#define NODE_TRIM(p) ((int)((p) & 1023u))
#define NODE_K(p) ((int)(((p) >> 10) & 31u))
#define NODE_MAN(p) ((int)(((p) >> 15) & 4095u))
#define NODE_COLS(p) ((int)(((p) >> 27) & 15u))

// byte offsets of the regions of the dynamic LDS allocation (all multiples of 16; api.cpp: layout_bulk / layout_sampled)
struct LdsLayout {
    uint32_t mask, man_index, pose, area;  // MPA tables
    uint32_t ref;                          // ref_x[16], ref_y[16], dtv[16]
    uint32_t shape;                        // sampled optimizer: shape A [8], shape B [8] (double2) + the wave's tally
    uint32_t path;                         // uint32 path[HP_MAX+2], soup / HDV offsets, the shared words, literal soup lengths
    uint32_t soup;                         // double2[soup_cap]
    uint32_t cand;                         // phase B's chunk state (12 B per thread); sampled optimizer: candidate segments of one edge check
    uint32_t vstate;                       // uint8[NV]: validity bytes
    uint32_t expand;                       // d_traveled table: dcum[16][16] doubles (+ the sampled optimizer's expansion scratch)
    uint32_t tree16;                       // sampled optimizer: its tree (children, parent, trim as uint16)
    uint32_t nodes;                        // NodeRec[NL]
    uint32_t total;
    uint32_t bk_near_key, bk_near_id;      // double[BK_PER * threads], uint32[BK_PER * threads]: the LDS part of the open set (the heap of a replay)
    uint32_t bk_ready;                     // uint32[bk_ready_cap] nodes of the round + uint32[bk_ready_cap] their collision flags
    uint32_t bk_hist;                      // uint32[3072]: refill histogram [2048] | goal list [1024], expansion groups [1024], their children's offsets [1024]
    uint32_t bk_misc;                      // 2 KB: path tables of the best goal candidate, scan partials, chunk table, tick counters, the reference's ids along the path | the selection's 256-bin histogram
    uint32_t bk_pshape;                    // double2[Hp][VMAX] + uint32[HP_MAX] + uint64[HP_MAX]: the areas along the path of the record written last and their column counts (what an arrival is checked against first); per step the predecessors whose areas differ from the expected ones
};

// The regions of the graph search's layout whose sizes do not depend on the automaton or on the obstacles come first, at offsets that
// are compile-time constants (sized for PDMPC_MAX_WAVES wavefronts and the largest ready list): the kernel addresses them with
// immediates instead of holding two dozen LDS pointers in scalar registers it does not have (layout_bulk fills LdsLayout with the
// same numbers; the automaton's tables, the soup, the validity bytes and the nodes follow at run-time offsets).
#define PDMPC_LK_ALIGN16(x) (((x) + 15u) & ~15u)
// (W = wavefronts the layout is sized for, RC = entries of its ready list.  The product kernels: PDMPC_MAX_WAVES and 2048 — one
// workgroup per CU; bulk_kernel_compact.hip: 8 wavefronts, 512 ready entries and two near entries per thread, which with the automaton's
// areas left in L2 fits 80 KB — TWO workgroups per CU.  A translation unit sets PDMPC_LK_WAVES / PDMPC_LK_READY_CAP / PDMPC_BK_PER
// before it includes this header; the host lays out with pdmpc_lk_fixed().)
#ifndef PDMPC_LK_WAVES
#define PDMPC_LK_WAVES PDMPC_MAX_WAVES
#endif
#ifndef PDMPC_LK_READY_CAP
#define PDMPC_LK_READY_CAP 2048u
#endif
#define PDMPC_LK_COMPACT_WAVES 8
#define PDMPC_LK_COMPACT_READY_CAP 512u
#define PDMPC_LK_COMPACT_BK_PER 2
#define PDMPC_LKX_THREADS(W) ((uint32_t)(W) * PDMPC_WAVE)
#define PDMPC_LKX_REF 0u
#define PDMPC_LKX_SHAPE (PDMPC_LKX_REF + 3u * PDMPC_HP_MAX * 8u)
#define PDMPC_LKX_PATH(W) (PDMPC_LKX_SHAPE + (uint32_t)(W) * (2u * PDMPC_VMAX + 1u) * 16u)
#define PDMPC_LKX_CAND(W) (PDMPC_LKX_PATH(W) + PDMPC_LK_ALIGN16((PDMPC_HP_MAX + 2u) * 4u + 2u * (PDMPC_HP_MAX + 1u) * 4u + PDMPC_SH_WORDS * 4u + PDMPC_HP_MAX * 4u))
#define PDMPC_LKX_EXPAND(W) (PDMPC_LKX_CAND(W) + PDMPC_LK_ALIGN16(12u * PDMPC_LKX_THREADS(W)))
#define PDMPC_LKX_NEAR_KEY(W) (PDMPC_LKX_EXPAND(W) + (2u * PDMPC_HP_MAX * PDMPC_HP_MAX) * 8u + 16u * 16u)
#define PDMPC_LKX_NEAR_ID(W, P) (PDMPC_LKX_NEAR_KEY(W) + PDMPC_LK_ALIGN16((uint32_t)(P) * PDMPC_LKX_THREADS(W) * 8u))
#define PDMPC_LKX_READY(W, P) (PDMPC_LKX_NEAR_ID(W, P) + PDMPC_LK_ALIGN16((uint32_t)(P) * PDMPC_LKX_THREADS(W) * 4u))
#define PDMPC_LKX_HIST(W, RC, P) (PDMPC_LKX_READY(W, P) + (uint32_t)(RC) * 8u)
#define PDMPC_LKX_MISC(W, RC, P) (PDMPC_LKX_HIST(W, RC, P) + 3072u * 4u)
#define PDMPC_LKX_PSHAPE(W, RC, P) (PDMPC_LKX_MISC(W, RC, P) + 2048u)
#define PDMPC_LKX_FIXED_END(W, RC, P) (PDMPC_LKX_PSHAPE(W, RC, P) + PDMPC_LK_ALIGN16(PDMPC_HP_MAX * PDMPC_VMAX * 16u + PDMPC_HP_MAX * 4u + PDMPC_HP_MAX * 8u))
#ifndef PDMPC_BK_PER
#define PDMPC_BK_PER 4
#endif
/* entries of the LDS open list per thread (a selection pass holds them in registers) */
#define PDMPC_LK_THREADS PDMPC_LKX_THREADS(PDMPC_LK_WAVES)
#define PDMPC_LK_REF PDMPC_LKX_REF
#define PDMPC_LK_SHAPE PDMPC_LKX_SHAPE
#define PDMPC_LK_PATH PDMPC_LKX_PATH(PDMPC_LK_WAVES)
#define PDMPC_LK_CAND PDMPC_LKX_CAND(PDMPC_LK_WAVES)
#define PDMPC_LK_EXPAND PDMPC_LKX_EXPAND(PDMPC_LK_WAVES)
#define PDMPC_LK_NEAR_KEY PDMPC_LKX_NEAR_KEY(PDMPC_LK_WAVES)
#define PDMPC_LK_NEAR_ID PDMPC_LKX_NEAR_ID(PDMPC_LK_WAVES, PDMPC_BK_PER)
#define PDMPC_LK_READY PDMPC_LKX_READY(PDMPC_LK_WAVES, PDMPC_BK_PER)
#define PDMPC_LK_HIST PDMPC_LKX_HIST(PDMPC_LK_WAVES, PDMPC_LK_READY_CAP, PDMPC_BK_PER)
#define PDMPC_LK_MISC PDMPC_LKX_MISC(PDMPC_LK_WAVES, PDMPC_LK_READY_CAP, PDMPC_BK_PER)
#define PDMPC_LK_PSHAPE PDMPC_LKX_PSHAPE(PDMPC_LK_WAVES, PDMPC_LK_READY_CAP, PDMPC_BK_PER)
#define PDMPC_LK_FIXED_END PDMPC_LKX_FIXED_END(PDMPC_LK_WAVES, PDMPC_LK_READY_CAP, PDMPC_BK_PER)

struct NodeArena {  // HBM arrays, per-vehicle stride = max_nodes entries
    NodeRec* nodes;
    double* key;       // open-list key f = g + h of node i (GraphSearch.m:100-102)
    unsigned long long* link;  // parent | packed << 32 per node, eight nodes to a 64-byte line (what walks and phase B read of the tree)
    uint8_t* vstate;   // validity bytes of the nodes beyond the LDS-resident NV (and the read-back copy of those)
    double* far_key;   // open entries outside LDS: keys and nodes of `far`; the heap of a replay beyond its LDS part
    uint32_t* far_id;
    double* mid_key;   // ... and of `mid` (what a heavy search refills near from; far feeds it)
    uint32_t* mid_id;  //     (a replay leaves its pop sequence here)
    double* pb_key;    // phase B: a node's branch maximum
    uint32_t* pb_d;    // phase B: where a node's branch leaves the goal's path; a replay: the node's id in the reference's tree
    double* walk;      // per node, 16 bytes: what its branch to the goal candidate's path looks like (bk_classify_wave)
    uint32_t* child0;  // 1-based arena index of a node's first child (its children are consecutive, ascending trim), 0 while it has none: what the replay of a tied search descends by
};

// the fixed part of the graph search's layout for (W wavefronts, ready list of RC entries): what the kernel built with those two
// numbers addresses with immediates; returns the first free byte
static inline uint32_t pdmpc_lk_fixed(uint32_t W, uint32_t RC, uint32_t P, LdsLayout* L) {
    L->ref = PDMPC_LKX_REF;
    L->shape = PDMPC_LKX_SHAPE;
    L->path = PDMPC_LKX_PATH(W);
    L->cand = PDMPC_LKX_CAND(W);
    L->expand = PDMPC_LKX_EXPAND(W);
    L->bk_near_key = PDMPC_LKX_NEAR_KEY(W);
    L->bk_near_id = PDMPC_LKX_NEAR_ID(W, P);
    L->bk_ready = PDMPC_LKX_READY(W, P);
    L->bk_hist = PDMPC_LKX_HIST(W, RC, P);
    L->bk_misc = PDMPC_LKX_MISC(W, RC, P);
    L->bk_pshape = PDMPC_LKX_PSHAPE(W, RC, P);
    return PDMPC_LKX_FIXED_END(W, RC, P);
}

#define PDMPC_HELP_CAP 2048 /* entries of a round that can be shared (= the ready list's capacity) */
/* a search's board (64-bit words): what its seated helper workgroups read and write (bulk_search.hpp) */
#define PDMPC_HB_SEATS_MAX 64
#define PDMPC_HB_N 1      /* entries of the round that is being shared */
#define PDMPC_HB_MASK 2   /* predecessors whose areas the owner has in its soup */
#define PDMPC_HB_SEATS 3  /* launch id << 32 | seats taken (a helper takes one with a fetch-and-add; the owner resets the word when it starts) */
#define PDMPC_HB_WANT 4   /* == launch id: the search shares its rounds, helpers may take seats; 0 once it has ended */
#define PDMPC_HB_WEIGHT 5 /* nodes the search has processed so far: helpers go where the work per seat is largest */
#define PDMPC_HB_ASSIGN 8 /* [64] per seat: round << 40 | first entry << 20 | entries: the range that seat checks */
#define PDMPC_HB_DONE 72  /* [64] per seat: the last round whose range that seat has finished */
#define PDMPC_HB_WORDS 136

struct KernelArgs {
    // MPA
    const uint64_t* succ_mask;
    const int16_t* man_index;
    const DevManPose* man_pose;
    const double* man_area;  // double2 pairs
    int32_t n_trims, n_words, n_man, Hp;
    int32_t checker;
    int32_t areas_in_lds;
    double dt;
    // batch
    const DevVehicle* veh;
    const double* points;  // double2 pairs
    const int32_t* pred;
    pdmpc_vehicle_out* out;
    uint32_t* done_flag;
    uint32_t epoch;
    int32_t first;  // slot of blockIdx.x == 0
    // arenas
    NodeArena arena;
    uint32_t max_nodes;
    int32_t* tree_size;  // per slot: nodes in the arena after the search | 0x40000000 (| 0x20000000: ended on a replay) (debug read-back)
    // LDS
    LdsLayout lds;
    int32_t NL, NV, soup_cap, cand_cap;
    int32_t n_waves;                   // wavefronts per workgroup of this launch (workgroup size / 64)
    const double* sampled_random;      // sampled optimizer: [max_vehicles][sampled_n_random] mt19937ar doubles (host-generated)
    int32_t sampled_n_random;
    unsigned long long* work_count;    // [0] edge checks evaluated, [1] segment pairs they stand for, [2] nodes processed, [3] rounds, [4] shared rounds, [5] nodes checked by helpers, [6] plans that are not planning results (cumulative, all vehicles)
    int32_t* tie_count;                // [0] searches that ended on the binary heap (equal keys), [2] arrival events (cumulative)
    uint32_t* progress;    // host-mapped (pinned) words, 64 per slot: live counters, for debugging a launch that does not end (or null)
    int32_t debug_tail;    // 1: the search leaves its round / node / tick counters in the unused last rows of pdmpc_vehicle_out.path_nodes
    // helper workgroups (blockIdx >= n_searches): CUs the launch leaves idle check tiles of other workgroups' large rounds
    int32_t n_searches;              // workgroups of this launch that run a search (the first ones)
    int32_t n_helpers;               // helper workgroups behind them (0: none)
    int32_t bk_flags;                // bit 0: a small open set is selected by the first wavefront alone (PDMPC_TUNING=fast_select; bulk_search.hpp)
    unsigned long long* help_board;  // [slot][PDMPC_HB_WORDS]: see above
    uint32_t* help_verdict;          // [slot][PDMPC_HELP_CAP] 1 collision-free, 2 colliding, 3 crosses expected areas only (written by helpers)
    uint32_t* help_finished;         // searches that have published their result (runs on from launch to launch) ...
    uint32_t help_fin_base;          // ... and its value when this launch went out
    uint32_t launch_id;              // number of this launch on its handle (never 0): tells this launch's board words from an earlier launch's
    int32_t bk_ready_cap;   // entries of the ready list (a round's nodes)
    int32_t bk_round0;      // nodes a round of a young search takes
    int32_t bk_round;       // ... and the most any round takes
    int32_t bk_ramp;        // a round of a search that is no longer young grows by 1 / bk_ramp of the nodes processed so far
    int32_t bk_helpers_first;  // helper workgroups dispatched in front of the searches (the others follow them)
    int32_t bk_seat_nodes;     // a search is entitled to its share of the launch's helpers per this many nodes processed (bulk_helper_body)
    int32_t bk_tentative;   // 1: predecessors that are still planning have their expected areas (the ones they publish when exhausted) in their soup slots
    int32_t bk_tile;        // the most entries of a shared round one seat takes (what a helper stages in LDS; at most 768)
    int32_t bk_mid_min;     // a far list longer than this is not scanned by every refill of near: a band of its smallest keys is moved to mid first
    int32_t bk_mid_fill;    // ... about this many entries at a time
    int32_t bk_share_min;   // a round with at least this many entries is shared with the helper workgroups
    int32_t bk_force_tie;    // testing only: every search is treated as if it had met equal keys, i.e. ends on the replay of its tree through the reference's binary heap
    int32_t bk_fast_arrival; // 1: a finished search checks an arriving predecessor's areas against its plan's path first and publishes its own areas as soon as the last one has passed (the other collision-free nodes are verified afterwards)
    double* bk_post;        // [slot][bk_ready_cap][3] double2: what a check item reads of the tree, posted per entry of a shared round
    int32_t speculate;  // 1: start searching before all predecessors have finished (results are identical)
    uint32_t spin_limit;
    int32_t reverse_dispatch;  // testing only: workgroup b plans slot first + n_searches - 1 - b, i.e. successors are
                               // dispatched before their predecessors -- the adversarial order the watchdog + resident slices must survive
};

#ifdef __cplusplus
extern "C" {
#endif
// bulk_kernel*.hip: the graph search as bulk-synchronous passes (count searches + args->n_helpers helper workgroups in ONE launch) for the
// InterX checker with one successor-mask word / with any number of them, and for the separating-axis checker; lds_high_water = the
// handle's record of the dynamic LDS size set so far on that kernel
int pdmpc_launch_bulk(const KernelArgs* args, int count, void* stream, uint32_t* lds_high_water);
int pdmpc_launch_bulk_wide(const KernelArgs* args, int count, void* stream, uint32_t* lds_high_water);
int pdmpc_launch_bulk_sat(const KernelArgs* args, int count, void* stream, uint32_t* lds_high_water);
// bulk_kernel_compact.hip: the InterX / one-mask-word kernel built for 8 wavefronts and at most 80 KB of LDS: two workgroups per CU
int pdmpc_launch_bulk_compact(const KernelArgs* args, int count, void* stream, uint32_t* lds_high_water);
// util_kernels.hip: (cost-to-come of the final node, status) of n result records into lean[2 * n]
int pdmpc_launch_gather_lean(const pdmpc_vehicle_out* out, int n, int Hp, double* lean, void* stream);
// sampled_kernel.hip: the sampled optimizer (MonteCarloTreeSearch.m), `count` workgroups of one wavefront
int pdmpc_launch_sampled(const KernelArgs* args, int count, void* stream);
// debug_kernels.hip: the open-list command script on one wavefront, and the collision primitives on given polygons (one wavefront per case)
int pdmpc_launch_heap_script(const int32_t* op, const int32_t* id, const double* key, int n, int32_t* out, unsigned long long* stats, double* gkey, uint32_t* gid, int HL,
                             void* stream);
int pdmpc_launch_edge_check(int mode, int n_cases, const int32_t* a_off, const double* a_x, const double* a_y, const int32_t* b_off, const double* b_x,
                            const double* b_y, int32_t* hit, void* stream);
#ifdef __cplusplus
}
#endif

#endif
