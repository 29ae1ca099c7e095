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
/* Wavefronts per vehicle (workgroup size / 64), chosen per launch: wave 0 owns the pop order; block-min mode: wave 1 expands,
 * wave 2 scouts, the others validate; heap mode: all others pre-validate.  More validators shorten the critical vehicle's
 * search (C2: 5 -> 6.87 ms, 7 -> 6.67 ms, 13 -> 6.64 ms per step) but cost throughput when the chip is full (C5). */
#define PDMPC_WAVES_LATENCY 12    /* launches with at most one workgroup per CU */
#define PDMPC_WAVES_CROWDED 12    /* launches with more workgroups than CUs: the build for six wavefronts per SIMD, two workgroups per CU
                                     (measured: C4, 512 workgroups: 26.3 steps/s against 23.8 with the regular build at 10 wavefronts and 20.2
                                     at 8; C5, 1280 workgroups: 216.7 against 198.3 with 10, 184.4 with 8, regular build 185 at 8) */
/* Twelve wavefronts per workgroup = three per SIMD = 168 VGPRs per lane.  With sixteen (128 VGPRs) the frontier kernel spilled 44
 * VGPRs next to ~320 SGPRs held in VGPR lanes; in that regime hipcc 7.2 produced code that lost spilled values depending on
 * unrelated source changes.  At 168 the kernels spill 2-4 VGPRs, and no workload is slower (C2 753 -> 770
 * steps/s, C3 680 -> 685, C4 and C5 unchanged: their launches ran twelve wavefronts before). */
#define PDMPC_MAX_WAVES 12
#define PDMPC_QUEUE_HEAP 0     /* open list = libstdc++-faithful binary heap (exact for any keys) */
#define PDMPC_QUEUE_BLOCKMIN 1 /* open list = block-min queue while the minimal key is unique, binary heap after the first tie */
#define PDMPC_MAX_THREADS (PDMPC_WAVE * PDMPC_MAX_WAVES)
#define PDMPC_SH_WORDS 128 /* 32-bit LDS words shared by the waves of a workgroup (state, mail boxes, counters of the frontier search) */

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

// One search-tree node (Tree.m:3-13 row + what the pop of this node needs), 64 bytes so a node is four
// 16-byte accesses and the children of an expansion form one contiguous, coalesced store.
struct NodeRec {
    double x, y, yaw, g;  // Tree.x/y/yaw/g
    double cs, sn;        // cos/sin(yaw), filled when the node is expanded (expand_node.m:50-51); its children's
                          // edge checks read them (GraphSearch.m:155-156)
    double h;             // Tree.h
    uint32_t parent;      // Tree.parent (1-based id, 0 for the root)
    uint32_t packed;      // trim (10 bit, 1-based) | k << 10 (5 bit) | maneuver index << 15 (12 bit) | area columns << 27 (4 bit) | popped << 31
};
#define NODE_TRIM(p) ((int)((p) & 1023u))
#define NODE_K(p) ((int)(((p) >> 10) & 31u))
#define NODE_MAN(p) ((int)(((p) >> 15) & 4095u))
#define NODE_COLS(p) ((int)(((p) >> 27) & 15u))
#define NODE_POPPED_BIT 0x80000000u /* set by the sequencing wave when the node was popped with a valid edge */

// byte offsets of the regions of the dynamic LDS allocation (all multiples of 16)
struct LdsLayout {
    uint32_t mask, man_index, pose, area;  // MPA tables
    uint32_t ref;                          // ref_x[16], ref_y[16], dtv[16]
    uint32_t shape;                        // shape A [8], shape B [8] (double2)
    uint32_t path;                         // uint32 path[HP_MAX+1] + misc scratch
    uint32_t soup;                         // double2[soup_cap]
    uint32_t cand;                         // uint32[waves][cand_cap]: compacted candidate segments of one edge check
    uint32_t vstate;                       // uint8[NV]: validity cache (0 unknown, 1 valid, 2 invalid)
    uint32_t expand;                       // expansion scratch: dcum[16][16], term[16][16] doubles, child xy[16] double2
    uint32_t heap_key, heap_id;            // double[HL], uint32[HL]
    uint32_t stage;                        // frontier kernel: NodeRec[2 * fr_stage_cap]: the records of a round's nodes and of their parents
    uint32_t nodes;                        // NodeRec[NL]
    uint32_t total;
    // bulk kernel (bulk_kernel.hip)
    uint32_t bk_near_key, bk_near_id;      // double[BK_PER * threads], uint32[BK_PER * threads]: the LDS part of the open set
    uint32_t bk_ready;                     // uint32[bk_ready_cap] nodes of the round + uint32[bk_ready_cap] their collision flags
    uint32_t bk_hist;                      // uint32[3072]: histogram [2048] | goal list [1024], collision-free nodes of the round [1024], their children's offsets [1024]
    uint32_t bk_misc;                      // 2 KB: path tables of the best goal candidate, scan partials, chunk table, the reference's ids along the path | the selection's 256-bin histogram
    uint32_t bk_pshape;                    // double2[Hp][VMAX] + uint32[HP_MAX]: the areas along the path of the record written last and their column counts (what an arrival is checked against first)
};

struct NodeArena {  // HBM arrays, per-vehicle stride = max_nodes entries
    NodeRec* nodes;
    double* heap_key;
    uint32_t* heap_id;
    uint8_t* vstate;  // validity cache for nodes beyond the LDS-resident NV
    double* pop_log;  // block-min mode with dropping: keys of the pops in pop order (front), drop stamps as uint32 (from the back)
                      // frontier kernel: keys of the `far` open entries (their nodes in heap_id), phase B: a node's branch maximum
    double* near_key;  // frontier kernel: keys and nodes of the `near` open entries
    uint32_t* near_id;
    double* walk;      // bulk kernel: per node, 16 bytes: what its branch to the goal candidate's path looks like (bk_classify_wave)
    double* mid_key;   // bulk kernel: keys and nodes of the `mid` open entries (what a heavy search refills near from; far feeds it)
    uint32_t* mid_id;
    unsigned long long* link;  // frontier kernel: parent | packed << 32 per node, eight nodes to a 64-byte line (a record's own copy shares its line with nothing the walks need)
    uint32_t* child0;          // bulk kernel: 1-based arena index of a node's first child (its children are consecutive, ascending trim), 0 while it has none: what the replay of a tied search descends by
};

/* record status of a search that met equal keys where the pop order depends on the reference's binary heap (bulk kernel): never leaves
 * the library -- api.cpp plans the call again with the kernel that carries the libstdc++-faithful heap */
#define PDMPC_INTERNAL_TIE 100

#ifndef PDMPC_BK_PER
#define PDMPC_BK_PER 4
#endif
/* bulk kernel: entries of the LDS open list per thread (a selection pass holds them in registers) */

#define PDMPC_HELP_CAP 2048 /* entries of a round that can be shared (= the ready list's capacity) */
#define PDMPC_HB_WORDS 8   /* 64-bit words of a board */
#define PDMPC_HB_TICKET 0
#define PDMPC_HB_N 1
#define PDMPC_HB_MASK 2
#define PDMPC_HB_DONE 3
#define PDMPC_HB_NNODES 4 /* expanding helpers: the search's node counter while a round is shared */
#define PDMPC_HB_FLAGS 6  /* bit 0: the arena is full (set by a helper), bit 1: expand what you find collision-free (set by the owner) */

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
    int32_t* pop_trace;
    int32_t trace_cap;
    int32_t* tree_size;  // per slot: nodes in the tree after the search (debug read-back)
    // LDS
    LdsLayout lds;
    int32_t HL, NL, NV, soup_cap, cand_cap;
    int32_t speculate_expansion;       // 1: the expander wave works ahead on the node the queue wave will most likely hand over next
    int32_t n_waves;                   // wavefronts per vehicle of this launch (workgroup size / 64)
    int32_t n_validators;              // validator waves that take part (block-min mode; at most n_waves - 3)
    int32_t dense;                     // 1: launch the kernels compiled for six wavefronts per SIMD
    int32_t crowded;                   // 1: more workgroups than compute units in this launch (idle waves sleep longer between polls)
    int32_t queue_mode, bm_kr, bm_nb;  // PDMPC_QUEUE_*; block-min ring entries (power of two) and block count (multiple of 64)
    int32_t drop_invalid;              // block-min mode: entries known to collide leave the open list on the side (counted as popped at the end)
    int32_t drop_beyond_lds;           // ... also entries whose validity byte lives in HBM (nodes beyond the LDS-resident NV)
    int32_t eager_validation;          // block-min mode: idle validator waves evaluate every node's edge in creation order
    const double* sampled_random;      // sampled optimizer: [max_vehicles][sampled_n_random] mt19937ar doubles (host-generated)
    int32_t sampled_n_random;
    unsigned long long* work_count;    // [0] edge checks evaluated, [1] segment pairs they stand for, [2] entries dropped from the open list, [3] those counted as pops (cumulative, all vehicles)
    int32_t* tie_count;                // [0] searches redone on the binary heap after a tied minimum, [1] speculation restarts, [2] arrivals (cumulative)
    uint32_t* progress;    // host-mapped (pinned) words, 16 per slot: the frontier kernel's live counters, for debugging a launch that does not end (or null)
    int32_t debug_tail;    // 1: the frontier kernel leaves its round / node counters in the unused last row of pdmpc_vehicle_out.path_nodes
    int32_t frontier;      // 1: this launch runs the frontier kernel (frontier_kernel.hip), 0: the pop-ordered kernel (search_kernel.hip)
    int32_t fr_round;      // frontier kernel: open entries a round aims to take (about four per wavefront)
    int32_t fr_stage_cap;  // ... entries of a round whose records (node + parent) are staged in LDS when the round is selected
    int32_t fr_ramp;       // ... a young search takes one entry per wavefront plus 1 / fr_ramp of the nodes processed so far
    int32_t fr_near_fill;  // ... entries a refill moves from far to near
    double fr_join_scale;  // ... the key range within which a wave goes on with a node's best child, as a multiple of the range the round's own entries span
    int32_t fr_near_max;   // ... size of near beyond which its tail is moved back to far
    int32_t fr_dive;       // ... largest round (entries) in which waves go on with best children (0: never)
    int32_t fr_root_dive;  // ... 1: the root's round follows best children without a key limit
    // frontier kernel, helper workgroups (blockIdx >= n_searches): CUs the launch leaves idle check edges of other workgroups' large rounds
    int32_t n_searches;              // workgroups of this launch that run a search (the first ones)
    int32_t n_helpers;               // helper workgroups behind them (0: none)
    int32_t fr_share_min;            // a round with at least this many entries is shared with the helpers
    int32_t fr_own_div;              // ... of which the owner keeps 1 / fr_own_div (at least two per wavefront) for itself
    int32_t help_chunk;              // entries a helper claims at a time (a multiple of 32: its verdict words own whole cache lines)
    unsigned long long* help_board;  // [slot][8]: ticket = round << 32 | shared entries << 16 | next unclaimed entry, -, incorporated predecessors, entries finished by helpers
    uint32_t* help_list;             // [slot][PDMPC_HELP_CAP] nodes of the shared part of the round
    uint32_t* help_verdict;          // [slot][PDMPC_HELP_CAP] 1 collision-free, 2 colliding (written by helpers; a word each: a helper's run of 64 entries owns whole cache lines)
    double* help_cs;                 // [slot][PDMPC_HELP_CAP][2] cos, sin of the yaw of the collision-free entries that will be expanded (written by helpers)
    int32_t help_expand;             // 1: helpers also expand the entries they find collision-free
    int32_t help_patience;           // ... polls without a new claim after which the owner closes the round and does the rest itself
    uint32_t* help_finished;         // searches of this launch that have published their result
    uint32_t help_fin_base;          // bulk kernel: the counter's value when the launch went out (it runs on from launch to launch: no clearing in between)
    // bulk kernel
    int32_t bulk;           // 1: this launch runs the bulk kernel (bulk_kernel.hip)
    int32_t bk_ready_cap;   // entries of the ready list (a round's nodes)
    int32_t bk_round0;      // nodes a round of a young search takes
    int32_t bk_round;       // ... and the most any round takes
    int32_t bk_tentative;   // 1: predecessors that are still planning have their expected areas (the ones they publish when exhausted) in their soup slots
    int32_t bk_tile;        // entries of a tile of a shared round (what a helper workgroup claims at a time; at most 128)
    int32_t bk_mid_min;     // a far list longer than this is not scanned by every refill of near: a band of its smallest keys is moved to mid first
    int32_t bk_mid_fill;    // ... about this many entries at a time
    int32_t bk_share_min;   // a round with at least this many entries is shared with the helper workgroups
    int32_t bk_force_tie;    // testing only (PDMPC_BK_FORCE_TIE): every search is treated as if it had met equal keys, i.e. ends on the replay of its tree through the reference's binary heap
    int32_t bk_fast_arrival; // 1: a finished search checks an arriving predecessor's areas against its plan's path first and publishes its own areas as soon as the last one has passed (the other collision-free nodes are re-checked afterwards)
    double* bk_post;        // [slot][bk_ready_cap][3] double2: what a check item reads of the tree, posted per entry of a shared round
    int32_t speculate;  // 1: start searching before all predecessors have finished (results are identical, see arrival_sync)
    uint32_t spin_limit;
    int32_t reverse_dispatch;  // testing only (PDMPC_TEST_REVERSE_DISPATCH): workgroup b plans slot first + n_searches - 1 - b, i.e. successors are
                               // dispatched before their predecessors -- the adversarial order the watchdog + resident slices must survive
};

#ifdef __cplusplus
extern "C" {
#endif
// defined in search_kernel.hip; launches `count` workgroups of 64 threads on `stream`
int pdmpc_launch_search(const KernelArgs* args, int count, void* stream);
// defined in search_kernel.hip; runs the open-list command script on one wavefront (debug / unit test)
int pdmpc_launch_heap_script(const int32_t* op, const int32_t* id, const double* key, int n, int32_t* out, unsigned long long* stats,
                             double* gkey, uint32_t* gid, int HL, void* stream);
// defined in sampled_kernel.hip: the sampled optimizer (MonteCarloTreeSearch.m), `count` workgroups of one wavefront
int pdmpc_launch_sampled(const KernelArgs* args, int count, void* stream);
// defined in frontier_kernel.hip: the search with all wavefronts of a workgroup working on open nodes side by side
int pdmpc_launch_frontier(const KernelArgs* args, int count, void* stream, uint32_t* lds_high_water /* [5], kept by the handle */);
// defined in bulk_kernel.hip: the search as bulk-synchronous passes (count searches + args->n_helpers helper workgroups in ONE launch); lds_high_water[2] = the handle's record of the dynamic LDS size set so far per kernel variant
int pdmpc_launch_bulk(const KernelArgs* args, int count, void* stream, uint32_t* lds_high_water);
// defined in frontier_kernel.hip; launches args->n_helpers helper workgroups (they serve the searches of a pdmpc_launch_frontier with the same args)
int pdmpc_launch_helpers(const KernelArgs* args, void* stream, uint32_t* lds_high_water);
// defined in debug_kernels.hip: the collision primitives of edge_checks.hpp on given polygons, one wavefront per case
int pdmpc_launch_edge_check(int mode, int n_cases, const int32_t* a_off, const double* a_x, const double* a_y, const int32_t* b_off, const double* b_x,
                            const double* b_y, int32_t* hit, void* stream);
int pdmpc_launch_bm_script(const int32_t* op, const double* key, int n, int32_t* out, unsigned long long* stats, double* gkey, int KR, int NB,
                           void* stream);
#ifdef __cplusplus
}
#endif

#endif
