// api.cpp — host side of libpdmpc_hip.so: the C ABI declared in include/pdmpc.h.
//
// Responsibilities: own all device memory of a handle, flatten the caller's IterationData slices into
// the pointer-free HBM blob of pdmpc_device.h (this is where vectorize_all_obstacles.m:36-62's
// "[polygon, NaN]" concatenation happens for literal obstacles), size the LDS regions, launch the search
// kernel on the handle's stream and time it with HIP events, copy results back.
// There is no CPU implementation of the search in this library: without a gfx950 device every planning
// entry point fails with PDMPC_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/pdmpc.h"
#include "mt19937ar.hpp"
#include "pdmpc_device.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                                 \
    do {                                                                                             \
        hipError_t e__ = (expr);                                                                     \
        if (e__ != hipSuccess) {                                                                     \
            char buf__[512];                                                                         \
            snprintf(buf__, sizeof buf__, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return fail(PDMPC_ERR_HIP, buf__);                                                       \
        }                                                                                            \
    } while (0)

inline uint32_t align16(uint32_t v) { return (v + 15u) & ~15u; }

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max(n, (size_t)64);
        want += want / 2;
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e != hipSuccess) return (int)e;
        cap = want;
        return 0;
    }
    int ensure_exact(size_t n) {  // no head room: the arenas are sized in gigabytes
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc((void**)&p, std::max(n, (size_t)64) * sizeof(T));
        if (e != hipSuccess) return (int)e;
        cap = std::max(n, (size_t)64);
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

template <class T>
struct PinnedBuf {
    T* p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max(n, (size_t)64);
        want += want / 2;
        hipError_t e = hipHostMalloc((void**)&p, want * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) return (int)e;
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

const size_t kLdsMax = 160 * 1024;  // gfx950: 160 KiB per CU (MI355X_MICROARCH.md)

}  // namespace

// one packed batch: host mirror (pinned) + device copy, each ONE allocation -- [DevVehicle records | points pool | predecessor slots] --
// so a pack is one host-to-device copy
struct PackedStep {
    PinnedBuf<unsigned char> h_blob;
    DevBuf<unsigned char> d_blob;
    DevVehicle* h_veh = nullptr;  // (views into the blobs, set by pack_common)
    double* h_pts = nullptr;
    int32_t* h_pred = nullptr;
    DevVehicle* d_veh = nullptr;
    double* d_pts = nullptr;
    int32_t* d_pred = nullptr;
    uint64_t staged_serial = ~0ull;  // the handle's sync_serial when the copy out of h_blob was queued (pack_common)
    int n_packed = 0;
    int soup_cap = 0;
    int cand_cap = 0;  // most segments any single edge check can see (one step's soups + the boundary)
    std::vector<int64_t> lit_cols;  // per slot: literal soup + boundary columns (for the bytes formula)
    std::vector<int32_t> perm;      // empty: slot s holds the caller's vehicle s; else slot s holds vehicle perm[s] (pack_common put the batch into level order)
    std::vector<int32_t> inv;       // ... and vehicle v sits in slot inv[v]
    void release() {
        h_blob.release();
        d_blob.release();
        h_veh = d_veh = nullptr;
        h_pts = d_pts = nullptr;
        h_pred = d_pred = nullptr;
    }
};

// Tuning knobs and A/B switches (environment variables PDMPC_*), read ONCE in pdmpc_create: a launch makes no getenv call.
// Every switch leaves the results bit-identical; the defaults are the measured optima quoted next to their use.
struct Tuning {
    int fr_stage = -1;          // PDMPC_FR_STAGE: records staged per round (-1: by LDS budget)
    int fr_two_per_cu = 0;      // PDMPC_FR_TWO_PER_CU
    int bk_two_per_cu = 0;      // PDMPC_BK_TWO_PER_CU: bulk kernel, launches with more than two searches per CU: two workgroups of six wavefronts per CU (measured: slower)
    int debug_lds = 0;          // PDMPC_DEBUG_LDS
    int hl_max = 8192;          // PDMPC_HL_MAX (pop-ordered kernel)
    int bm_ring = -1;           // PDMPC_BM_RING (pop-ordered kernel; -1: by launch size)
    int nv_max = 65536;         // PDMPC_NV_MAX (pop-ordered kernel)
    int fr_ramp = -1;           // PDMPC_FR_RAMP (-1: 4, or 2 with expanding helpers)
    double fr_join_scale = 4.0; // PDMPC_FR_JOIN_SCALE
    int fr_root_dive = 0;       // PDMPC_FR_ROOT_DIVE
    int fr_dive = 1024;         // PDMPC_FR_DIVE
    uint32_t spin_limit = 1u << 22;  // PDMPC_SPIN_LIMIT
    int debug_tail = 0;         // PDMPC_DEBUG_TAIL
    int debug_progress = 0;     // PDMPC_DEBUG_PROGRESS
    int dense = -1;             // PDMPC_DENSE (-1: follows the layout)
    int drop = -1;              // PDMPC_DROP (pop-ordered kernel)
    int drop_beyond_lds = 1;    // PDMPC_DROP_BEYOND_LDS
    int eager = -1;             // PDMPC_EAGER
    int fr_share_min = 128;     // PDMPC_FR_SHARE_MIN
    int fr_own_div = 8;         // PDMPC_FR_OWN_DIV
    int help_chunk = 0;         // PDMPC_HELP_CHUNK (0: 32 expanding, 64 checking)
    int helpers = -1;           // PDMPC_HELPERS (-1: by launch size, 0: none)
    int helpers_oversub = -1;   // PDMPC_HELPERS_OVERSUB
    int help_expand = 1;        // PDMPC_HELP_EXPAND
    int help_patience = 8;      // PDMPC_HELP_PATIENCE
    int help_expand_oversub = -1;  // PDMPC_HELP_EXPAND_OVERSUB (-1: up to two searches per CU)
    int bk_round0 = 24;         // PDMPC_BK_ROUND0: nodes a round of a young search takes (bulk kernel)
    int bk_round = -1;          // PDMPC_BK_ROUND: the most a round takes (bulk kernel; -1: 1000 with helper workgroups, else 256)
    int bk_tentative = 1;       // PDMPC_BK_TENTATIVE: expected areas of predecessors that are still planning (A/B switch: results are identical)
    int bk_tile = -1;           // PDMPC_BK_TILE: nodes of a tile of a shared round (-1: by launch size)
    int bk_mid_min = 24576;     // PDMPC_BK_MID_MIN: far lists longer than this feed near through the mid list (a band of far's smallest keys)
    int bk_mid_fill = 12288;    // PDMPC_BK_MID_FILL: entries a refill of mid aims at
    int bk_share_min = 192;     // PDMPC_BK_SHARE_MIN: a round with at least this many nodes is shared with the helper workgroups
    int bk_force_tie = 0;       // PDMPC_BK_FORCE_TIE: testing only: every search ends on the replay through the reference's binary heap (as if it had met equal keys)
    int bk_fast_arrival = 1;    // PDMPC_BK_FAST_ARRIVAL: finished searches check arrivals against their plan's path first and publish early (A/B switch: results are identical)
    int bk_ramp = -1;           // PDMPC_BK_RAMP: a round grows by 1 / bk_ramp of the nodes processed so far (-1: 2 with helper workgroups, else 4)
    int fr_slice = -1;          // PDMPC_FR_SLICE (-1: only after a predecessor time-out, 0: never, 1: always when oversubscribed)
    int debug_host = 0;         // PDMPC_DEBUG_HOST
    int help_first = 0;         // PDMPC_HELP_FIRST (diagnostic: the helper kernel enqueued in front of the searches)
    int help_prio = 1;          // PDMPC_HELP_PRIO (0: the helper stream at the launch stream's priority)
    int slot_order_reverse = 0; // PDMPC_TEST_REVERSE_DISPATCH: testing only, see launch_range
};

struct pdmpc_handle {
    pdmpc_config cfg{};
    Tuning tune{};
    hipStream_t stream = nullptr;
    int n_cu = 256;
    // MPA
    bool has_mpa = false;
    int n_trims = 0, n_words = 0, n_man = 0;
    DevBuf<uint64_t> d_mask;
    DevBuf<int16_t> d_mi;
    DevBuf<DevManPose> d_pose;
    DevBuf<double> d_area;
    size_t mask_bytes = 0, mi_bytes = 0;
    int64_t mpa_alg_bytes = 0;
    // arenas
    uint32_t max_nodes = 0;
    uint32_t max_nodes_limit = 0;  // pdmpc_plan_* may grow the arenas up to this many nodes per vehicle (0: as far as HBM allows)
    int64_t arena_regrows = 0;     // times an overflowed call was re-planned with larger arenas
    int64_t safe_replans = 0;      // times a call was re-planned in resident slices after a predecessor time-out
    bool safe_launches = false;    // pdmpc_set_safe_launch: every launch in resident slices
    int max_vehicles = 0;
    DevBuf<NodeRec> anodes;
    DevBuf<double> ahk;
    DevBuf<uint32_t> ahid;
    DevBuf<double> alog;
    DevBuf<double> ankey;   // frontier kernel: near list
    DevBuf<unsigned long long> alink;  // frontier kernel: parent | packed << 32 of every node (the walks' and the counting pass's compact view of the tree)
    DevBuf<uint32_t> anid;
    DevBuf<double> awalk;   // bulk kernel: two doubles per node (NodeArena::walk)
    DevBuf<double> amidk;   // bulk kernel: mid list
    DevBuf<uint32_t> amidi;
    DevBuf<uint32_t> achild0;  // bulk kernel: first child per node (NodeArena::child0)
    DevBuf<uint8_t> avs;
    DevBuf<pdmpc_vehicle_out> d_out;
    DevBuf<uint32_t> d_flag;
    DevBuf<int32_t> d_tree_size;
    DevBuf<int32_t> d_tie_count;
    DevBuf<unsigned long long> d_work_count;
    DevBuf<unsigned long long> d_help_board;  // frontier kernel, helper workgroups (pdmpc_device.h)
    DevBuf<uint32_t> d_help_list, d_help_finished;
    DevBuf<uint32_t> d_help_verdict;
    DevBuf<double> d_help_cs;
    int helpers_max = 64;
    hipStream_t help_stream = nullptr;  // the helper kernel runs next to the searches, on its own stream ...
    hipStream_t help_stream_low = nullptr;  // ... of the lowest priority when the launch has more searches than CUs
    hipEvent_t ev_help_pre = nullptr, ev_help_done = nullptr;
    DevBuf<double> d_random;  // sampled optimizer: random numbers of the batch
    int sampled_n_random = 0;
    bool sampled_launch = false;
    int kernel_frontier = 1;  // 1: round-based kernels (open nodes processed side by side, the reference's order reconstructed), 0: the pop-ordered kernel of round 1
    int kernel_bulk = 1;      // 1: of the round-based kernels the bulk one (bulk_kernel.hip) where it applies (InterX checker), 0: the frontier kernel
    bool force_frontier = false;  // this launch: the frontier kernel, which carries the binary heap (a search of the bulk kernel met a tie)
    bool last_launch_bulk = false;
    int last_first = 0, last_count = 0;  // slots of the last launch_range
    std::vector<std::pair<int, int>> step_ranges;  // the slot ranges launched with the bulk kernel since the step began (this epoch): what a tie plans again
    uint32_t step_ranges_epoch = 0;
    bool boards_dirty = true;            // the helper boards / the finished counter need clearing before the bulk kernel's helpers may read them
    uint32_t help_fin_total = 0;         // value of the finished counter once every launch so far has ended (bulk kernel)
    double dbg_us[4] = {0, 0, 0, 0};     // PDMPC_DEBUG_HOST=2: pack, launch, fetch (host clock) and kernel (events) time of the plan_batch calls
    uint64_t sync_serial = 0;            // stream synchronisations through sync_stream so far (PackedStep::staged_serial)
    std::vector<double> pack_pts;        // pack_common's scratch (kept: a pack allocates nothing once warm)
    std::vector<int32_t> pack_pred;
    std::vector<DevVehicle> pack_veh;
    PinnedBuf<pdmpc_vehicle_out> h_out;  // pdmpc_fetch_results: the records land in pinned memory (a copy into the caller's pageable array goes through the runtime's staging otherwise)
    uint32_t bulk_lds_hw[3] = {0, 0, 0}; // dynamic LDS size set so far on the bulk kernel's variants and its helper kernel (hipFuncSetAttribute is a maximum)
    uint32_t frontier_lds_hw[5] = {0, 0, 0, 0, 0};  // ... and on the frontier kernel's four variants and its helper kernel
    DevBuf<double> d_bk_post;            // bulk kernel: records posted for the helper workgroups (pdmpc_device.h)
    int bk_ready_cap = 2048;     // entries of the bulk kernel's ready list with helper workgroups (PDMPC_BK_READY), half of it without
    int bk_ready_launch = 2048;  // ... of the last layout
    int64_t tie_replans = 0;  // launches planned again with the heap-carrying kernel because a search of the bulk kernel met a tie
    int fr_round = 0, fr_near_fill = 2048, fr_near_max = 4096;  // measured on C2 / C3 (round cap 768): 1024/2048 -> 358 / 345 steps/s, 2048/4096 -> 369 / 357, 4096/8192 -> 356 / 351
    bool last_launch_frontier = false;
    uint32_t* progress = nullptr;  // pinned, PDMPC_DEBUG_PROGRESS=1
    int queue_mode = PDMPC_QUEUE_BLOCKMIN;
    int speculate_expansion = 1;
    int waves_latency = PDMPC_WAVES_LATENCY, waves_crowded = PDMPC_WAVES_CROWDED;
    int n_validators = PDMPC_MAX_WAVES;  // (all there are)
    int n_waves = PDMPC_WAVES_LATENCY;   // of the last layout
    int bm_kr = 0, bm_nb = 0;
    int two_per_cu = 0;   // the layout of the last launch leaves room for two workgroups per CU (80 KB each, dense build)
    int fr_stage_cap = 0;
    int fr_cand_cap = 0;  // frontier kernel: 32-bit words of a wave's scratch (its candidate list)
    DevBuf<int32_t> d_trace;
    // batch blobs: several packed steps can stay resident side by side ("banks", pdmpc_select_bank)
    std::vector<PackedStep> banks;
    int bank = 0;
    uint32_t epoch = 1;  // done flags start at 0, so no slot looks solved before its first launch
    // launches
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    LdsLayout lds{};
    int HL = 0, NL = 0, NV = 0, areas_in_lds = 0;
    int speculate = 1;
    pdmpc_stats stats{};
};

namespace {

// hipStreamSynchronize on the launch stream, counted: a bank whose staging copy was queued before is free again (pack_common)
inline hipError_t sync_stream(pdmpc_handle* h) {
    const hipError_t e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) h->sync_serial += 1;
    return e;
}

// LDS layout of the frontier kernel for one choice of (budget, wavefronts, maneuver areas in LDS or read through L2).
// Regions: MPA tables, reference, per-wave shapes, shared words, obstacle soup, per-wave scratch (candidate list of an edge
// check = expansion scratch of 16 x HP_MAX cost terms + 16 child positions = 12 B per thread of phase B's chunk state),
// d_traveled table, ready list + histogram (also where the binary heap of the tie fallback lives), validity bytes, nodes.
bool layout_frontier(pdmpc_handle* h, size_t budget, int n_waves, int areas, int soup_cap, int cand_cap, LdsLayout& L, uint32_t& hl, uint32_t& nv, uint32_t& nl,
                     uint32_t& wscr, uint32_t& stage_cap) {
    uint32_t off = 0;
    L.mask = off;
    off = align16(off + (uint32_t)h->mask_bytes);
    L.man_index = off;
    off = align16(off + (uint32_t)h->mi_bytes);
    L.pose = off;
    off = align16(off + (uint32_t)(h->n_man * sizeof(DevManPose)));
    L.area = off;
    if (areas) off = align16(off + (uint32_t)(h->n_man * 3 * PDMPC_VMAX * 16));
    L.ref = off;
    off += 3 * PDMPC_HP_MAX * 8;
    L.shape = off;
    off += (uint32_t)n_waves * (2 * PDMPC_VMAX + 1) * 16;
    L.path = off;
    off += align16((PDMPC_HP_MAX + 2) * 4 + 2 * (PDMPC_HP_MAX + 1) * 4 + PDMPC_SH_WORDS * 4 + PDMPC_HP_MAX * 4);
    L.soup = off;
    off = align16(off + (uint32_t)std::max(soup_cap, 1) * 16);
    wscr = align16(std::max<uint32_t>((uint32_t)std::max(cand_cap, 1) * 4u, 16u * PDMPC_HP_MAX * 8u + 16u * 16u));
    L.cand = off;
    off += wscr * (uint32_t)n_waves;
    L.expand = off;
    off += (2 * PDMPC_HP_MAX * PDMPC_HP_MAX) * 8 + 16 * 16;
    const uint32_t region = 2048u * 4u + 2048u * 4u + 256u;
    const uint32_t min_nodes = 64 * (uint32_t)sizeof(NodeRec) + 1024;
    if ((size_t)off + region + min_nodes + 256 > budget) return false;
    L.heap_key = off;
    hl = std::min((region / 12u) & ~3u, h->max_nodes & ~3u);
    L.heap_id = off + align16(hl * 8);
    off += region;
    // staged records of a round (node + parent, 128 B per entry): up to 256 entries where a CU has the LDS to itself
    uint32_t rest = (uint32_t)(budget - off - 256);
    stage_cap = (budget > kLdsMax / 2) ? 256u : 64u;
    if (h->tune.fr_stage >= 0) stage_cap = (uint32_t)std::min(1024, h->tune.fr_stage);  // tuning knob
    while (stage_cap && stage_cap * 128u + min_nodes > rest) stage_cap /= 2u;
    L.stage = off;
    off += stage_cap * 128u;
    rest -= stage_cap * 128u;
    nv = std::min<uint32_t>(32768u, rest / 2);
    nv = std::min(nv, h->max_nodes) & ~15u;
    nl = std::min((rest - nv) / (uint32_t)sizeof(NodeRec), h->max_nodes);
    L.vstate = off;
    off += align16(nv);
    L.nodes = off;
    off += nl * (uint32_t)sizeof(NodeRec);
    L.total = align16(off);
    return L.total <= budget;
}

// LDS layout of the bulk kernel: MPA tables, reference, per-wave tallies, shared words, obstacle soup, phase B's chunk state (12 B per
// thread), d_traveled table, the LDS part of the open set (PDMPC_BK_PER entries per thread), the ready list with its collision
// flags, the histogram / goal list / expansion lists, 1 KB of small tables, validity bytes, then as many node records as fit.
bool layout_bulk(pdmpc_handle* h, size_t budget, int n_waves, int areas, int soup_cap, LdsLayout& L, uint32_t& nv, uint32_t& nl, uint32_t ready_cap) {
    const uint32_t threads = (uint32_t)n_waves * PDMPC_WAVE;
    uint32_t off = 0;
    L.mask = off;
    off = align16(off + (uint32_t)h->mask_bytes);
    L.man_index = off;
    off = align16(off + (uint32_t)h->mi_bytes);
    L.pose = off;
    off = align16(off + (uint32_t)(h->n_man * sizeof(DevManPose)));
    L.area = off;
    if (areas) off = align16(off + (uint32_t)(h->n_man * 3 * PDMPC_VMAX * 16));
    L.ref = off;
    off += 3 * PDMPC_HP_MAX * 8;
    L.shape = off;
    off += (uint32_t)n_waves * (2 * PDMPC_VMAX + 1) * 16;
    L.path = off;
    off += align16((PDMPC_HP_MAX + 2) * 4 + 2 * (PDMPC_HP_MAX + 1) * 4 + PDMPC_SH_WORDS * 4 + PDMPC_HP_MAX * 4);
    L.soup = off;
    off = align16(off + (uint32_t)std::max(soup_cap, 1) * 16);
    L.cand = off;
    off += align16(12u * threads);
    L.expand = off;
    off += (2 * PDMPC_HP_MAX * PDMPC_HP_MAX) * 8 + 16 * 16;
    L.bk_near_key = off;
    off += align16(PDMPC_BK_PER * threads * 8u);
    L.bk_near_id = off;
    off += align16(PDMPC_BK_PER * threads * 4u);
    L.bk_ready = off;
    off += align16(ready_cap * 8u);
    L.bk_hist = off;
    off += 3072u * 4u;
    L.bk_misc = off;
    off += 2048u;
    L.bk_pshape = off;
    off += align16((uint32_t)h->cfg.Hp * PDMPC_VMAX * 16u + PDMPC_HP_MAX * 4u);
    L.heap_key = L.bk_hist;  // (the prologue derives pointers from these; the bulk kernel never follows them)
    L.heap_id = L.bk_hist;
    L.stage = L.bk_hist;
    const uint32_t min_nodes = 64 * (uint32_t)sizeof(NodeRec) + 1024;
    if ((size_t)off + min_nodes + 256 > budget) return false;
    const uint32_t rest = (uint32_t)(budget - off - 256);
    nv = std::min<uint32_t>(16384u, std::max<uint32_t>(1024u, rest / 6));
    nv = std::min(nv, h->max_nodes) & ~15u;
    nl = std::min((rest - nv) / (uint32_t)sizeof(NodeRec), h->max_nodes);
    L.vstate = off;
    off += align16(nv);
    L.nodes = off;
    off += nl * (uint32_t)sizeof(NodeRec);
    L.total = align16(off);
    return L.total <= budget;
}

bool use_bulk(const pdmpc_handle* h) {
    // (InterX checker, one successor-mask word: every BASELINE automaton; the SAT checker, automata with more than 64 trims and a
    // search that met a tie run the frontier kernel)
    return h->kernel_frontier && h->kernel_bulk && !h->force_frontier && !h->sampled_launch && h->cfg.checker == PDMPC_CHECK_INTERX && h->n_words == 1;
}

// helper workgroups serve the bulk kernel's launches that leave CUs idle (launch_range)
bool bulk_has_helpers(const pdmpc_handle* h, int n_launch) {
    if (!h->speculate || h->tune.helpers == 0) return false;
    if (n_launch > h->n_cu) return n_launch <= 2 * h->n_cu && h->tune.helpers_oversub != 0;  // (the tail of a launch with up to two searches per CU)
    return n_launch <= h->n_cu - 2;
}

int compute_lds_bulk(pdmpc_handle* h, int n_launch, int soup_cap) {
    // large rounds pay where helper workgroups share them; without helpers the LDS is better spent on node records
    h->bk_ready_launch = bulk_has_helpers(h, n_launch) ? h->bk_ready_cap : std::max(256, h->bk_ready_cap / 2);
    // Launches with more than two searches per CU (C5: five) CAN run two workgroups of half as many wavefronts per CU, 80 KB of LDS each —
    // the same twelve wavefronts and the same register budget per CU, and while one search sits at a barrier or waits for memory the
    // other computes (PDMPC_BK_TWO_PER_CU=1).  Measured on C5: 388 steps/s against 479 with one workgroup of twelve per CU — in 80 KB the
    // maneuver areas have to go to L2 and only 70-110 node records stay in LDS (471 otherwise), and a search of 450 nodes lives on those.
    struct Try { size_t budget; int waves, areas; };
    std::vector<Try> tries;
    if (h->tune.bk_two_per_cu && n_launch > 2 * h->n_cu && !bulk_has_helpers(h, n_launch)) {
        tries.push_back({kLdsMax / 2, h->waves_latency / 2, 1});
        tries.push_back({kLdsMax / 2, h->waves_latency / 2, 0});
    }
    tries.push_back({kLdsMax, h->waves_latency, 1});
    tries.push_back({kLdsMax, h->waves_latency, 0});
    for (const Try& t : tries) {
        const int areas = t.areas;
        LdsLayout L{};
        uint32_t nv = 0, nl = 0;
        const int ready = std::min(h->bk_ready_launch, 3 * PDMPC_WAVE * t.waves);
        if (!layout_bulk(h, t.budget, t.waves, areas, soup_cap, L, nv, nl, (uint32_t)ready)) continue;
        if (h->tune.debug_lds)
            fprintf(stderr, "pdmpc LDS layout (bulk): launch %d budget %zu waves %d areas %d near %u ready %d nv %u nl %u total %u\n", n_launch, t.budget, t.waves, areas,
                    PDMPC_BK_PER * (uint32_t)t.waves * PDMPC_WAVE, ready, nv, nl, L.total);
        h->bk_ready_launch = ready;
        h->lds = L;
        h->n_waves = t.waves;
        h->HL = 0;
        h->NL = (int)nl;
        h->NV = (int)nv;
        h->areas_in_lds = areas;
        h->bm_kr = 0;
        h->bm_nb = 64;
        h->fr_cand_cap = 0;
        h->fr_stage_cap = 0;
        h->two_per_cu = 0;
        return PDMPC_OK;
    }
    char buf[256];
    snprintf(buf, sizeof buf, "obstacle soup (%d columns) + MPA tables do not fit into %zu B of LDS", soup_cap, kLdsMax);
    return fail(PDMPC_ERR_CAPACITY, buf);
}

int compute_lds_frontier(pdmpc_handle* h, int n_launch, int soup_cap, int cand_cap) {
    // More workgroups than CUs: two workgroups of waves_crowded wavefronts per CU (80 KB each) if the problem fits, with the
    // maneuver areas read through L2 if need be; otherwise one workgroup per CU with the whole LDS.
    struct Try { size_t budget; int waves, areas, two_per_cu; };
    std::vector<Try> tries;
    // (two workgroups per CU only on request: with more searches than CUs one 16-wave workgroup per CU at a time is faster,
    // C5: 345 steps/s against 324 with 2 x 8 wavefronts)
    const bool crowded = n_launch > h->n_cu && h->tune.fr_two_per_cu != 0;  // tuning knob
    if (crowded) {
        tries.push_back({kLdsMax / 2, h->waves_crowded, 1, 1});
        tries.push_back({kLdsMax / 2, h->waves_crowded, 0, 1});
    }
    tries.push_back({kLdsMax, h->waves_latency, 1, 0});
    tries.push_back({kLdsMax, h->waves_latency, 0, 0});
    for (const Try& t : tries) {
        LdsLayout L{};
        uint32_t hl = 0, nv = 0, nl = 0, wscr = 0, stage_cap = 0;
        if (!layout_frontier(h, t.budget, t.waves, t.areas, soup_cap, cand_cap, L, hl, nv, nl, wscr, stage_cap)) continue;
        if (h->tune.debug_lds)
            fprintf(stderr, "pdmpc LDS layout (frontier): launch %d budget %zu waves %d areas %d wscr %u heap fallback %u stage %u nv %u nl %u total %u\n", n_launch, t.budget,
                    t.waves, t.areas, wscr, hl, stage_cap, nv, nl, L.total);
        h->lds = L;
        h->n_waves = t.waves;
        h->HL = (int)hl;
        h->NL = (int)nl;
        h->NV = (int)nv;
        h->areas_in_lds = t.areas;
        h->bm_kr = 0;
        h->bm_nb = 64;
        h->fr_cand_cap = (int)(wscr / 4u);
        h->fr_stage_cap = (int)stage_cap;
        h->two_per_cu = t.two_per_cu;
        return PDMPC_OK;
    }
    char buf[256];
    snprintf(buf, sizeof buf, "obstacle soup (%d columns) + MPA tables do not fit into %zu B of LDS", soup_cap, kLdsMax);
    return fail(PDMPC_ERR_CAPACITY, buf);
}

int compute_lds(pdmpc_handle* h, int n_launch, int soup_cap_in, int cand_cap_in) {
    if (use_bulk(h)) return compute_lds_bulk(h, n_launch, soup_cap_in);
    if (h->kernel_frontier && !h->sampled_launch) return compute_lds_frontier(h, n_launch, soup_cap_in, cand_cap_in);
    h->two_per_cu = n_launch > h->n_cu ? 1 : 0;
    const int Hp = h->cfg.Hp;
    struct { int soup_cap, cand_cap; } hb{soup_cap_in, cand_cap_in};
    const size_t budget = (n_launch > h->n_cu) ? kLdsMax / 2 : kLdsMax;  // 2 workgroups of 4 waves per CU still fit
    LdsLayout L{};
    uint32_t off = 0;
    L.mask = off;
    off = align16(off + (uint32_t)h->mask_bytes);
    L.man_index = off;
    off = align16(off + (uint32_t)h->mi_bytes);
    L.pose = off;
    off = align16(off + (uint32_t)(h->n_man * sizeof(DevManPose)));
    const uint32_t area_bytes = (uint32_t)(h->n_man * 3 * PDMPC_VMAX * 16);
    // fixed part after the tables
    const uint32_t ref_bytes = 3 * PDMPC_HP_MAX * 8;
    // one workgroup per CU: all the wavefronts a workgroup can have; more: twelve each, two workgroups per CU
    const int n_waves = (n_launch > h->n_cu) ? h->waves_crowded : h->waves_latency;
    h->n_waves = n_waves;
    const uint32_t shape_bytes = (uint32_t)n_waves * (2 * PDMPC_VMAX + 1) * 16;  // two shapes + the wave's work tally
    const uint32_t path_bytes = align16((PDMPC_HP_MAX + 2) * 4 + 2 * (PDMPC_HP_MAX + 1) * 4 + PDMPC_SH_WORDS * 4 + PDMPC_HP_MAX * 4);  // ... + shared words + ...
    const uint32_t soup_bytes = (uint32_t)std::max(hb.soup_cap, 1) * 16;
    const uint32_t expand_bytes = (2 * PDMPC_HP_MAX * PDMPC_HP_MAX) * 8 + 16 * 16;
    const uint32_t fixed_rest = ref_bytes + shape_bytes + path_bytes + soup_bytes + expand_bytes;
    const uint32_t cand_bytes = align16((uint32_t)std::max(hb.cand_cap, 1) * 4 * (uint32_t)n_waves);
    const uint32_t min_bytes = 64 * 12 + 64 * (uint32_t)sizeof(NodeRec);
    // The maneuver areas stay in LDS only if the open list still gets a useful share: 32 KB (block-min queue with a
    // 2048-entry key ring for 64 k nodes); otherwise the edge checks read them through L2.
    int areas = 1;
    if ((size_t)off + area_bytes + fixed_rest + cand_bytes + std::max<uint32_t>(min_bytes, 44 * 1024) + 256 > budget) areas = 0;
    L.area = off;
    if (areas) off = align16(off + area_bytes);
    L.ref = off;
    off += ref_bytes;
    L.shape = off;
    off += shape_bytes;
    L.path = off;
    off += path_bytes;
    L.soup = off;
    off = align16(off + soup_bytes);
    L.cand = off;
    off += cand_bytes;
    L.expand = off;
    off += expand_bytes;
    if ((size_t)off + min_bytes + 256 > budget) {
        char buf[256];
        snprintf(buf, sizeof buf, "obstacle soup (%d columns) + MPA tables need %u B of LDS, budget %zu B", hb.soup_cap, off, budget);
        return fail(PDMPC_ERR_CAPACITY, buf);
    }
    // The open list gets up to three quarters of what is left.  Its region serves the binary heap (hl entries x 12 B)
    // or the block-min queue (key ring kr x 8 B; block minima and popped bits nb x 16 B; group minima 512 B).
    const uint32_t rest = (uint32_t)(budget - off - 256);
    const uint32_t region_cap = rest * 3 / 4;
    const uint32_t hl_max = (uint32_t)std::max(64, h->tune.hl_max);  // default 8192 = 13 heap levels; measured on C2: 4096 -> 8192 entries = +3.5 % steps/s
    uint32_t hl = std::min(hl_max, region_cap / 12);
    hl = std::min(hl, h->max_nodes) & ~3u;
    uint32_t region = align16(hl * 8) + align16(hl * 4);
    {
        const uint32_t nb = ((h->max_nodes + 63u) / 64u + 63u) & ~63u;
        const uint32_t bm_fixed = nb * 16u + 512u;
        h->bm_nb = (int)nb;
        h->bm_kr = 0;
        if (nb <= 4096u && region_cap >= bm_fixed + 512u * 8u) {
            // measured on C2: 1024 .. 8192 entries make no difference.  With two workgroups per CU the LDS is better spent on
            // validity bytes (C4: 23.8 steps/s with 512 entries, 23.5 with 1024, 22.5 with 2048)
            uint32_t kr = 512, kr_max = (n_launch > h->n_cu) ? 512 : 2048;
            if (h->tune.bm_ring > 0) kr_max = (uint32_t)std::max(512, h->tune.bm_ring);  // tuning knob
            while (kr * 2u * 8u + bm_fixed <= region_cap && kr * 2u <= kr_max) kr *= 2u;
            h->bm_kr = (int)kr;
            region = std::max(region, kr * 8u + bm_fixed);
        }
        if (h->queue_mode == PDMPC_QUEUE_BLOCKMIN && h->bm_kr != 0) {
            // the heap only runs after a fallback: it takes what the block-min queue needs, not the other way round
            region = (uint32_t)h->bm_kr * 8u + bm_fixed;
            while (align16(hl * 8) + align16(hl * 4) > region) hl -= 4;
        }
        // (bm_kr == 0: no room, or more than 262144 nodes per vehicle -> this launch uses the binary heap)
    }
    // validity cache: one byte per node for the first NV nodes (three quarters of what is left, at most 65536)
    const uint32_t nv_max = (uint32_t)std::max(1024, h->tune.nv_max);  // tuning knob (default 65536)
    uint32_t nv = std::min(nv_max, (rest - region) / 4 * 3);
    nv = std::min(nv, h->max_nodes) & ~15u;
    uint32_t nl = (rest - region - nv) / (uint32_t)sizeof(NodeRec);
    nl = std::min(nl, h->max_nodes);
    L.heap_key = off;
    L.heap_id = off + align16(hl * 8);
    off += region;
    L.vstate = off;
    off += align16(nv);
    L.nodes = off;
    off += nl * (uint32_t)sizeof(NodeRec);
    L.total = align16(off);
    if (L.total > budget) return fail(PDMPC_ERR_CAPACITY, "internal: LDS layout exceeds budget");
    if (h->tune.debug_lds)
        fprintf(stderr, "pdmpc LDS layout: launch %d budget %zu fixed %u rest %u region %u (cap %u) hl %u ring %d blocks %d nv %u nl %u total %u queue %d\n", n_launch,
                budget, (unsigned)(budget - 256 - rest), rest, region, region_cap, hl, h->bm_kr, h->bm_nb, nv, nl, L.total, h->queue_mode);
    h->lds = L;
    h->HL = (int)hl;
    h->NL = (int)nl;
    h->NV = (int)nv;
    h->areas_in_lds = areas;
    (void)Hp;
    return PDMPC_OK;
}

inline void push_pt(std::vector<double>& pts, double x, double y) {
    pts.push_back(x);
    pts.push_back(y);
}

int check_set(const pdmpc_polygon_set& s, const char* what) {
    if (s.n_polygons < 0) return fail(PDMPC_ERR_INVALID, std::string(what) + ": negative polygon count");
    if (s.n_polygons > 0 && (!s.offset || !s.x || !s.y)) return fail(PDMPC_ERR_INVALID, std::string(what) + ": null pointer");
    for (int i = 0; i < s.n_polygons; ++i)
        if (s.offset[i + 1] < s.offset[i]) return fail(PDMPC_ERR_INVALID, std::string(what) + ": offsets not monotone");
    return PDMPC_OK;
}

int pack_common(pdmpc_handle* h, int n, const pdmpc_vehicle_in* in, const int32_t* pred_offset, const int32_t* pred_index,
                const pdmpc_polygon_set* fallback) {
    if (!h) return fail(PDMPC_ERR_INVALID, "null handle");
    if (!h->has_mpa) return fail(PDMPC_ERR_NO_MPA, "pdmpc_upload_mpa has not been called");
    if (n < 0 || (n > 0 && !in)) return fail(PDMPC_ERR_INVALID, "bad vehicle array");
    if (n > h->max_vehicles) return fail(PDMPC_ERR_CAPACITY, "batch larger than config.max_vehicles");
    const int Hp = h->cfg.Hp;
    PackedStep& B = h->banks[h->bank];
    const double qnan = std::numeric_limits<double>::quiet_NaN();
    // the staging blob is reused: a copy out of it that may still be in flight (no stream synchronisation since it was queued) ends first
    if (B.staged_serial == h->sync_serial) HIPCHK(sync_stream(h));
    std::vector<double>& pts = h->pack_pts;
    std::vector<int32_t>& pred = h->pack_pred;
    std::vector<DevVehicle>& veh = h->pack_veh;
    pts.clear();
    pred.clear();
    veh.resize((size_t)std::max(n, 1));
    B.lit_cols.assign((size_t)n, 0);
    int soup_cap = 0, cand_cap = 0;
    // Slot order.  A search spins for predecessors of the same launch, so every predecessor must sit in a lower slot than its
    // successors (launch_range: forward progress of oversubscribed launches).  Callers hand the vehicles over in level order
    // (kahn.m); a batch that is not is put into level order here -- computation levels by longest path, stable within a level --
    // and pdmpc_fetch_results hands the records back in the caller's order.
    B.perm.clear();
    B.inv.clear();
    if (pred_offset) {
        bool ordered = true;
        for (int i = 0; i < n && ordered; ++i)
            for (int q = pred_offset[i]; q < pred_offset[i + 1]; ++q) {
                const int ps = pred_index[q];
                if (ps < 0 || ps >= h->max_vehicles) return fail(PDMPC_ERR_INVALID, "predecessor slot out of range");
                if (ps < n && ps >= i) ordered = false;
            }
        if (!ordered) {
            std::vector<int32_t> level(n, 0), indeg(n, 0), succ_off(n + 1, 0), succ, queue;
            for (int i = 0; i < n; ++i)
                for (int q = pred_offset[i]; q < pred_offset[i + 1]; ++q)
                    if (pred_index[q] >= 0 && pred_index[q] < n) {
                        if (pred_index[q] == i) return fail(PDMPC_ERR_INVALID, "a vehicle is its own predecessor");
                        succ_off[pred_index[q] + 1] += 1;
                        indeg[i] += 1;
                    }
            for (int i = 0; i < n; ++i) succ_off[i + 1] += succ_off[i];
            succ.resize((size_t)succ_off[n]);
            std::vector<int32_t> fill(succ_off.begin(), succ_off.end() - 1);
            for (int i = 0; i < n; ++i)
                for (int q = pred_offset[i]; q < pred_offset[i + 1]; ++q)
                    if (pred_index[q] >= 0 && pred_index[q] < n) succ[(size_t)fill[pred_index[q]]++] = i;
            for (int i = 0; i < n; ++i)
                if (indeg[i] == 0) {
                    level[i] = 1;
                    queue.push_back(i);
                }
            for (size_t qi = 0; qi < queue.size(); ++qi) {
                const int u = queue[qi];
                for (int q = succ_off[u]; q < succ_off[u + 1]; ++q) {
                    const int w = succ[(size_t)q];
                    level[w] = std::max(level[w], level[u] + 1);
                    if (--indeg[w] == 0) queue.push_back(w);
                }
            }
            if ((int)queue.size() != n) return fail(PDMPC_ERR_INVALID, "the sequential coupling graph has a cycle");
            B.perm.resize((size_t)n);
            for (int i = 0; i < n; ++i) B.perm[(size_t)i] = i;
            std::stable_sort(B.perm.begin(), B.perm.end(), [&](int32_t x, int32_t y) { return level[x] < level[y]; });
            B.inv.resize((size_t)n);
            for (int sl = 0; sl < n; ++sl) B.inv[(size_t)B.perm[(size_t)sl]] = sl;
        }
    }
    const bool permuted = !B.perm.empty();
    for (int slot_i = 0; slot_i < n; ++slot_i) {
        const int i = slot_i;  // (slot: index into the packed arrays)
        const int vi = permuted ? B.perm[(size_t)slot_i] : slot_i;  // (the caller's vehicle)
        const pdmpc_vehicle_in& v = in[vi];
        DevVehicle& d = veh[(size_t)i];
        std::memset(&d, 0, sizeof d);
        if (!v.ref_x || !v.ref_y || !v.v_ref) return fail(PDMPC_ERR_INVALID, "reference trajectory missing");
        if (v.trim0 < 1 || v.trim0 > h->n_trims) return fail(PDMPC_ERR_INVALID, "trim0 out of range");
        int rc;
        if ((rc = check_set(v.obstacles, "obstacles"))) return rc;
        if ((rc = check_set(v.dynamic_obstacles, "dynamic_obstacles"))) return rc;
        if ((rc = check_set(v.hdv_reachable_sets, "hdv_reachable_sets"))) return rc;
        if (v.dynamic_obstacles.n_polygons % Hp) return fail(PDMPC_ERR_INVALID, "dynamic_obstacles must hold n_d * Hp polygons");
        if (v.hdv_reachable_sets.n_polygons % Hp) return fail(PDMPC_ERR_INVALID, "hdv_reachable_sets must hold n_h * Hp polygons");
        if (v.n_left < 0 || v.n_right < 0 || v.n_left == 1 || v.n_right == 1)
            return fail(PDMPC_ERR_INVALID, "lanelet boundary needs 0 or >= 2 points per side");
        d.x0 = v.x0;
        d.y0 = v.y0;
        d.yaw0 = v.yaw0;
        d.trim0 = v.trim0;
        for (int k = 0; k < Hp; ++k) {
            d.ref_x[k] = v.ref_x[k];
            d.ref_y[k] = v.ref_y[k];
            d.v_ref[k] = v.v_ref[k];
        }
        const int n_pred = pred_offset ? pred_offset[vi + 1] - pred_offset[vi] : 0;
        d.n_pred = n_pred;
        d.pred_off = (int32_t)pred.size();
        for (int q = 0; q < n_pred; ++q) {
            const int ps = pred_index[pred_offset[vi] + q];
            if (ps < 0 || ps >= h->max_vehicles) return fail(PDMPC_ERR_INVALID, "predecessor slot out of range");
            pred.push_back(permuted && ps < n ? B.inv[(size_t)ps] : ps);
        }
        const int n_dyn = v.dynamic_obstacles.n_polygons / Hp;
        const int n_hdv = v.hdv_reachable_sets.n_polygons / Hp;
        auto append_poly = [&](const pdmpc_polygon_set& s, int p, bool sep) {
            for (int q = s.offset[p]; q < s.offset[p + 1]; ++q) push_pt(pts, s.x[q], s.y[q]);
            if (sep) push_pt(pts, qnan, qnan);
        };
        int need = 0;
        // vehicle_obstacles{k} = [static..., dynamic(:, k)...], each followed by [NaN; NaN]   vectorize_all_obstacles.m:36-62
        for (int k = 0; k < Hp; ++k) {
            d.lit_off[k] = (int32_t)(pts.size() / 2);
            for (int p = 0; p < v.obstacles.n_polygons; ++p) append_poly(v.obstacles, p, true);
            for (int r = 0; r < n_dyn; ++r) append_poly(v.dynamic_obstacles, r * Hp + k, true);
            need += (int)(pts.size() / 2) - d.lit_off[k] + n_pred * PDMPC_VMAX;
        }
        d.lit_off[Hp] = (int32_t)(pts.size() / 2);
        B.lit_cols[i] = d.lit_off[Hp] - d.lit_off[0];
        for (int k = 0; k < Hp; ++k) {
            d.hdv_off[k] = (int32_t)(pts.size() / 2);
            for (int r = 0; r < n_hdv; ++r) append_poly(v.hdv_reachable_sets, r * Hp + k, true);
        }
        d.hdv_off[Hp] = (int32_t)(pts.size() / 2);
        need += d.hdv_off[Hp] - d.hdv_off[0];
        // lanelet_boundary = [left, NaN, right, NaN]                                          vectorize_all_obstacles.m:27-30
        d.ll_off = (int32_t)(pts.size() / 2);
        for (int q = 0; q < v.n_left; ++q) push_pt(pts, v.left_x[q], v.left_y[q]);
        push_pt(pts, qnan, qnan);
        for (int q = 0; q < v.n_right; ++q) push_pt(pts, v.right_x[q], v.right_y[q]);
        push_pt(pts, qnan, qnan);
        d.ll_len = (int32_t)(pts.size() / 2) - d.ll_off;
        need += d.ll_len;
        B.lit_cols[i] += d.ll_len;
        if (fallback && fallback[vi].n_polygons > 0) {
            if (fallback[vi].n_polygons != Hp) return fail(PDMPC_ERR_INVALID, "fallback_shapes must hold Hp polygons per vehicle");
            int rc2;
            if ((rc2 = check_set(fallback[vi], "fallback_shapes"))) return rc2;
            for (int k = 0; k < Hp; ++k) {
                d.fb_off[k] = (int32_t)(pts.size() / 2);
                if (fallback[vi].offset[k + 1] - fallback[vi].offset[k] > PDMPC_VMAX)
                    return fail(PDMPC_ERR_INVALID, "fallback area has more than PDMPC_VMAX columns");
                append_poly(fallback[vi], k, false);
            }
            d.fb_off[Hp] = (int32_t)(pts.size() / 2);
        } else {
            for (int k = 0; k <= Hp; ++k) d.fb_off[k] = -1;
        }
        soup_cap = std::max(soup_cap, need);
        for (int k = 0; k < Hp; ++k)
            cand_cap = std::max(cand_cap, (d.lit_off[k + 1] - d.lit_off[k]) + n_pred * PDMPC_VMAX + (d.hdv_off[k + 1] - d.hdv_off[k]) + d.ll_len);
    }
    // a trailing pad so 16-byte staged copies never run past the allocation
    push_pt(pts, qnan, qnan);
    pred.push_back(0);
    B.soup_cap = soup_cap + 2;
    B.cand_cap = (cand_cap + 4 + 3) & ~3;
    const size_t veh_bytes = ((size_t)std::max(n, 1) * sizeof(DevVehicle) + 15) & ~(size_t)15;
    const size_t pts_bytes = (pts.size() * sizeof(double) + 15) & ~(size_t)15;
    const size_t pred_bytes = (pred.size() * sizeof(int32_t) + 15) & ~(size_t)15;
    const size_t total = veh_bytes + pts_bytes + pred_bytes;
    if (B.h_blob.ensure(total)) return fail(PDMPC_ERR_HIP, "hipHostMalloc failed");
    if (B.d_blob.ensure(total)) return fail(PDMPC_ERR_HIP, "hipMalloc failed for the batch blob");
    B.h_veh = (DevVehicle*)B.h_blob.p;
    B.h_pts = (double*)(B.h_blob.p + veh_bytes);
    B.h_pred = (int32_t*)(B.h_blob.p + veh_bytes + pts_bytes);
    B.d_veh = (DevVehicle*)B.d_blob.p;
    B.d_pts = (double*)(B.d_blob.p + veh_bytes);
    B.d_pred = (int32_t*)(B.d_blob.p + veh_bytes + pts_bytes);
    std::memcpy(B.h_veh, veh.data(), (size_t)std::max(n, 1) * sizeof(DevVehicle));
    std::memcpy(B.h_pts, pts.data(), pts.size() * sizeof(double));
    std::memcpy(B.h_pred, pred.data(), pred.size() * sizeof(int32_t));
    // one copy, not waited for: whatever the stream does next is ordered behind it, and the next pack into this bank waits (above)
    HIPCHK(hipMemcpyAsync(B.d_blob.p, B.h_blob.p, total, hipMemcpyHostToDevice, h->stream));
    B.staged_serial = h->sync_serial;
    B.n_packed = n;
    h->events_used = 0;
    std::memset(&h->stats, 0, sizeof h->stats);
    return PDMPC_OK;
}

// per-vehicle arenas for `nodes` tree nodes each (contents are scratch: every search starts from an empty tree)
int alloc_arenas(pdmpc_handle* h, uint32_t nodes) {
    nodes = (nodes + 1u) & ~1u;
    const size_t tot = (size_t)h->max_vehicles * nodes;
    h->anodes.release();
    h->ahk.release();
    h->ahid.release();
    h->avs.release();
    h->alog.release();
    h->ankey.release();
    h->anid.release();
    h->amidk.release();
    h->amidi.release();
    h->achild0.release();
    h->awalk.release();
    h->alink.release();
    h->max_nodes = 0;
    int bad = 0;
    bad |= h->anodes.ensure_exact(tot) | h->ahk.ensure_exact(tot) | h->ahid.ensure_exact(tot) | h->avs.ensure_exact(tot) | h->alog.ensure_exact(tot);
    bad |= h->ankey.ensure_exact(tot) | h->anid.ensure_exact(tot) | h->alink.ensure_exact(tot) | h->amidk.ensure_exact(tot) | h->amidi.ensure_exact(tot) | h->achild0.ensure_exact(tot) | h->awalk.ensure_exact(2 * tot);
    if (bad) return bad;
    h->max_nodes = nodes;
    return 0;
}

// safe == true: the recovery path after a predecessor time-out (plan_packed_growing): slices that are resident as a whole,
// no helper workgroups next to an oversubscribed launch, the default spin limit.
int launch_range(pdmpc_handle* h, int first, int count, bool safe = false) {
    if (!h) return fail(PDMPC_ERR_INVALID, "null handle");
    PackedStep& B = h->banks[h->bank];
    if (first < 0 || count < 0 || first + count > B.n_packed) return fail(PDMPC_ERR_INVALID, "launch range outside the packed batch");
    if (!B.perm.empty() && (first != 0 || count != B.n_packed)) return fail(PDMPC_ERR_INVALID, "range launches need a batch packed in level order (predecessors in lower slots)");
    if (count == 0) return PDMPC_OK;
    const Tuning& T = h->tune;
    int rc = compute_lds(h, count, B.soup_cap, B.cand_cap);
    if (rc) return rc;
    // the sampled optimizer keeps its tree (288 nodes x (16 children + parent + trim) x 2 B) where the open list would be
    if (h->sampled_launch && h->lds.total - h->lds.heap_key < 288u * 18u * 2u) return fail(PDMPC_ERR_CAPACITY, "not enough LDS left for the sampled optimizer's tree");
    KernelArgs a{};
    a.succ_mask = h->d_mask.p;
    a.man_index = h->d_mi.p;
    a.man_pose = h->d_pose.p;
    a.man_area = h->d_area.p;
    a.n_trims = h->n_trims;
    a.n_words = h->n_words;
    a.n_man = h->n_man;
    a.Hp = h->cfg.Hp;
    a.checker = h->cfg.checker;
    a.areas_in_lds = h->areas_in_lds;
    a.dt = h->cfg.dt_seconds;
    a.veh = B.d_veh;
    a.points = B.d_pts;
    a.pred = B.d_pred;
    a.out = h->d_out.p;
    a.done_flag = h->d_flag.p;
    a.epoch = h->epoch;
    a.first = first;
    a.arena.nodes = h->anodes.p;
    a.arena.heap_key = h->ahk.p;
    a.arena.heap_id = h->ahid.p;
    a.arena.pop_log = h->alog.p;
    a.arena.vstate = h->avs.p;
    a.arena.near_key = h->ankey.p;
    a.arena.near_id = h->anid.p;
    a.arena.walk = h->awalk.p;
    a.arena.mid_key = h->amidk.p;
    a.arena.mid_id = h->amidi.p;
    a.arena.child0 = h->achild0.p;
    a.arena.link = h->alink.p;
    a.max_nodes = h->max_nodes;
    a.pop_trace = h->d_trace.p;
    a.trace_cap = h->cfg.trace_pops;
    a.tree_size = h->d_tree_size.p;
    a.lds = h->lds;
    a.HL = h->HL;
    a.NL = h->NL;
    a.NV = h->NV;
    const bool frontier = h->kernel_frontier && !h->sampled_launch;
    const bool bulk = use_bulk(h);
    a.bulk = bulk ? 1 : 0;
    a.bk_ready_cap = std::min(h->bk_ready_launch, 3 * PDMPC_WAVE * h->n_waves);  // (the verdict pass handles three entries per thread)
    a.bk_round0 = std::max(1, T.bk_round0);
    {
        // measured on C2 / C3 (20 / 128 searches, helpers): cap 256, ramp 4 -> 646 / 589 steps/s; 512, 2 -> 735 / 786; 1000, 2 -> 769 / 909; 1000, 1 -> 620 / 772
        const bool helped = bulk && bulk_has_helpers(h, count);
        const int cap = T.bk_round > 0 ? T.bk_round : (helped ? 1000 : 256);
        a.bk_round = std::min(h->bk_ready_launch / 2 - 16, std::max(a.bk_round0, cap));
    }
    a.soup_cap = B.soup_cap;
    a.cand_cap = frontier ? h->fr_cand_cap : B.cand_cap;
    a.frontier = frontier ? 1 : 0;
    a.fr_round = h->fr_round > 0 ? h->fr_round : 768;  // cap of a round; measured on C2 / C3 (with the early-exit InterX): 256 -> 342 / 322 steps/s, 512 -> 355 / 342, 768 -> 358 / 345, 1024 -> 358 / 345
    a.fr_stage_cap = h->fr_stage_cap;
    if (bulk) a.fr_ramp = T.bk_ramp > 0 ? T.bk_ramp : (bulk_has_helpers(h, count) ? 2 : 4);
    else a.fr_ramp = T.fr_ramp > 0 ? T.fr_ramp : 4;  // a round grows by a quarter of the nodes done so far; measured on C2 / C3 (cap 768): 2 -> 331 / 354 steps/s, 3 -> 354 / 352, 4 -> 358 / 345, 6 -> 355 / 338
    a.fr_near_fill = h->fr_near_fill;
    a.fr_near_max = h->fr_near_max;
    a.fr_join_scale = T.fr_join_scale;  // measured on C2 / C3 / C5: 1 -> 419 / 396 / 347 steps/s, 4 -> 434 / 396 / 353, 8 -> 434 / 397 / 345, 32 -> 429 / 394 / 344
    a.fr_dive = T.fr_dive;  // in rounds of up to this many entries a wave goes on with the best child while its key stays within the round's range (frontier_kernel.hip, fr_process); measured on C2 / C3 / C5: 0 -> 408 / 391 / 307 steps/s, 64 -> 409 / 391 / 314, 1024 -> 419 / 397 / 323
    a.fr_root_dive = T.fr_root_dive;
    a.spin_limit = safe ? (1u << 22) : T.spin_limit;
    a.debug_tail = T.debug_tail;
    if (T.debug_progress && !h->progress) {
        if (hipHostMalloc((void**)&h->progress, (size_t)h->max_vehicles * 64 * 4, hipHostMallocMapped) != hipSuccess) h->progress = nullptr;
        if (h->progress) std::memset(h->progress, 0, (size_t)h->max_vehicles * 64 * 4);
    }
    a.progress = h->progress;
    a.speculate = h->speculate;
    a.crowded = count > h->n_cu ? 1 : 0;
    a.dense = T.dense >= 0 ? T.dense : h->two_per_cu;
    a.speculate_expansion = h->speculate_expansion;
    a.n_validators = h->n_validators;
    a.n_waves = h->n_waves;
    a.queue_mode = (h->queue_mode == PDMPC_QUEUE_BLOCKMIN && h->bm_kr != 0) ? PDMPC_QUEUE_BLOCKMIN : PDMPC_QUEUE_HEAP;
    a.bm_kr = h->bm_kr;
    a.bm_nb = h->bm_nb;
    // dropping hides pops from the pop trace: off while tracing (PDMPC_DROP=2 forces it, for tests that only compare records)
    a.drop_invalid = h->cfg.trace_pops > 0 ? 0 : 1;
    if (T.drop >= 0) a.drop_invalid = T.drop == 2 ? 1 : (T.drop == 0 ? 0 : a.drop_invalid);
    a.eager_validation = T.eager >= 0 ? T.eager : a.drop_invalid;
    a.drop_beyond_lds = T.drop_beyond_lds;
    a.tie_count = h->d_tie_count.p;
    a.work_count = h->d_work_count.p;
    a.sampled_random = h->d_random.p;
    a.sampled_n_random = h->sampled_n_random;
    a.reverse_dispatch = (!safe && T.slot_order_reverse) ? 1 : 0;
    if (h->events_used == h->events.size()) {
        hipEvent_t e0, e1;
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        h->events.emplace_back(e0, e1);
    }
    // Oversubscribed launches (more searches than CUs at one 16-wave workgroup per CU).  A resident search spins for
    // predecessors of the same launch; slots are in level order (pack_common sees to it), so as long as the hardware hands out
    // workgroups in index order every predecessor was dispatched before its successors and the launch cannot stall.  That order
    // is not a documented guarantee: should a launch ever stall, the watchdog (spin_limit) ends the waiting searches with an
    // error status and plan_packed_growing plans the call again with safe == true, in slices that are resident as a whole (a
    // slice's predecessors are in it or in an earlier slice) -- forward progress then needs no assumption at all.
    // PDMPC_FR_SLICE=1 slices always (measured on C4: 9.8 steps/s in one launch against 7.4 in slices), =0 never.
    const bool oversub = frontier && !h->two_per_cu && count > h->n_cu;
    const bool slice = oversub && (T.fr_slice == 1 || (safe && T.fr_slice != 0));
    // helper workgroups on the CUs this launch leaves idle (frontier kernel, InterX).  A helper spins until every search of
    // the launch has published, so in the safe mode an oversubscribed launch gets none (they would hold CUs a slice counts on).
    a.n_searches = count;
    a.n_helpers = 0;
    a.fr_share_min = T.fr_share_min;
    a.fr_own_div = T.fr_own_div;
    a.help_chunk = T.help_chunk;  // (0: chosen below, once it is known whether the helpers expand)
    if (frontier && !h->sampled_launch && h->cfg.checker == PDMPC_CHECK_INTERX && h->speculate && !slice) {
        // (measured on C2, 20 searches, / C3, 128: 8 helpers 535 / 476 steps/s, 16: 529 / 506, 32: 516 / 522, 128: 507 / 483)
        int want = std::min(h->helpers_max, std::max(32, count / 2));  // (helpers that expand: the owner of a shared round waits for them, more of them with shorter runs finish sooner)
        if (T.helpers >= 0) want = T.helpers;  // A/B switch (0: none): results are identical
        a.n_helpers = std::max(0, std::min(want, h->n_cu - count));
        if (a.n_helpers < 2) a.n_helpers = 0;
        if (count > h->n_cu) {
            // More searches than CUs: the helpers are there for the tail of the launch, when CUs fall idle while a few long searches
            // still run; until then they cost the CUs they sit on (their stream has the lowest priority, a search that is waiting
            // for a CU gets it first).  Measured on C4 (512 searches) / C5 (1280): none 25.6 / 352 steps/s, 16 helpers 38.5 / 353,
            // 32: 41.5 / 352, 64: 43.4 / 332, 96: 42.8 / 299; with helpers that also expand (C4 only, see below) 64: 43.6, 96: 45.9, 128: 46.4.
            a.n_helpers = count <= 2 * h->n_cu ? 96 : 32;
            if (T.helpers_oversub >= 0) a.n_helpers = std::min(T.helpers_oversub, h->n_cu / 2);
            if (T.helpers >= 0) a.n_helpers = std::min(a.n_helpers, T.helpers);  // (0 switches every helper off)
            if (bulk && count > 2 * h->n_cu) a.n_helpers = 0;  // (five searches per CU: a helper only takes a CU away from a search)
        }
    }
    if (safe) a.n_helpers = 0;  // the recovery path counts on nothing but slot order: no helper workgroup sits where a search could run
    a.bk_share_min = T.bk_share_min;
    a.bk_mid_min = T.bk_mid_min;
    a.bk_mid_fill = T.bk_mid_fill;
    // a tile costs a helper ≈ 10 us besides its checks (claim, acquire, records, verdicts, release, report): launches with many searches,
    // whose helpers hop between boards, do better with larger tiles (measured C2 / C3 / C4: 64 -> 1 049 / 1 015 / 70.4 steps/s, 96 -> 1 032 / 1 023 / 71.2,
    // 128 -> 1 028 / 1 028 / 71.6; 32 -> 951 / 833 / -)
    a.bk_tile = T.bk_tile > 0 ? T.bk_tile : (count >= 64 ? 128 : 64);
    a.bk_tentative = T.bk_tentative;
    a.bk_fast_arrival = T.bk_fast_arrival;
    a.bk_force_tie = T.bk_force_tie;
    a.bk_post = h->d_bk_post.p;
    a.help_board = h->d_help_board.p;
    a.help_list = h->d_help_list.p;
    a.help_verdict = h->d_help_verdict.p;
    a.help_cs = h->d_help_cs.p;
    a.help_expand = T.help_expand;  // A/B switch: results are identical
    a.help_patience = T.help_patience;
    // with more searches than CUs helpers are scarce and an owner that waits for them loses (C5, 5 searches per CU: 332 against 355
    // steps/s); up to two searches per CU the tail of the launch is long enough for expanding helpers to pay (C4: 42.1 -> 46)
    const bool expand_oversub = T.help_expand_oversub >= 0 ? T.help_expand_oversub != 0 : count <= 2 * h->n_cu;
    if (h->n_words != 1 || h->fr_stage_cap < 128 || (count > h->n_cu && !expand_oversub)) a.help_expand = 0;  // (the helper kernel expands one successor-mask word per node; a run's records sit in its staging area)
    if (!bulk && a.help_expand && a.n_helpers > 0 && count <= h->n_cu) {
        // helpers take the bulk of a large round off the owner, so rounds may grow faster and larger (measured on C2 / C3 with expanding
        // helpers: ramp 4, cap 768 -> 629 / 616 steps/s; 3, 768 -> 673 / 651; 2, 768 -> 680 / 662; 2, 1024 -> 686 / 656; 1, 1024 -> 632 / 604)
        if (T.fr_ramp <= 0) a.fr_ramp = 2;
        if (h->fr_round <= 0) a.fr_round = 1280;  // (ready list: 1536 entries; measured on C2 / C3: 1024 -> 734 / 665 steps/s, 1280 -> 747 / 675, 1536 -> 749 / 667)
    }
    if (a.help_chunk == 0) a.help_chunk = a.help_expand ? 32 : 64;  // measured on C2 / C3: expanding helpers 64 -> 555 / 595 steps/s, 32 -> 595 / 584; checking only: 64 best (C4 43.3 against 40.9)
    a.help_finished = h->d_help_finished.p;
    a.help_fin_base = 0;
    if (a.n_helpers > 0 && bulk && !h->boards_dirty) {
        // The bulk kernel leaves its boards closed (a search closes every round it shares before it uses the verdicts, and a closed
        // ticket word offers nothing) and counts finished searches on from launch to launch: nothing to clear between launches --
        // two memset dispatches less per launch.  Anything else that touched the boards (the frontier kernel's helpers, a launch
        // that ended with a watchdog status) marks them dirty and the next launch clears them as before.
        a.help_fin_base = h->help_fin_total;
        h->help_fin_total += (uint32_t)count;
    } else if (a.n_helpers > 0) {
        HIPCHK(hipMemsetAsync(h->d_help_board.p, 0, (size_t)h->max_vehicles * PDMPC_HB_WORDS * sizeof(unsigned long long), h->stream));
        HIPCHK(hipMemsetAsync(h->d_help_finished.p, 0, 16 * sizeof(uint32_t), h->stream));
        h->help_fin_total = bulk ? (uint32_t)count : 0u;
        h->boards_dirty = !bulk;
        if (!bulk) HIPCHK(hipEventRecord(h->ev_help_pre, h->stream));  // the boards are clean (the bulk kernel's helpers are workgroups of the same launch: nothing to order)
        if (!bulk && T.help_first) {  // diagnostic: the old order, helpers in front of the searches
            hipStream_t hst0 = count > h->n_cu ? h->help_stream_low : h->help_stream;
            HIPCHK(hipStreamWaitEvent(hst0, h->ev_help_pre, 0));
            const int hrc0 = pdmpc_launch_helpers(&a, (void*)hst0, h->frontier_lds_hw);
            if (hrc0 != 0) return fail(PDMPC_ERR_HIP, "helper kernel launch failed");
            HIPCHK(hipEventRecord(h->ev_help_done, hst0));
        }
    }
    auto& ev = h->events[h->events_used++];
    HIPCHK(hipEventRecord(ev.first, h->stream));
    h->last_launch_frontier = frontier;
    h->last_launch_bulk = bulk;
    h->last_first = first;
    h->last_count = count;
    if (h->step_ranges_epoch != h->epoch) {
        h->step_ranges.clear();
        h->step_ranges_epoch = h->epoch;
    }
    if (bulk) h->step_ranges.emplace_back(first, count);
    auto launch_round_based = [&](const KernelArgs* ka, int cnt) {
        return bulk ? pdmpc_launch_bulk(ka, cnt, (void*)h->stream, h->bulk_lds_hw) : pdmpc_launch_frontier(ka, cnt, (void*)h->stream, h->frontier_lds_hw);
    };
    int lrc = 0;
    if (slice) {
        for (int done = 0; done < count && lrc == 0; done += h->n_cu) {
            KernelArgs part = a;
            part.first = first + done;
            part.n_searches = std::min(h->n_cu, count - done);
            lrc = launch_round_based(&part, part.n_searches);
        }
    } else {
        lrc = h->sampled_launch ? pdmpc_launch_sampled(&a, count, (void*)h->stream)
                                : (frontier ? launch_round_based(&a, count) : pdmpc_launch_search(&a, count, (void*)h->stream));
    }
    if (lrc != 0) {
        h->boards_dirty = true;  // (the searches that were to count themselves finished never ran: the next launch starts from cleared counters)
        char buf[256];
        snprintf(buf, sizeof buf, "kernel launch failed: %s (LDS %u B)", hipGetErrorString((hipError_t)lrc), h->lds.total);
        return fail(PDMPC_ERR_HIP, buf);
    }
    HIPCHK(hipEventRecord(ev.second, h->stream));
    if (bulk) {
        // (helpers are the trailing workgroups of the search launch itself)
    } else if (a.n_helpers > 0 && T.help_first) {
        HIPCHK(hipStreamWaitEvent(h->stream, h->ev_help_done, 0));
    } else if (a.n_helpers > 0) {
        // The helper kernel goes out BEHIND the searches, on a stream of its own: it starts once the boards are clean and runs next
        // to the searches.  The order matters where the runtime maps both streams onto one hardware queue (few queues, many
        // streams: seen with RCCL initialised in the process): a helper spins until every search has published, so helpers in front
        // of the searches on a shared queue would hold them up until the helpers' idle time-out (0.7 s per launch); behind them
        // they find every search finished and leave at once -- the launch then simply ran without helpers.  Searches never wait for
        // a helper that has not claimed anything.  Everything the launch stream does after the searches also waits for the helpers
        // to have left (they leave as soon as the last search has published).
        hipStream_t hst = count > h->n_cu ? h->help_stream_low : h->help_stream;
        HIPCHK(hipStreamWaitEvent(hst, h->ev_help_pre, 0));
        const int hrc = pdmpc_launch_helpers(&a, (void*)hst, h->frontier_lds_hw);
        if (hrc != 0) return fail(PDMPC_ERR_HIP, std::string("helper kernel launch failed: ") + hipGetErrorString((hipError_t)hrc));
        HIPCHK(hipEventRecord(h->ev_help_done, hst));
        HIPCHK(hipStreamWaitEvent(h->stream, h->ev_help_done, 0));
    }
    h->stats.lds_bytes = h->lds.total;
    h->stats.lds_nodes = h->NL;
    h->stats.queue_mode = a.queue_mode;
    h->stats.queue_ring_entries = h->bm_kr;
    return PDMPC_OK;
}

}  // namespace

extern "C" {

const char* pdmpc_last_error(void) { return g_err.c_str(); }
const char* pdmpc_version(void) { return "pdmpc-hip 0.1 (gfx950)"; }

int pdmpc_create(const pdmpc_config* config, pdmpc_handle** out_handle) {
    if (!config || !out_handle) return fail(PDMPC_ERR_INVALID, "null argument");
    if (config->Hp < 1 || config->Hp > PDMPC_HP_MAX) return fail(PDMPC_ERR_INVALID, "Hp must be in 1..PDMPC_HP_MAX");
    if (config->checker != PDMPC_CHECK_SAT && config->checker != PDMPC_CHECK_INTERX) return fail(PDMPC_ERR_INVALID, "unknown checker");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(PDMPC_ERR_NO_DEVICE, "no HIP device visible: this backend has no CPU fallback");
    if (config->device < 0 || config->device >= ndev) return fail(PDMPC_ERR_NO_DEVICE, "device ordinal out of range");
    HIPCHK(hipSetDevice(config->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, config->device));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        return fail(PDMPC_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    pdmpc_handle* h = new pdmpc_handle();
    h->cfg = *config;
    h->banks.resize(1);
    {
        // every environment switch is read here, once (A/B switches for benchmarking and tuning knobs; results are identical)
        auto env_i = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
        Tuning& T = h->tune;
        h->speculate = env_i("PDMPC_SPECULATE", 1) != 0;
        if (getenv("PDMPC_WAVES")) h->waves_latency = h->waves_crowded = std::min(PDMPC_MAX_WAVES, std::max(4, env_i("PDMPC_WAVES", 16)));
        if (getenv("PDMPC_VALIDATORS")) h->n_validators = std::max(1, env_i("PDMPC_VALIDATORS", 1));
        h->speculate_expansion = env_i("PDMPC_SPEC_EXPAND", 1) != 0;
        if (const char* e = getenv("PDMPC_KERNEL")) {  // A/B switch: bulk (default) | frontier | serial
            h->kernel_frontier = std::string(e) != "serial";
            h->kernel_bulk = std::string(e) != "frontier";
        }
        T.bk_round0 = std::max(1, env_i("PDMPC_BK_ROUND0", T.bk_round0));
        if (getenv("PDMPC_BK_ROUND")) T.bk_round = std::max(1, env_i("PDMPC_BK_ROUND", 256));
        if (getenv("PDMPC_BK_RAMP")) T.bk_ramp = std::max(1, env_i("PDMPC_BK_RAMP", 4));
        T.bk_share_min = std::max(64, env_i("PDMPC_BK_SHARE_MIN", T.bk_share_min));
        T.bk_mid_min = std::max(0, env_i("PDMPC_BK_MID_MIN", T.bk_mid_min));
        T.bk_mid_fill = std::max(256, env_i("PDMPC_BK_MID_FILL", T.bk_mid_fill));
        T.bk_tentative = env_i("PDMPC_BK_TENTATIVE", T.bk_tentative) != 0;
        T.bk_fast_arrival = env_i("PDMPC_BK_FAST_ARRIVAL", T.bk_fast_arrival) != 0;
        T.bk_force_tie = env_i("PDMPC_BK_FORCE_TIE", T.bk_force_tie) != 0;
        if (getenv("PDMPC_BK_TILE")) T.bk_tile = std::min(128, std::max(16, env_i("PDMPC_BK_TILE", 64)));
        h->bk_ready_cap = std::min(2048, std::max(256, env_i("PDMPC_BK_READY", h->bk_ready_cap))) & ~63;  // (the most a launch may use: launches without helpers lay out half of it)
        if (getenv("PDMPC_FR_ROUND")) h->fr_round = std::max(1, env_i("PDMPC_FR_ROUND", 0));
        if (getenv("PDMPC_FR_NEAR_FILL")) h->fr_near_fill = std::max(64, env_i("PDMPC_FR_NEAR_FILL", 0));
        if (getenv("PDMPC_FR_NEAR_MAX")) h->fr_near_max = std::max(256, env_i("PDMPC_FR_NEAR_MAX", 0));
        if (getenv("PDMPC_QUEUE")) h->queue_mode = env_i("PDMPC_QUEUE", 1) != 0 ? PDMPC_QUEUE_BLOCKMIN : PDMPC_QUEUE_HEAP;
        if (getenv("PDMPC_FR_STAGE")) T.fr_stage = std::max(0, env_i("PDMPC_FR_STAGE", 0));
        T.fr_two_per_cu = env_i("PDMPC_FR_TWO_PER_CU", 0) != 0;
        T.bk_two_per_cu = env_i("PDMPC_BK_TWO_PER_CU", T.bk_two_per_cu) != 0;
        T.debug_lds = getenv("PDMPC_DEBUG_LDS") != nullptr;
        T.hl_max = env_i("PDMPC_HL_MAX", T.hl_max);
        T.bm_ring = env_i("PDMPC_BM_RING", T.bm_ring);
        T.nv_max = env_i("PDMPC_NV_MAX", T.nv_max);
        if (getenv("PDMPC_FR_RAMP")) T.fr_ramp = std::max(1, env_i("PDMPC_FR_RAMP", 4));
        if (const char* e = getenv("PDMPC_FR_JOIN_SCALE")) T.fr_join_scale = atof(e);
        T.fr_root_dive = env_i("PDMPC_FR_ROOT_DIVE", 0) != 0;
        T.fr_dive = std::max(0, env_i("PDMPC_FR_DIVE", T.fr_dive));
        if (getenv("PDMPC_SPIN_LIMIT")) T.spin_limit = (uint32_t)std::max(1024, env_i("PDMPC_SPIN_LIMIT", 0));  // debugging: fail fast
        T.debug_tail = env_i("PDMPC_DEBUG_TAIL", 0);
        T.debug_progress = getenv("PDMPC_DEBUG_PROGRESS") != nullptr;
        if (getenv("PDMPC_DENSE")) T.dense = env_i("PDMPC_DENSE", 0) != 0;
        T.drop = env_i("PDMPC_DROP", -1);
        T.drop_beyond_lds = env_i("PDMPC_DROP_BEYOND_LDS", 1) != 0;
        if (getenv("PDMPC_EAGER")) T.eager = env_i("PDMPC_EAGER", 0) != 0;
        T.fr_share_min = std::max(64, env_i("PDMPC_FR_SHARE_MIN", T.fr_share_min));
        T.fr_own_div = std::max(1, env_i("PDMPC_FR_OWN_DIV", T.fr_own_div));
        if (getenv("PDMPC_HELP_CHUNK")) T.help_chunk = std::min(128, std::max(16, env_i("PDMPC_HELP_CHUNK", 32) / 16 * 16));
        if (getenv("PDMPC_HELPERS")) T.helpers = std::max(0, env_i("PDMPC_HELPERS", 0));
        if (getenv("PDMPC_HELPERS_OVERSUB")) T.helpers_oversub = std::max(0, env_i("PDMPC_HELPERS_OVERSUB", 0));
        T.help_expand = env_i("PDMPC_HELP_EXPAND", 1) != 0;
        T.help_patience = std::max(0, env_i("PDMPC_HELP_PATIENCE", T.help_patience));
        if (getenv("PDMPC_HELP_EXPAND_OVERSUB")) T.help_expand_oversub = env_i("PDMPC_HELP_EXPAND_OVERSUB", 0) != 0;
        if (getenv("PDMPC_FR_SLICE")) T.fr_slice = env_i("PDMPC_FR_SLICE", 0) != 0;
        T.debug_host = getenv("PDMPC_DEBUG_HOST") ? std::max(1, atoi(getenv("PDMPC_DEBUG_HOST"))) : 0;  // 1: a line per launch; 2: the host-time breakdown of the literal path only
        T.help_first = getenv("PDMPC_HELP_FIRST") != nullptr;
        T.help_prio = env_i("PDMPC_HELP_PRIO", 1) != 0;
        T.slot_order_reverse = env_i("PDMPC_TEST_REVERSE_DISPATCH", 0) != 0;
    }
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const uint32_t want_nodes = config->max_nodes > 0 ? (uint32_t)config->max_nodes : 32768u;  // default arena: 256 x 32768 nodes, about 0.6 GB
    h->max_vehicles = config->max_vehicles > 0 ? config->max_vehicles : 256;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    // (the helper stream above the launch stream's priority: streams of different priorities do not share a hardware queue, and a
    // helper only ever occupies CUs the launch leaves idle)
    if (!h->tune.help_prio) prio_greatest = 0;  // A/B switch: the helper stream at the launch stream's priority
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&h->help_stream, hipStreamNonBlocking, prio_greatest);
    if (e == hipSuccess) {
        // Two streams for the helper kernel.  Where searches and helpers compete for CUs (more searches than CUs) it runs at the
        // lowest priority: a search that is waiting for a CU gets it first.  Otherwise at the searches' priority: on the
        // low-priority stream its dispatch was now and then held back until the searches were through (one step in a hundred
        // then ran without helpers, 5 ms instead of 2: seen as p99 outliers in two of four bench runs, in none of four this way).
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        e = hipStreamCreateWithPriority(&h->help_stream_low, hipStreamNonBlocking, prio_lo);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_help_pre, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_help_done, hipEventDisableTiming);
    if (e != hipSuccess) {
        delete h;
        return fail(PDMPC_ERR_HIP, "hipStreamCreate failed");
    }
    int bad = alloc_arenas(h, want_nodes);
    bad |= h->d_out.ensure((size_t)h->max_vehicles) | h->d_flag.ensure((size_t)h->max_vehicles) | h->d_tree_size.ensure((size_t)h->max_vehicles) | h->d_tie_count.ensure(4) | h->d_work_count.ensure(8);
    bad |= h->d_help_board.ensure((size_t)h->max_vehicles * PDMPC_HB_WORDS) | h->d_help_list.ensure((size_t)h->max_vehicles * PDMPC_HELP_CAP) |
           h->d_help_verdict.ensure((size_t)h->max_vehicles * PDMPC_HELP_CAP) | h->d_help_cs.ensure((size_t)h->max_vehicles * PDMPC_HELP_CAP * 2) |  h->d_help_finished.ensure(16);
    bad |= h->d_bk_post.ensure((size_t)h->max_vehicles * (size_t)h->bk_ready_cap * 6);
    bad |= h->d_trace.ensure((size_t)h->max_vehicles * (size_t)std::max(config->trace_pops, 1));
    if (bad) {
        pdmpc_destroy(h);
        return fail(PDMPC_ERR_HIP, "hipMalloc failed for the per-vehicle arenas (lower max_nodes / max_vehicles)");
    }
    (void)hipMemsetAsync(h->d_flag.p, 0, h->d_flag.cap * sizeof(uint32_t), h->stream);
    (void)hipMemsetAsync(h->d_tree_size.p, 0, h->d_tree_size.cap * sizeof(int32_t), h->stream);
    (void)hipMemsetAsync(h->d_tie_count.p, 0, 4 * sizeof(int32_t), h->stream);
    (void)hipMemsetAsync(h->d_work_count.p, 0, 8 * sizeof(unsigned long long), h->stream);
    (void)hipMemsetAsync(h->d_out.p, 0, h->d_out.cap * sizeof(pdmpc_vehicle_out), h->stream);
    (void)hipStreamSynchronize(h->stream);
    *out_handle = h;
    return PDMPC_OK;
}

int pdmpc_destroy(pdmpc_handle* h) {
    if (!h) return PDMPC_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->help_stream) (void)hipStreamSynchronize(h->help_stream);
    if (h->help_stream_low) (void)hipStreamSynchronize(h->help_stream_low);
    for (auto& ev : h->events) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    h->d_mask.release();
    h->d_mi.release();
    h->d_pose.release();
    h->d_area.release();
    h->anodes.release();
    h->ahk.release();
    h->ahid.release();
    h->alog.release();
    h->ankey.release();
    h->anid.release();
    h->amidk.release();
    h->amidi.release();
    h->achild0.release();
    h->awalk.release();
    h->alink.release();
    h->avs.release();
    h->d_out.release();
    h->h_out.release();
    h->d_flag.release();
    h->d_tree_size.release();
    h->d_tie_count.release();
    h->d_work_count.release();
    h->d_help_board.release();
    h->d_help_list.release();
    h->d_help_verdict.release();
    h->d_help_cs.release();
    h->d_bk_post.release();
    h->d_help_finished.release();
    h->d_random.release();
    h->d_trace.release();
    for (auto& b : h->banks) b.release();
    if (h->ev_help_pre) (void)hipEventDestroy(h->ev_help_pre);
    if (h->ev_help_done) (void)hipEventDestroy(h->ev_help_done);
    if (h->help_stream) (void)hipStreamDestroy(h->help_stream);
    if (h->help_stream_low) (void)hipStreamDestroy(h->help_stream_low);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PDMPC_OK;
}

int pdmpc_upload_mpa(pdmpc_handle* h, const pdmpc_mpa* mpa) {
    if (!h || !mpa) return fail(PDMPC_ERR_INVALID, "null argument");
    if (mpa->n_trims < 1 || mpa->n_trims > 1023) return fail(PDMPC_ERR_INVALID, "n_trims must be in 1..1023");
    if (mpa->Hp < h->cfg.Hp) return fail(PDMPC_ERR_INVALID, "mpa.Hp smaller than config.Hp");
    if (!mpa->transition || !mpa->maneuver_index || (mpa->n_maneuvers > 0 && !mpa->maneuvers)) return fail(PDMPC_ERR_INVALID, "null table");
    HIPCHK(hipSetDevice(h->cfg.device));
    const int n = mpa->n_trims, Hp = h->cfg.Hp;
    const int nw = (n + 63) / 64;
    std::vector<uint64_t> mask((size_t)Hp * n * nw + 2, 0);
    for (int k = 0; k < Hp; ++k)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                if (mpa->transition[((size_t)k * n + i) * n + j]) {
                    const int mi = mpa->maneuver_index[i * n + j];
                    if (mi < 0 || mi >= mpa->n_maneuvers) return fail(PDMPC_ERR_INVALID, "transition allowed but maneuver missing");
                    mask[((size_t)k * n + i) * nw + j / 64] |= 1ull << (j % 64);
                }
    for (size_t q = 0; q < mask.size(); ++q)
        if (__builtin_popcountll(mask[q]) > 16) return fail(PDMPC_ERR_CAPACITY, "a trim has more than 16 successors within one 64-trim word");
    std::vector<int16_t> mi((size_t)n * n + 8, -1);
    for (int i = 0; i < n * n; ++i) mi[i] = (int16_t)mpa->maneuver_index[i];
    const int T = mpa->n_maneuvers;
    std::vector<DevManPose> pose((size_t)std::max(T, 1));
    std::vector<double> area((size_t)std::max(T, 1) * 3 * PDMPC_VMAX * 2, 0.0);
    for (int t = 0; t < T; ++t) {
        const pdmpc_maneuver& m = mpa->maneuvers[t];
        if (m.n_cols < 2 || m.n_cols > PDMPC_VMAX) return fail(PDMPC_ERR_INVALID, "maneuver area column count out of range");
        pose[t].dx = m.dx;
        pose[t].dy = m.dy;
        pose[t].dyaw = m.dyaw;
        pose[t].n_cols = m.n_cols;
        pose[t].pad = 0;
        const double(*src[3])[PDMPC_VMAX] = {m.area, m.area_without_offset, m.area_large_offset};
        for (int a = 0; a < 3; ++a)
            for (int v = 0; v < m.n_cols; ++v) {
                area[(((size_t)t * 3 + a) * PDMPC_VMAX + v) * 2 + 0] = src[a][0][v];
                area[(((size_t)t * 3 + a) * PDMPC_VMAX + v) * 2 + 1] = src[a][1][v];
            }
    }
    if (h->d_mask.ensure(mask.size()) || h->d_mi.ensure(mi.size()) || h->d_pose.ensure(pose.size()) || h->d_area.ensure(area.size()))
        return fail(PDMPC_ERR_HIP, "hipMalloc failed for the MPA tables");
    HIPCHK(hipMemcpy(h->d_mask.p, mask.data(), mask.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_mi.p, mi.data(), mi.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_pose.p, pose.data(), pose.size() * sizeof(DevManPose), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_area.p, area.data(), area.size() * 8, hipMemcpyHostToDevice));
    h->n_trims = n;
    h->n_words = nw;
    h->n_man = T;
    h->mask_bytes = (size_t)Hp * n * nw * 8;
    h->mi_bytes = (size_t)n * n * 2;
    // B_mpa = 8*T*(3 + 6*VMAX) + n*n*Hp/8
    h->mpa_alg_bytes = (int64_t)8 * T * (3 + 6 * PDMPC_VMAX) + (int64_t)n * n * Hp / 8;
    h->has_mpa = true;
    return PDMPC_OK;
}

int pdmpc_pack_batch(pdmpc_handle* h, int32_t n, const pdmpc_vehicle_in* in) {
    if (h) HIPCHK(hipSetDevice(h->cfg.device));
    return pack_common(h, n, in, nullptr, nullptr, nullptr);
}

int pdmpc_pack_step(pdmpc_handle* h, int32_t n, const pdmpc_vehicle_in* in, const int32_t* pred_offset, const int32_t* pred_index,
                    const pdmpc_polygon_set* fallback_shapes) {
    if (h) HIPCHK(hipSetDevice(h->cfg.device));
    if (pred_offset && !pred_index) return fail(PDMPC_ERR_INVALID, "pred_index missing");
    return pack_common(h, n, in, pred_offset, pred_index, fallback_shapes);
}

int pdmpc_launch_packed(pdmpc_handle* h) {
    if (!h) return fail(PDMPC_ERR_INVALID, "null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    h->epoch += 1;  // a new step: results of earlier launches no longer satisfy predecessor waits
    return launch_range(h, 0, h->banks[h->bank].n_packed, h->safe_launches);
}

int pdmpc_set_safe_launch(pdmpc_handle* h, int32_t on) {
    if (!h) return fail(PDMPC_ERR_INVALID, "null handle");
    h->safe_launches = on != 0;
    return PDMPC_OK;
}

int pdmpc_begin_step(pdmpc_handle* h) {
    if (!h) return fail(PDMPC_ERR_INVALID, "null handle");
    h->epoch += 1;
    return PDMPC_OK;
}

int pdmpc_select_bank(pdmpc_handle* h, int32_t bank) {
    if (!h) return fail(PDMPC_ERR_INVALID, "null handle");
    if (bank < 0 || bank >= 4096) return fail(PDMPC_ERR_INVALID, "bank out of range");
    if ((size_t)bank >= h->banks.size()) h->banks.resize((size_t)bank + 1);
    h->bank = bank;
    return PDMPC_OK;
}

int pdmpc_reset_stats(pdmpc_handle* h) {
    if (!h) return fail(PDMPC_ERR_INVALID, "null handle");
    HIPCHK(hipStreamSynchronize(h->stream));
    h->events_used = 0;
    HIPCHK(hipMemsetAsync(h->d_tie_count.p, 0, 4 * sizeof(int32_t), h->stream));
    HIPCHK(hipMemsetAsync(h->d_work_count.p, 0, 8 * sizeof(unsigned long long), h->stream));
    return PDMPC_OK;
}

int pdmpc_launch_range(pdmpc_handle* h, int32_t first, int32_t count) {
    if (!h) return fail(PDMPC_ERR_INVALID, "null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    return launch_range(h, first, count, h->safe_launches);
}

int pdmpc_synchronize(pdmpc_handle* h) {
    if (!h) return fail(PDMPC_ERR_INVALID, "null handle");
    HIPCHK(sync_stream(h));
    return PDMPC_OK;
}

int pdmpc_fetch_results(pdmpc_handle* h, int32_t n, pdmpc_vehicle_out* out) {
    if (!h || (n > 0 && !out)) return fail(PDMPC_ERR_INVALID, "null argument");
    if (n < 0 || n > h->max_vehicles) return fail(PDMPC_ERR_INVALID, "bad record count");
    HIPCHK(hipSetDevice(h->cfg.device));
    PackedStep& B = h->banks[h->bank];
    const bool permuted = !B.perm.empty();
    if (permuted && n != B.n_packed) return fail(PDMPC_ERR_INVALID, "a batch that pdmpc_pack_step put into level order is fetched as a whole");
    if (h->h_out.ensure((size_t)std::max(n, 1))) return fail(PDMPC_ERR_HIP, "hipHostMalloc failed");
    if (n > 0) HIPCHK(hipMemcpyAsync(h->h_out.p, h->d_out.p, (size_t)n * sizeof(pdmpc_vehicle_out), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(sync_stream(h));
    if (n > 0) std::memcpy(out, h->h_out.p, (size_t)n * sizeof(pdmpc_vehicle_out));
    for (int i = 0; i < n; ++i)
        if (out[i].status == PDMPC_ERR_HIP) h->boards_dirty = true;  // (a search left through its watchdog: launch_range clears the helper boards)
    if (h->last_launch_bulk) {
        // A search of the bulk kernel that meets equal keys where the pop order depends on the layout of the reference's binary heap
        // (priority_queue_interface_mex.cpp:19-31) ends with an internal status: the slots of that launch are planned again by the
        // frontier kernel, which redoes such a search on the libstdc++-faithful heap.  Same epoch (results of other launches of the
        // step stay valid), done flags of the range cleared first.
        bool tie = false;
        for (int i = 0; i < n; ++i) tie = tie || out[i].status == PDMPC_INTERNAL_TIE;
        if (tie) {
            // every range this handle launched in the step (a level-sharded step launches one per level): a later level has read the
            // tied search's record
            h->tie_replans += 1;
            const std::vector<std::pair<int, int>> ranges = h->step_ranges;
            for (const auto& r : ranges) HIPCHK(hipMemsetAsync(h->d_flag.p + r.first, 0, (size_t)r.second * sizeof(uint32_t), h->stream));
            h->force_frontier = true;
            int rc = 0;
            for (const auto& r : ranges)
                if (!rc) rc = launch_range(h, r.first, r.second);
            h->force_frontier = false;
            if (rc) return rc;
            HIPCHK(hipMemcpyAsync(h->h_out.p, h->d_out.p, (size_t)n * sizeof(pdmpc_vehicle_out), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(sync_stream(h));
            std::memcpy(out, h->h_out.p, (size_t)n * sizeof(pdmpc_vehicle_out));
        }
    }
    // counters + SURVEY.md 8(d) algorithmic bytes of one pass over the packed batch
    pdmpc_stats& s = h->stats;
    const int Hp = h->cfg.Hp;
    const int m = std::min(n, B.n_packed);
    s.n_vehicles = m;
    s.nodes_popped = s.nodes_generated = s.obstacle_columns = 0;
    int64_t bytes = h->mpa_alg_bytes;
    for (int i = 0; i < m; ++i) {
        const pdmpc_vehicle_out& o = out[i];
        const DevVehicle& d = B.h_veh[i];
        int64_t cols = B.lit_cols[i];
        for (int q = 0; q < d.n_pred; ++q) {
            const int ps = B.h_pred[d.pred_off + q];
            if (ps < n)
                for (int k = 0; k < Hp; ++k) cols += out[ps].shape_cols[k] + 1;
        }
        const int64_t P = o.n_popped, C = std::max(o.n_expanded - 1, 0);
        s.nodes_popped += P;
        s.nodes_generated += C;
        s.obstacle_columns += cols;
        bytes += 8 * (4 + 3 * Hp) + 16 * cols;                             // B_in
        bytes += P * (60 + 16);                                            // B_pop
        bytes += C * (60 + 16);                                            // B_child
        bytes += 8 * (3 * Hp + Hp + (Hp + 1)) + 16 * PDMPC_VMAX * Hp;      // B_out
    }
    s.algorithmic_bytes = bytes;
    if (permuted) {  // back into the caller's order (the tree_path ids are per search: nothing else refers to slots)
        std::vector<pdmpc_vehicle_out> tmp(out, out + n);
        for (int sl = 0; sl < n; ++sl) out[B.perm[(size_t)sl]] = tmp[(size_t)sl];
    }
    return PDMPC_OK;
}

namespace {
// The reference's tree grows without bound (Tree.m:54-70); the arenas here are finite.  A call whose search outgrows them
// is planned again from scratch with arenas twice as large (searches are deterministic, so the vehicles that did fit
// produce the same records again) until it fits, the limit set with pdmpc_set_arena_limit is reached, or HBM runs out.
int plan_packed_growing(pdmpc_handle* h, int32_t n, pdmpc_vehicle_out* out) {
    bool safe = h->safe_launches;
    for (;;) {
        const bool dbg = h->tune.debug_host == 1;
        if (dbg) fprintf(stderr, "pdmpc: launching %d vehicles, arena %u nodes%s\n", n, h->max_nodes, safe ? " (resident slices)" : "");
        HIPCHK(hipSetDevice(h->cfg.device));
        h->epoch += 1;  // a new step: results of earlier launches no longer satisfy predecessor waits
        const auto t0 = std::chrono::steady_clock::now();
        int rc = launch_range(h, 0, h->banks[h->bank].n_packed, safe);
        if (rc) return rc;
        const auto t1 = std::chrono::steady_clock::now();
        rc = pdmpc_fetch_results(h, n, out);
        if (rc) return rc;
        if (h->tune.debug_host == 2) {  // (PDMPC_DEBUG_HOST=2: where a call's host time goes, printed by pdmpc_plan_step_literal)
            h->dbg_us[1] += std::chrono::duration<double, std::micro>(t1 - t0).count();
            h->dbg_us[2] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
            float ms = 0.f;
            if (h->events_used > 0 && hipEventElapsedTime(&ms, h->events[h->events_used - 1].first, h->events[h->events_used - 1].second) == hipSuccess) h->dbg_us[3] += 1e3 * ms;
        }
        if (dbg) fprintf(stderr, "pdmpc: fetched, status[0] %d\n", n > 0 ? out[0].status : 0);
        bool overflow = false, timed_out = false;
        for (int i = 0; i < n; ++i) {
            overflow = overflow || out[i].status == PDMPC_ARENA_OVERFLOW;
            timed_out = timed_out || out[i].status == PDMPC_ERR_HIP;
        }
        if (timed_out && safe)
            return fail(PDMPC_ERR_HIP, "a search gave up waiting for a predecessor although the call was planned in resident slices without helper workgroups (records carry PDMPC_ERR_HIP)");
        if (timed_out && !safe) {
            // A search gave up waiting for a predecessor of the same launch (the watchdog of frontier_kernel.hip): the launch was
            // oversubscribed and the dispatch order starved a predecessor, or a helper sat where a search should have run.  Plan
            // the call again in slices that are resident as a whole: forward progress then rests on nothing but slot order.
            safe = true;
            h->safe_replans += 1;
            continue;
        }
        if (!overflow) return PDMPC_OK;
        const uint64_t next = (uint64_t)h->max_nodes * 2u;
        if ((h->max_nodes_limit && next > h->max_nodes_limit) || next > (1ull << 30)) return PDMPC_OK;  // statuses tell
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        const size_t per_node = sizeof(NodeRec) + 8 + 4 + 8 + 1 + 8 + 4 + 8 + 8 + 4 + 4 + 16;
        const size_t have = (size_t)h->max_vehicles * h->max_nodes * per_node;
        if ((size_t)h->max_vehicles * next * per_node > free_b + have) return PDMPC_OK;  // no room to grow
        const uint32_t before = h->max_nodes;
        HIPCHK(hipStreamSynchronize(h->stream));
        if (alloc_arenas(h, (uint32_t)next)) {
            if (alloc_arenas(h, before)) return fail(PDMPC_ERR_HIP, "hipMalloc failed while restoring the arenas");
            return PDMPC_OK;
        }
        h->arena_regrows += 1;
    }
}
}  // namespace

int pdmpc_plan_batch(pdmpc_handle* h, int32_t n, const pdmpc_vehicle_in* in, pdmpc_vehicle_out* out) {
    const bool dbg = h && h->tune.debug_host == 2;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = pdmpc_pack_batch(h, n, in);
    if (rc) return rc;
    if (dbg) h->dbg_us[0] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    return plan_packed_growing(h, n, out);
}

int pdmpc_plan_step(pdmpc_handle* h, int32_t n, const pdmpc_vehicle_in* in, const int32_t* pred_offset, const int32_t* pred_index,
                    const pdmpc_polygon_set* fallback_shapes, pdmpc_vehicle_out* out) {
    int rc = pdmpc_pack_step(h, n, in, pred_offset, pred_index, fallback_shapes);
    if (rc) return rc;
    return plan_packed_growing(h, n, out);
}

// The step as an UNMODIFIED reference controller drives this backend (GraphSearchHip.m behind OptimizerInterface): one
// run_optimizer call per vehicle (PrioritizedController.m:335-341) in kahn order (PrioritizedSequentialController.m:77-94), every
// call a pdmpc_plan_batch of one vehicle -- pack, H2D, launch, D2H -- and the hand-over of solved areas on the host
// (PrioritizedController.m:476-491: the predecessors' info.shapes(1, :), or their published fallback areas, appended to the
// vehicle's dynamic obstacles).  Same arguments and records as pdmpc_plan_step; slots must be in level order.
int pdmpc_plan_step_literal(pdmpc_handle* h, int32_t n, const pdmpc_vehicle_in* in, const int32_t* pred_offset, const int32_t* pred_index,
                            const pdmpc_polygon_set* fallback_shapes, pdmpc_vehicle_out* out) {
    if (!h || n < 0 || (n > 0 && (!in || !out))) return fail(PDMPC_ERR_INVALID, "null argument");
    if (pred_offset && !pred_index) return fail(PDMPC_ERR_INVALID, "pred_index missing");
    const int Hp = h->cfg.Hp;
    std::vector<int32_t> off;
    std::vector<double> xs, ys;
    for (int s = 0; s < n; ++s) {
        pdmpc_vehicle_in v = in[s];
        const int np = pred_offset ? pred_offset[s + 1] - pred_offset[s] : 0;
        if (np > 0) {
            const pdmpc_polygon_set& d = in[s].dynamic_obstacles;
            if (d.n_polygons % Hp) return fail(PDMPC_ERR_INVALID, "dynamic_obstacles must hold n_d * Hp polygons");
            off.assign(1, 0);
            xs.clear();
            ys.clear();
            auto push_poly = [&](const double* px, const double* py, int cnt) {
                xs.insert(xs.end(), px, px + cnt);
                ys.insert(ys.end(), py, py + cnt);
                off.push_back((int32_t)xs.size());
            };
            for (int p = 0; p < d.n_polygons; ++p) push_poly(d.x + d.offset[p], d.y + d.offset[p], d.offset[p + 1] - d.offset[p]);
            int rows = d.n_polygons / Hp;
            for (int e = pred_offset[s]; e < pred_offset[s + 1]; ++e) {
                const int ps = pred_index[e];
                if (ps < 0 || ps >= s) return fail(PDMPC_ERR_INVALID, "pdmpc_plan_step_literal needs the slots in level order");
                const pdmpc_vehicle_out& po = out[ps];
                if (po.status == PDMPC_OK) {
                    for (int k = 0; k < Hp; ++k) push_poly(po.shapes[k][0], po.shapes[k][1], po.shape_cols[k]);
                    rows += 1;
                } else if (fallback_shapes && fallback_shapes[ps].n_polygons == Hp) {
                    const pdmpc_polygon_set& fb = fallback_shapes[ps];
                    for (int k = 0; k < Hp; ++k) push_poly(fb.x + fb.offset[k], fb.y + fb.offset[k], fb.offset[k + 1] - fb.offset[k]);
                    rows += 1;
                }
            }
            static const double zero = 0.0;
            v.dynamic_obstacles.n_polygons = rows * Hp;
            v.dynamic_obstacles.offset = off.data();
            v.dynamic_obstacles.x = xs.empty() ? &zero : xs.data();
            v.dynamic_obstacles.y = ys.empty() ? &zero : ys.data();
        }
        int rc = pdmpc_plan_batch(h, 1, &v, out + s);
        if (rc) return rc;
        // the single-launch path publishes the fallback areas of an exhausted vehicle in its record: the same record here
        if (out[s].status == PDMPC_EXHAUSTED && fallback_shapes && fallback_shapes[s].n_polygons == Hp) {
            const pdmpc_polygon_set& fb = fallback_shapes[s];
            for (int k = 0; k < Hp; ++k) {
                const int cnt = std::min(fb.offset[k + 1] - fb.offset[k], (int32_t)PDMPC_VMAX);
                out[s].shape_cols[k] = cnt;
                for (int c = 0; c < cnt; ++c) {
                    out[s].shapes[k][0][c] = fb.x[fb.offset[k] + c];
                    out[s].shapes[k][1][c] = fb.y[fb.offset[k] + c];
                }
            }
        }
    }
    if (h->tune.debug_host == 2) {
        fprintf(stderr, "pdmpc: literal step of %d calls: pack %.0f us, launch %.0f us, fetch (incl. waiting for the kernel) %.0f us, kernels %.0f us\n", n, h->dbg_us[0], h->dbg_us[1],
                h->dbg_us[2], h->dbg_us[3]);
        h->dbg_us[0] = h->dbg_us[1] = h->dbg_us[2] = h->dbg_us[3] = 0;
    }
    return PDMPC_OK;
}

int pdmpc_get_config(pdmpc_handle* h, pdmpc_config* config, int32_t* mpa_uploaded) {
    if (!h) return fail(PDMPC_ERR_INVALID, "null handle");
    if (config) {
        *config = h->cfg;
        config->max_vehicles = h->max_vehicles;
        config->max_nodes = (int32_t)h->max_nodes;
    }
    if (mpa_uploaded) *mpa_uploaded = h->has_mpa ? 1 : 0;
    return PDMPC_OK;
}

int pdmpc_set_arena_limit(pdmpc_handle* h, int32_t max_nodes_limit) {
    if (!h || max_nodes_limit < 0) return fail(PDMPC_ERR_INVALID, "bad argument");
    h->max_nodes_limit = (uint32_t)max_nodes_limit;
    return PDMPC_OK;
}

int pdmpc_grow_arena(pdmpc_handle* h, int32_t max_nodes) {
    if (!h || max_nodes <= 0) return fail(PDMPC_ERR_INVALID, "bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    if ((uint32_t)max_nodes <= h->max_nodes) return PDMPC_OK;
    HIPCHK(hipStreamSynchronize(h->stream));
    const uint32_t before = h->max_nodes;
    if (alloc_arenas(h, (uint32_t)max_nodes)) {
        if (alloc_arenas(h, before)) return fail(PDMPC_ERR_HIP, "hipMalloc failed while restoring the arenas");
        return fail(PDMPC_ERR_CAPACITY, "not enough HBM for arenas of that size");
    }
    return PDMPC_OK;
}

int pdmpc_arena_nodes(pdmpc_handle* h, int32_t* max_nodes, int64_t* regrows) {
    if (!h) return fail(PDMPC_ERR_INVALID, "null handle");
    if (max_nodes) *max_nodes = (int32_t)h->max_nodes;
    if (regrows) *regrows = h->arena_regrows;
    return PDMPC_OK;
}

namespace {
// what MATLAB's rand(RandStream('mt19937ar', Seed = s), 1, n) draws (MonteCarloTreeSearch.m:32,53)
void mt19937ar_doubles(uint32_t seed, int n, double* out) {
    Mt19937ar rng(seed);
    for (int i = 0; i < n; ++i) out[i] = rng.rand();
}
}  // namespace

int pdmpc_plan_batch_sampled(pdmpc_handle* h, int32_t n, const pdmpc_vehicle_in* in, const uint32_t* seeds, pdmpc_vehicle_out* out) {
    if (!h || n < 0 || (n > 0 && (!in || !seeds || !out))) return fail(PDMPC_ERR_INVALID, "null argument");
    int rc = pdmpc_pack_batch(h, n, in);
    if (rc) return rc;
    if (n == 0) return PDMPC_OK;
    HIPCHK(hipSetDevice(h->cfg.device));
    const int per = h->cfg.Hp * 250;  // Hp * n_expansions_max                             MonteCarloTreeSearch.m:53
    std::vector<double> rnd((size_t)n * per);
    for (int i = 0; i < n; ++i) mt19937ar_doubles(seeds[i], per, rnd.data() + (size_t)i * per);
    if (h->d_random.ensure(rnd.size())) return fail(PDMPC_ERR_HIP, "hipMalloc failed for the random numbers");
    HIPCHK(hipMemcpyAsync(h->d_random.p, rnd.data(), rnd.size() * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));  // (rnd goes out of scope)
    h->sampled_n_random = per;
    h->sampled_launch = true;
    rc = pdmpc_launch_packed(h);
    h->sampled_launch = false;
    if (rc) return rc;
    return pdmpc_fetch_results(h, n, out);
}

int pdmpc_result_device_buffer(pdmpc_handle* h, void** dev_ptr, size_t* nbytes) {
    if (!h || !dev_ptr || !nbytes) return fail(PDMPC_ERR_INVALID, "null argument");
    if (!h->banks[h->bank].perm.empty()) return fail(PDMPC_ERR_INVALID, "the packed batch was put into level order by the library: raw slots are not the caller's vehicles (pack it in level order to use the device-resident record path)");
    *dev_ptr = h->d_out.p;
    *nbytes = (size_t)h->max_vehicles * sizeof(pdmpc_vehicle_out);
    return PDMPC_OK;
}

int pdmpc_import_results(pdmpc_handle* h, int32_t first, int32_t n, const void* dev_records) {
    if (!h || (n > 0 && !dev_records)) return fail(PDMPC_ERR_INVALID, "null argument");
    if (!h->banks[h->bank].perm.empty()) return fail(PDMPC_ERR_INVALID, "the packed batch was put into level order by the library: raw slots are not the caller's vehicles (pack it in level order to use the device-resident record path)");
    if (first < 0 || n < 0 || first + n > h->max_vehicles) return fail(PDMPC_ERR_INVALID, "slot range out of bounds");
    HIPCHK(hipSetDevice(h->cfg.device));
    if (n == 0) return PDMPC_OK;
    const void* dst = (const void*)(h->d_out.p + first);
    if (dev_records != dst)
        HIPCHK(hipMemcpyAsync(h->d_out.p + first, dev_records, (size_t)n * sizeof(pdmpc_vehicle_out), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(h->d_flag.p + first), (int)h->epoch, (size_t)n, h->stream));
    return PDMPC_OK;
}

int pdmpc_export_results(pdmpc_handle* h, int32_t first, int32_t n, void* dev_records) {
    if (!h || (n > 0 && !dev_records)) return fail(PDMPC_ERR_INVALID, "null argument");
    if (!h->banks[h->bank].perm.empty()) return fail(PDMPC_ERR_INVALID, "the packed batch was put into level order by the library: raw slots are not the caller's vehicles (pack it in level order to use the device-resident record path)");
    if (first < 0 || n < 0 || first + n > h->max_vehicles) return fail(PDMPC_ERR_INVALID, "slot range out of bounds");
    HIPCHK(hipSetDevice(h->cfg.device));
    if (n > 0) HIPCHK(hipMemcpyAsync(dev_records, h->d_out.p + first, (size_t)n * sizeof(pdmpc_vehicle_out), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PDMPC_OK;
}

int pdmpc_export_results_async(pdmpc_handle* h, int32_t first, int32_t n, void* dev_records) {
    if (!h || (n > 0 && !dev_records)) return fail(PDMPC_ERR_INVALID, "null argument");
    if (!h->banks[h->bank].perm.empty()) return fail(PDMPC_ERR_INVALID, "the packed batch was put into level order by the library: raw slots are not the caller's vehicles (pack it in level order to use the device-resident record path)");
    if (first < 0 || n < 0 || first + n > h->max_vehicles) return fail(PDMPC_ERR_INVALID, "slot range out of bounds");
    HIPCHK(hipSetDevice(h->cfg.device));
    if (n > 0) HIPCHK(hipMemcpyAsync(dev_records, h->d_out.p + first, (size_t)n * sizeof(pdmpc_vehicle_out), hipMemcpyDeviceToDevice, h->stream));
    return PDMPC_OK;
}

int pdmpc_stream(pdmpc_handle* h, void** hip_stream) {
    if (!h || !hip_stream) return fail(PDMPC_ERR_INVALID, "null argument");
    *hip_stream = (void*)h->stream;
    return PDMPC_OK;
}

int pdmpc_get_last_stats(pdmpc_handle* h, pdmpc_stats* stats) {
    if (!h || !stats) return fail(PDMPC_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    double ms = 0.0;
    for (size_t i = 0; i < h->events_used; ++i) {
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, h->events[i].first, h->events[i].second));
        ms += t;
    }
    h->stats.kernel_ms = ms;
    h->stats.n_launches = (int64_t)h->events_used;
    int32_t ctr[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(ctr, h->d_tie_count.p, sizeof ctr, hipMemcpyDeviceToHost));
    h->stats.queue_fallbacks = ctr[0];
    h->stats.speculation_restarts = ctr[1];
    h->stats.speculation_arrivals = ctr[2];
    h->stats.speculation_wasted_pops = ctr[3];
    unsigned long long work[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpy(work, h->d_work_count.p, sizeof work, hipMemcpyDeviceToHost));
    h->stats.edge_checks = (int64_t)work[0];
    h->stats.segment_pair_tests = (int64_t)work[1];
    h->stats.kernel = h->last_launch_bulk ? 2 : (h->last_launch_frontier ? 1 : 0);
    h->stats.entries_dropped = h->last_launch_frontier ? 0 : (int64_t)work[2];
    h->stats.dropped_counted_as_pops = h->last_launch_frontier ? 0 : (int64_t)work[3];
    h->stats.nodes_processed = h->last_launch_frontier ? (int64_t)work[2] : 0;
    h->stats.rounds = h->last_launch_frontier ? (int64_t)work[3] : 0;
    h->stats.shared_rounds = (int64_t)work[4];
    h->stats.helper_checked = (int64_t)work[5];
    h->stats.safe_replans = h->safe_replans;
    h->stats.bad_status_plans = (int64_t)work[6];
    *stats = h->stats;
    return PDMPC_OK;
}

int pdmpc_debug_heap_script(pdmpc_handle* h, int32_t n, const int32_t* op, const int32_t* id, const double* key, int32_t lds_entries,
                            int32_t* popped, int32_t* n_popped, double* cycles_per_pop, double* cycles_per_push) {
    if (!h || n < 0 || (n > 0 && (!op || !id || !key)) || !popped || !n_popped) return fail(PDMPC_ERR_INVALID, "null argument");
    if (lds_entries < 64 || lds_entries > 8192 || (lds_entries & 1)) return fail(PDMPC_ERR_INVALID, "lds_entries must be even and in 64..8192");
    HIPCHK(hipSetDevice(h->cfg.device));
    int32_t *d_op = nullptr, *d_id = nullptr, *d_out = nullptr;
    double *d_key = nullptr, *d_gkey = nullptr;
    uint32_t* d_gid = nullptr;
    unsigned long long* d_stats = nullptr;
    const size_t m = (size_t)std::max(n, 1);
    HIPCHK(hipMalloc((void**)&d_op, m * 4));
    HIPCHK(hipMalloc((void**)&d_id, m * 4));
    HIPCHK(hipMalloc((void**)&d_out, m * 4));
    HIPCHK(hipMalloc((void**)&d_key, m * 8));
    HIPCHK(hipMalloc((void**)&d_gkey, (m + 2) * 8));
    HIPCHK(hipMalloc((void**)&d_gid, (m + 2) * 4));
    HIPCHK(hipMalloc((void**)&d_stats, 4 * 8));
    HIPCHK(hipMemcpy(d_op, op, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_id, id, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_key, key, (size_t)n * 8, hipMemcpyHostToDevice));
    int lrc = pdmpc_launch_heap_script(d_op, d_id, d_key, n, d_out, d_stats, d_gkey, d_gid, lds_entries, (void*)h->stream);
    if (lrc != 0) return fail(PDMPC_ERR_HIP, "heap script launch failed");
    HIPCHK(hipStreamSynchronize(h->stream));
    unsigned long long st[4];
    HIPCHK(hipMemcpy(st, d_stats, sizeof st, hipMemcpyDeviceToHost));
    int cnt = 0;
    for (int i = 0; i < n; ++i) cnt += op[i] == 1;
    *n_popped = cnt;
    if (cnt > 0) HIPCHK(hipMemcpy(popped, d_out, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    if (cycles_per_pop) *cycles_per_pop = st[1] ? (double)st[0] / (double)st[1] : 0.0;
    if (cycles_per_push) *cycles_per_push = st[3] ? (double)st[2] / (double)st[3] : 0.0;
    (void)hipFree(d_op);
    (void)hipFree(d_id);
    (void)hipFree(d_out);
    (void)hipFree(d_key);
    (void)hipFree(d_gkey);
    (void)hipFree(d_gid);
    (void)hipFree(d_stats);
    return PDMPC_OK;
}

int pdmpc_debug_blockmin_script(pdmpc_handle* h, int32_t n, const int32_t* op, const double* key, int32_t ring_entries, int32_t* popped,
                                int32_t* n_popped, int32_t* tie, double* cycles_per_pop, double* cycles_per_push) {
    if (!h || n < 0 || (n > 0 && (!op || !key)) || !popped || !n_popped || !tie) return fail(PDMPC_ERR_INVALID, "null argument");
    if (ring_entries < 64 || ring_entries > 8192 || (ring_entries & (ring_entries - 1))) return fail(PDMPC_ERR_INVALID, "ring_entries must be a power of two in 64..8192");
    int n_push = 0;
    for (int i = 0; i < n; ++i) n_push += op[i] == 0;
    if (n_push > 64 * 4096) return fail(PDMPC_ERR_INVALID, "at most 262144 pushes");
    const int NB = (n_push + 63) / 64 + 64;
    HIPCHK(hipSetDevice(h->cfg.device));
    int32_t *d_op = nullptr, *d_out = nullptr;
    double *d_key = nullptr, *d_gkey = nullptr;
    unsigned long long* d_stats = nullptr;
    const size_t m = (size_t)std::max(n, 1);
    HIPCHK(hipMalloc((void**)&d_op, m * 4));
    HIPCHK(hipMalloc((void**)&d_out, m * 4));
    HIPCHK(hipMalloc((void**)&d_key, m * 8));
    HIPCHK(hipMalloc((void**)&d_gkey, ((size_t)n_push + 128) * 8));
    HIPCHK(hipMalloc((void**)&d_stats, 8 * 8));
    HIPCHK(hipMemcpy(d_op, op, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_key, key, (size_t)n * 8, hipMemcpyHostToDevice));
    int lrc = pdmpc_launch_bm_script(d_op, d_key, n, d_out, d_stats, d_gkey, ring_entries, NB, (void*)h->stream);
    if (lrc != 0) return fail(PDMPC_ERR_HIP, "block-min script launch failed");
    HIPCHK(hipStreamSynchronize(h->stream));
    unsigned long long st[5];
    HIPCHK(hipMemcpy(st, d_stats, sizeof st, hipMemcpyDeviceToHost));
    const int cnt = n - n_push;
    *n_popped = cnt;
    *tie = (int32_t)st[4];
    if (cnt > 0) HIPCHK(hipMemcpy(popped, d_out, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    if (cycles_per_pop) *cycles_per_pop = st[1] ? (double)st[0] / (double)st[1] : 0.0;
    if (cycles_per_push) *cycles_per_push = st[3] ? (double)st[2] / (double)st[3] : 0.0;
    (void)hipFree(d_op);
    (void)hipFree(d_out);
    (void)hipFree(d_key);
    (void)hipFree(d_gkey);
    (void)hipFree(d_stats);
    return PDMPC_OK;
}

namespace {
#define PDMPC_TREE_FRONTIER 0x40000000  /* d_tree_size marker: the arena holds the frontier kernel's raw tree (creation order differs from the reference's) */
#define PDMPC_TREE_REPLAYED 0x20000000  /* ... and the search ended on the replay through the binary heap (equal keys): its pop sequence, in arena indices, is in the mid list's array */
#define PDMPC_TREE_SIZE(sz) ((sz) & ~(PDMPC_TREE_FRONTIER | PDMPC_TREE_REPLAYED))

// The frontier kernel processes open nodes in parallel, so its arena holds the reference's tree plus some nodes the
// reference never creates, in another order.  This turns it back into the reference's tree and pop sequence, on the host
// and independently of the kernel's phase B (it sorts the popped nodes instead of counting them), for the debug read-backs
// the parity tests use.  Order (frontier_kernel.hip): X is popped before Y iff X is an ancestor of Y or the largest key on
// the path (LCA, X] is smaller than the largest key on (LCA, Y].
struct RefTree {
    std::vector<NodeRec> rec;        // raw records
    std::vector<uint32_t> pops;      // raw indices in the reference's pop order
    std::vector<uint32_t> ref_nodes; // raw index of reference node id r (0-based position = id - 1)
    std::vector<uint32_t> ref_id;    // raw index -> reference id (0: not in the reference's tree)
};
int reconstruct_reference_tree(pdmpc_handle* h, int vehicle, uint32_t raw_n, RefTree& T, bool replayed = false) {
    const size_t off = (size_t)vehicle * h->max_nodes;
    const int Hp = h->cfg.Hp;
    T.rec.resize(raw_n);
    std::vector<double> key(raw_n);
    std::vector<uint8_t> vs(raw_n);
    HIPCHK(hipMemcpy(T.rec.data(), h->anodes.p + off, (size_t)raw_n * sizeof(NodeRec), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(key.data(), h->ahk.p + off, (size_t)raw_n * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(vs.data(), h->avs.p + off, (size_t)raw_n, hipMemcpyDeviceToHost));
    pdmpc_vehicle_out out;
    HIPCHK(hipMemcpy(&out, h->d_out.p + vehicle, sizeof out, hipMemcpyDeviceToHost));
    const std::vector<NodeRec>& R = T.rec;
    std::vector<uint8_t> alive(raw_n, 0);
    alive[0] = 1;
    for (uint32_t i = 1; i < raw_n; ++i) {
        const uint32_t p = R[i].parent - 1;
        alive[i] = alive[p] && vs[p] == 1;
    }
    auto depth = [&](uint32_t i) { return NODE_K(R[i].packed); };
    // -1: x first, +1: y first, 0: same node
    auto before = [&](uint32_t x, uint32_t y) -> int {
        if (x == y) return 0;
        double mx = -1.0, my = -1.0;
        uint32_t a = x, b = y;
        while (depth(a) > depth(b)) {
            mx = std::max(mx, key[a]);
            a = R[a].parent - 1;
        }
        while (depth(b) > depth(a)) {
            my = std::max(my, key[b]);
            b = R[b].parent - 1;
        }
        if (a == b) return depth(x) < depth(y) ? -1 : 1;  // ancestor first
        while (a != b) {
            mx = std::max(mx, key[a]);
            my = std::max(my, key[b]);
            a = R[a].parent - 1;
            b = R[b].parent - 1;
        }
        return mx < my ? -1 : 1;
    };
    // the goal: the first collision-free node at the horizon
    int64_t goal = -1;
    if (out.status == PDMPC_OK)
        for (uint32_t i = 0; i < raw_n; ++i)
            if (alive[i] && vs[i] == 1 && depth(i) == Hp && (goal < 0 || before(i, (uint32_t)goal) < 0)) goal = i;
    T.pops.clear();
    if (replayed) {
        // equal keys: the order is the binary heap's, which the kernel's replay has run (bulk_search.hpp, bk_replay) and left behind
        T.pops.resize((size_t)std::max(out.n_popped, 0));
        if (!T.pops.empty()) HIPCHK(hipMemcpy(T.pops.data(), h->amidi.p + off, T.pops.size() * 4, hipMemcpyDeviceToHost));
    } else {
        for (uint32_t i = 0; i < raw_n; ++i)
            if (alive[i] && (goal < 0 || i == (uint32_t)goal || before(i, (uint32_t)goal) < 0)) T.pops.push_back(i);
        std::sort(T.pops.begin(), T.pops.end(), [&](uint32_t x, uint32_t y) { return before(x, y) < 0; });
    }
    // children of a node are consecutive raw indices in ascending trim order
    std::vector<uint32_t> first_child(raw_n, 0), n_child(raw_n, 0);
    for (uint32_t i = raw_n; i-- > 1;) {
        const uint32_t p = R[i].parent - 1;
        first_child[p] = i;
        n_child[p] += 1;
    }
    T.ref_id.assign(raw_n, 0);
    T.ref_nodes.clear();
    T.ref_nodes.push_back(0);
    T.ref_id[0] = 1;
    for (uint32_t x : T.pops) {
        if (vs[x] != 1 || depth(x) == Hp) continue;  // discarded (GraphSearch.m:75-77) or the goal
        for (uint32_t c = 0; c < n_child[x]; ++c) {
            T.ref_nodes.push_back(first_child[x] + c);
            T.ref_id[first_child[x] + c] = (uint32_t)T.ref_nodes.size();
        }
    }
    return PDMPC_OK;
}
}  // namespace

int pdmpc_debug_pop_trace(pdmpc_handle* h, int32_t vehicle, int32_t capacity, int32_t* ids, int32_t* n) {
    if (!h || !ids || !n) return fail(PDMPC_ERR_INVALID, "null argument");
    if (h->cfg.trace_pops <= 0 && !h->last_launch_frontier) return fail(PDMPC_ERR_INVALID, "handle was created with trace_pops == 0");
    if (vehicle < 0 || vehicle >= h->max_vehicles) return fail(PDMPC_ERR_INVALID, "vehicle slot out of range");
    if (!h->banks[h->bank].inv.empty() && vehicle < h->banks[h->bank].n_packed) vehicle = h->banks[h->bank].inv[(size_t)vehicle];  // (the batch was put into level order)
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    {
        int32_t sz = 0;
        HIPCHK(hipMemcpy(&sz, h->d_tree_size.p + vehicle, 4, hipMemcpyDeviceToHost));
        if (sz & PDMPC_TREE_FRONTIER) {
            RefTree T;
            int rc = reconstruct_reference_tree(h, vehicle, (uint32_t)PDMPC_TREE_SIZE(sz), T, (sz & PDMPC_TREE_REPLAYED) != 0);
            if (rc) return rc;
            *n = (int32_t)T.pops.size();
            for (size_t i = 0; i < T.pops.size() && (int)i < capacity; ++i) ids[i] = (int32_t)T.ref_id[T.pops[i]];
            return PDMPC_OK;
        }
    }
    pdmpc_vehicle_out rec;
    HIPCHK(hipMemcpy(&rec, h->d_out.p + vehicle, sizeof rec, hipMemcpyDeviceToHost));
    const int cnt = std::min(rec.n_popped, h->cfg.trace_pops);
    *n = cnt;
    const int m = std::min(cnt, capacity);
    if (m > 0) HIPCHK(hipMemcpy(ids, h->d_trace.p + (size_t)vehicle * h->cfg.trace_pops, (size_t)m * 4, hipMemcpyDeviceToHost));
    return PDMPC_OK;
}

int pdmpc_debug_edge_check(pdmpc_handle* h, int32_t mode, int32_t n_cases, const int32_t* a_off, const double* a_x, const double* a_y, const int32_t* b_off,
                           const double* b_x, const double* b_y, int32_t* hit) {
    if (!h || n_cases < 0 || (n_cases > 0 && (!a_off || !a_x || !a_y || !b_off || !b_x || !b_y || !hit))) return fail(PDMPC_ERR_INVALID, "null argument");
    if (mode < 0 || mode > 2) return fail(PDMPC_ERR_INVALID, "mode must be 0 (InterX), 1 (intersect_sat) or 2 (intersect_lanelet_boundary)");
    if (n_cases == 0) return PDMPC_OK;
    for (int c = 0; c < n_cases; ++c) {
        const int na = a_off[c + 1] - a_off[c], nb = b_off[c + 1] - b_off[c];
        if (na < 0 || na > PDMPC_VMAX) return fail(PDMPC_ERR_INVALID, "first operand: at most PDMPC_VMAX columns");
        if (nb < 0 || nb > 1024) return fail(PDMPC_ERR_INVALID, "second operand: at most 1024 columns");
    }
    HIPCHK(hipSetDevice(h->cfg.device));
    const size_t ta = (size_t)a_off[n_cases], tb = (size_t)b_off[n_cases];
    // (DevBuf-style owners: every early return frees what was allocated)
    struct Owned {
        void* p = nullptr;
        ~Owned() {
            if (p) (void)hipFree(p);
        }
        hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 8)); }
    } o_ao, o_bo, o_hit, o_ax, o_ay, o_bx, o_by;
    HIPCHK(o_ao.alloc(((size_t)n_cases + 1) * 4));
    HIPCHK(o_bo.alloc(((size_t)n_cases + 1) * 4));
    HIPCHK(o_hit.alloc((size_t)n_cases * 4));
    HIPCHK(o_ax.alloc(ta * 8));
    HIPCHK(o_ay.alloc(ta * 8));
    HIPCHK(o_bx.alloc(tb * 8));
    HIPCHK(o_by.alloc(tb * 8));
    int32_t *d_ao = (int32_t*)o_ao.p, *d_bo = (int32_t*)o_bo.p, *d_hit = (int32_t*)o_hit.p;
    double *d_ax = (double*)o_ax.p, *d_ay = (double*)o_ay.p, *d_bx = (double*)o_bx.p, *d_by = (double*)o_by.p;
    HIPCHK(hipMemcpy(d_ao, a_off, ((size_t)n_cases + 1) * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_bo, b_off, ((size_t)n_cases + 1) * 4, hipMemcpyHostToDevice));
    if (ta) {
        HIPCHK(hipMemcpy(d_ax, a_x, ta * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_ay, a_y, ta * 8, hipMemcpyHostToDevice));
    }
    if (tb) {
        HIPCHK(hipMemcpy(d_bx, b_x, tb * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_by, b_y, tb * 8, hipMemcpyHostToDevice));
    }
    const int lrc = pdmpc_launch_edge_check(mode, n_cases, d_ao, d_ax, d_ay, d_bo, d_bx, d_by, d_hit, (void*)h->stream);
    if (lrc != 0) return fail(PDMPC_ERR_HIP, "edge-check launch failed");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(hit, d_hit, (size_t)n_cases * 4, hipMemcpyDeviceToHost));
    return PDMPC_OK;
}

int pdmpc_debug_raw_tree(pdmpc_handle* h, int32_t vehicle, int32_t capacity, double* x, double* y, double* yaw, double* g, double* hh, int32_t* trim,
                         int32_t* k, int32_t* parent, double* key, uint8_t* validity, int32_t* n) {
    if (!h || !n) return fail(PDMPC_ERR_INVALID, "null argument");
    if (vehicle < 0 || vehicle >= h->max_vehicles) return fail(PDMPC_ERR_INVALID, "vehicle slot out of range");
    if (!h->banks[h->bank].inv.empty() && vehicle < h->banks[h->bank].n_packed) vehicle = h->banks[h->bank].inv[(size_t)vehicle];  // (the batch was put into level order)
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    int32_t sz = 0;
    HIPCHK(hipMemcpy(&sz, h->d_tree_size.p + vehicle, 4, hipMemcpyDeviceToHost));
    sz = PDMPC_TREE_SIZE(sz);
    *n = sz;
    const size_t m = (size_t)std::max(std::min(sz, capacity), 0);
    if (m == 0) return PDMPC_OK;
    const size_t off = (size_t)vehicle * h->max_nodes;
    std::vector<NodeRec> rec(m);
    HIPCHK(hipMemcpy(rec.data(), h->anodes.p + off, m * sizeof(NodeRec), hipMemcpyDeviceToHost));
    if (key) HIPCHK(hipMemcpy(key, h->ahk.p + off, m * 8, hipMemcpyDeviceToHost));
    if (validity) HIPCHK(hipMemcpy(validity, h->avs.p + off, m, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < m; ++i) {
        if (x) x[i] = rec[i].x;
        if (y) y[i] = rec[i].y;
        if (yaw) yaw[i] = rec[i].yaw;
        if (g) g[i] = rec[i].g;
        if (hh) hh[i] = rec[i].h;
        if (parent) parent[i] = (int32_t)rec[i].parent;
        if (trim) trim[i] = NODE_TRIM(rec[i].packed);
        if (k) k[i] = NODE_K(rec[i].packed);
    }
    return PDMPC_OK;
}

int pdmpc_debug_progress(pdmpc_handle* h, int32_t vehicle, uint32_t* words16) {
    if (!h || !words16 || vehicle < 0 || vehicle >= h->max_vehicles) return fail(PDMPC_ERR_INVALID, "bad argument");
    if (!h->banks[h->bank].inv.empty() && vehicle < h->banks[h->bank].n_packed) vehicle = h->banks[h->bank].inv[(size_t)vehicle];  // (the batch was put into level order)
    for (int i = 0; i < 32; ++i) words16[i] = h->progress ? ((volatile uint32_t*)h->progress)[vehicle * 64 + i] : 0u;
    return PDMPC_OK;
}

int pdmpc_debug_tree(pdmpc_handle* h, int32_t vehicle, int32_t capacity, double* x, double* y, double* yaw, double* g, double* hh,
                     int32_t* trim, int32_t* k, int32_t* parent, int32_t* n) {
    if (!h || !n) return fail(PDMPC_ERR_INVALID, "null argument");
    if (vehicle < 0 || vehicle >= h->max_vehicles) return fail(PDMPC_ERR_INVALID, "vehicle slot out of range");
    if (!h->banks[h->bank].inv.empty() && vehicle < h->banks[h->bank].n_packed) vehicle = h->banks[h->bank].inv[(size_t)vehicle];  // (the batch was put into level order)
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    int32_t sz = 0;
    HIPCHK(hipMemcpy(&sz, h->d_tree_size.p + vehicle, 4, hipMemcpyDeviceToHost));
    if (sz & PDMPC_TREE_FRONTIER) {
        RefTree T;
        int rc = reconstruct_reference_tree(h, vehicle, (uint32_t)PDMPC_TREE_SIZE(sz), T, (sz & PDMPC_TREE_REPLAYED) != 0);
        if (rc) return rc;
        *n = (int32_t)T.ref_nodes.size();
        for (size_t i = 0; i < T.ref_nodes.size() && (int)i < capacity; ++i) {
            const NodeRec& r = T.rec[T.ref_nodes[i]];
            if (x) x[i] = r.x;
            if (y) y[i] = r.y;
            if (yaw) yaw[i] = r.yaw;
            if (g) g[i] = r.g;
            if (hh) hh[i] = r.h;
            if (parent) parent[i] = r.parent ? (int32_t)T.ref_id[r.parent - 1] : 0;
            if (trim) trim[i] = NODE_TRIM(r.packed);
            if (k) k[i] = NODE_K(r.packed);
        }
        return PDMPC_OK;
    }
    *n = sz;
    const size_t m = (size_t)std::max(std::min(sz, capacity), 0);
    if (m == 0) return PDMPC_OK;
    const size_t off = (size_t)vehicle * h->max_nodes;
    std::vector<NodeRec> rec(m);
    HIPCHK(hipMemcpy(rec.data(), h->anodes.p + off, m * sizeof(NodeRec), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < m; ++i) {
        if (x) x[i] = rec[i].x;
        if (y) y[i] = rec[i].y;
        if (yaw) yaw[i] = rec[i].yaw;
        if (g) g[i] = rec[i].g;
        if (hh) hh[i] = rec[i].h;
        if (parent) parent[i] = (int32_t)rec[i].parent;
        if (trim) trim[i] = NODE_TRIM(rec[i].packed);
        if (k) k[i] = NODE_K(rec[i].packed);
    }
    return PDMPC_OK;
}

}  // extern "C"
