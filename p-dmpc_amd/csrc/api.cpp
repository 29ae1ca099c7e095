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
#include <map>
#include <mutex>
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

// The current device belongs to the caller (torch reads it with hipGetDevice: a collective issued after a call into this library
// must not find itself on another GPU).  Every entry point that works on the handle's device switches to it through this guard,
// which puts the caller's device back on every exit path.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched && prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define ON_DEVICE(dev)                 \
    DeviceGuard device_guard__((dev)); \
    HIPCHK(device_guard__.err)

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
    int ensure_keep(size_t n, size_t keep) {  // as ensure, the first `keep` elements carried over
        if (n <= cap) return 0;
        size_t want = std::max(n, (size_t)64);
        want += want / 2;
        T* q = nullptr;
        hipError_t e = hipHostMalloc((void**)&q, want * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) return (int)e;
        if (p && keep) std::memcpy(q, p, std::min(keep, cap) * sizeof(T));
        if (p) (void)hipHostFree(p);
        p = q;
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
    bool pack_failed = false;  // the last pack into this bank did not finish: nothing to launch or fetch
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

// Tuning knobs and A/B / test switches of the graph search.  The defaults are the measured optima quoted next to their use; every
// setting leaves the results bit-identical.  ONE environment variable overrides them, read once in pdmpc_create (a launch makes no
// getenv call):  PDMPC_TUNING="key=value,key=value,..."  with the keys below (include/pdmpc.h documents the variable).
// what pack_common tells vehicles that hand over the same arrays by: the pointers and counts of a vehicle's polygon sets
struct SoupKey {
    const void* p[13];
    int32_t c[6];
};

struct Tuning {
    int round0 = -1;        // nodes a round of a young search takes (-1: 24; 32 for a launch that leaves CUs idle but has fewer than four helpers per search, C3, and for one of more than two searches per CU, C5)
    int round = -1;         // the most a round takes (-1: 1000 with helper workgroups, else 256)
    int ramp = -1;          // a round grows by 1 / ramp of the nodes processed so far (-1: 2 with helper workgroups, else 4)
    int ready = 2048;       // entries of the ready list with helper workgroups (half of it without): the most a round can take
    int share_min = -1;     // a round with at least this many nodes is shared with the helper workgroups (-1: by the number of helpers per search, launch_range)
    int own_div = 8;        // (accepted, no effect since round 5: the owner takes an equal share of a shared round)
    int tile = -1;          // the most nodes of a shared round one seated helper takes (-1: 256; what it stages in LDS: at most 768)
    int mid_min = 24576;    // far lists longer than this feed near through the mid list (a band of far's smallest keys)
    int mid_fill = 12288;   // entries a refill of mid aims at
    int tentative = 1;      // expected areas of predecessors that are still planning (A/B switch)
    int fast_arrival = 1;   // finished searches check arrivals against their plan's path first and publish early (A/B switch)
    int helpers = -1;       // helper workgroups of a launch with at most one search per CU (-1: by launch size, 0: none)
    int helpers_oversub = -1;  // ... of a launch with more searches than CUs (-1: 200)
    int seat_nodes = 256;   // a search may hold its share of the launch's helpers (helpers / searches) per this many nodes it has processed
    int helpers_first = -1; // ... of them dispatched in front of the searches (-1: half the CUs when most searches of the launch have predecessors)
    int speculate = 1;      // 0: every search waits for all its predecessors before it starts
    int dispatch_order = 1; // 0: ignore pdmpc_set_step_weights (slots in level order, never by priority)
    int fast_select = 1;    // 0: every selection goes through the sixteen-wavefront histogram, also while the open set is small (A/B)
    int poll_every = 1;     // a search that has just run a round looks for arrived predecessors at every K-th round boundary only (1 .. 8)
    int lazy_verify = 0;    // 1: an arrival into a RUNNING search brings the parked nodes back at once but re-checks the collision-free nodes only when the search stalls or is done
    int compact = -1;       // 1: the kernel built for two workgroups per CU (8 wavefronts, <= 80 KB of LDS: bulk_kernel_compact.hip) where it applies (InterX, one mask word, the soup fits); 0: never; -1: for launches of more than two searches per CU; 2-5: layout experiments (one workgroup per CU with the compact kernel, slack behind the layout)
    int waves = -1;         // wavefronts per workgroup (4 .. PDMPC_MAX_WAVES; -1: 16 for the InterX kernels — 12 for a launch of more than two searches per CU —, 12 for the separating-axis kernel)
    uint32_t spin_limit = 1u << 22;  // the watchdog's limit of polls / rounds (debugging: fail fast)
    int force_tie = 0;      // testing only: every search ends on the replay through the reference's binary heap (as if it had met equal keys)
    int reverse_dispatch = 0;  // testing only: workgroup b takes slot n - 1 - b (successors dispatched before their predecessors)
    int debug_tail = 0;     // round / node / tick counters of every search in the unused rows of its record's path_nodes (tools/fr_step_profile.py)
    int debug_lds = 0;      // print the LDS layout of every launch
    int debug_host = 0;     // 1: a line per launch; 2: the host-time breakdown of the literal path
    int debug_progress = 0; // live counters in host-mapped memory (pdmpc_debug_progress)
};

namespace {
// parses PDMPC_TUNING; an unknown key or a malformed entry is an error (a typo must not silently measure the default)
bool parse_tuning(const char* text, Tuning& T, std::string& err) {
    struct Key { const char* name; int* dst; };
    int spin = (int)T.spin_limit;
    const Key keys[] = {{"round0", &T.round0}, {"round", &T.round}, {"ramp", &T.ramp}, {"ready", &T.ready}, {"share_min", &T.share_min}, {"own_div", &T.own_div},
                        {"tile", &T.tile}, {"mid_min", &T.mid_min}, {"mid_fill", &T.mid_fill}, {"tentative", &T.tentative}, {"fast_arrival", &T.fast_arrival}, {"helpers_first", &T.helpers_first}, {"seat_nodes", &T.seat_nodes},
                        {"helpers", &T.helpers}, {"helpers_oversub", &T.helpers_oversub}, {"speculate", &T.speculate}, {"waves", &T.waves}, {"compact", &T.compact}, {"lazy_verify", &T.lazy_verify}, {"poll_every", &T.poll_every}, {"fast_select", &T.fast_select}, {"dispatch_order", &T.dispatch_order}, {"spin_limit", &spin},
                        {"force_tie", &T.force_tie}, {"reverse_dispatch", &T.reverse_dispatch}, {"debug_tail", &T.debug_tail}, {"debug_lds", &T.debug_lds},
                        {"debug_host", &T.debug_host}, {"debug_progress", &T.debug_progress}};
    std::string str(text ? text : "");
    size_t pos = 0;
    while (pos < str.size()) {
        size_t end = str.find(',', pos);
        if (end == std::string::npos) end = str.size();
        const std::string item = str.substr(pos, end - pos);
        pos = end + 1;
        if (item.empty()) continue;
        const size_t eq = item.find('=');
        if (eq == std::string::npos || eq == 0 || eq + 1 >= item.size()) {
            err = "PDMPC_TUNING: '" + item + "' is not key=value";
            return false;
        }
        const std::string name = item.substr(0, eq);
        char* tail = nullptr;
        const long v = strtol(item.c_str() + eq + 1, &tail, 10);
        if (!tail || *tail) {
            err = "PDMPC_TUNING: value of '" + name + "' is not an integer";
            return false;
        }
        bool found = false;
        for (const Key& k : keys)
            if (name == k.name) {
                *k.dst = (int)v;
                found = true;
            }
        if (!found) {
            err = "PDMPC_TUNING: unknown key '" + name + "'";
            return false;
        }
    }
    if (T.round0 >= 0) T.round0 = std::max(1, T.round0);
    if (T.round >= 0) T.round = std::max(1, T.round);
    if (T.ramp >= 0) T.ramp = std::max(1, T.ramp);
    T.ready = std::min(2048, std::max(256, T.ready)) & ~63;
    if (T.share_min >= 0) T.share_min = std::max(32, T.share_min);
    T.own_div = std::max(1, T.own_div);
    if (T.tile >= 0) T.tile = std::min(768, std::max(8, T.tile));
    T.mid_min = std::max(0, T.mid_min);
    T.mid_fill = std::max(256, T.mid_fill);
    if (T.waves >= 0) T.waves = std::min(PDMPC_MAX_WAVES, std::max(4, T.waves));
    T.spin_limit = (uint32_t)std::max(1024, spin);
    return true;
}
}  // namespace

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
    // arenas (NodeArena, pdmpc_device.h)
    uint32_t max_nodes = 0;
    uint32_t max_nodes_limit = 0;  // pdmpc_plan_* may grow the arenas up to this many nodes per vehicle (0: as far as HBM allows)
    int64_t arena_regrows = 0;     // times an overflowed call was re-planned with larger arenas
    int64_t safe_replans = 0;      // times a call was re-planned in resident slices after a predecessor time-out
    bool safe_launches = false;    // pdmpc_set_safe_launch: every launch in resident slices
    int max_vehicles = 0;
    DevBuf<NodeRec> anodes;
    DevBuf<double> akey, afark, amidk, apbk, awalk;
    DevBuf<unsigned long long> alink;
    DevBuf<uint32_t> afari, amidi, apbd, achild0;
    DevBuf<uint8_t> avs;
    DevBuf<pdmpc_vehicle_out> d_out;
    DevBuf<uint32_t> d_flag;
    DevBuf<int32_t> d_tree_size;
    DevBuf<int32_t> d_tie_count;
    DevBuf<unsigned long long> d_work_count;
    DevBuf<unsigned long long> d_help_board;  // helper workgroups (pdmpc_device.h)
    DevBuf<uint32_t> d_help_verdict, d_help_finished;
    DevBuf<double> d_bk_post;                 // records posted for the helper workgroups
    DevBuf<double> d_random;  // sampled optimizer: random numbers of the batch
    int sampled_n_random = 0;
    bool sampled_launch = false;
    bool last_launch_search = false;     // the last launch ran the graph search (not the sampled optimizer)
    int last_first = 0, last_count = 0;  // slots of the last launch_range
    bool last_safe = false;              // ... and whether it went out in resident slices
    int device_share = 1;                // handles of one process that launch on this device side by side (pdmpc_set_device_share: a group's logical ranks)
    bool boards_dirty = true;            // the helper boards / the finished counter need clearing before the helper workgroups may read them
    uint32_t help_fin_total = 0;         // value of the finished counter once every launch so far has ended
    uint32_t launch_serial = 0;          // launches of this handle so far (KernelArgs::launch_id)
    double last_us[3] = {0, 0, 0};       // pdmpc_last_call_timing: pack, enqueue, wait + read-back of the last pdmpc_plan_batch / pdmpc_plan_step
    double dbg_us[4] = {0, 0, 0, 0};     // debug_host 2: pack, launch, fetch (host clock) and kernel (events) time of the plan_batch calls
    uint64_t sync_serial = 0;            // stream synchronisations through sync_stream so far (PackedStep::staged_serial)
    std::vector<SoupKey> pack_soup_keys;  // pack_common's scratch: the distinct soup keys of the batch, the slots they were packed in, the hash table over them
    std::vector<int32_t> pack_soup_slot, pack_soup_table;
    std::vector<double> next_weights;    // pdmpc_set_step_weights: expected work per vehicle of the NEXT packed step (the caller's order); consumed by that pack
    PinnedBuf<double> h_lean;            // fetch_lean: (cost, status) per slot
    DevBuf<double> d_lean;
    PinnedBuf<pdmpc_vehicle_out> h_out;  // pdmpc_fetch_results: the records land in pinned memory (a copy into the caller's pageable array goes through the runtime's staging otherwise)
    int bk_ready_launch = 2048;          // entries of the ready list of the last layout
    uint32_t* progress = nullptr;        // pinned, debug_progress
    int n_waves = PDMPC_MAX_WAVES;       // of the last layout
    bool compact_layout = false;         // the last layout is the compact kernel's (two workgroups per CU)
    // batch blobs: several packed steps can stay resident side by side ("banks", pdmpc_select_bank)
    std::vector<PackedStep> banks;
    int bank = 0;
    uint32_t epoch = 1;  // done flags start at 0, so no slot looks solved before its first launch
    // launches
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    double folded_kernel_ms = 0.0;  // launches whose event pairs were recycled (resident launches without a pack or reset in between)
    int64_t folded_launches = 0;
    LdsLayout lds{};
    int NL = 0, NV = 0, areas_in_lds = 0;
    pdmpc_stats stats{};
};

namespace {

// The dynamic LDS size of a kernel is an attribute of the function ON THE DEVICE, not of a handle (hipFuncSetAttribute sets a
// maximum): the largest size set so far is kept per device and kernel (0 bulk, 1 bulk wide, 2 bulk SAT, 3 bulk compact), shared by every handle.
const size_t kMaxLaunchEvents = 4096;  // event pairs a handle keeps before it folds their times (launch_range)
std::mutex g_lds_mutex;
uint32_t g_lds_high_water[64][4];

// hipStreamSynchronize on the launch stream, counted: a bank whose staging copy was queued before is free again (pack_common)
inline hipError_t sync_stream(pdmpc_handle* h) {
    const hipError_t e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) h->sync_serial += 1;
    return e;
}

// LDS layout of the graph search: MPA tables, reference, per-wave tallies, shared words, obstacle soup, phase B's chunk state (12 B per
// thread), d_traveled table, the LDS part of the open set (PDMPC_BK_PER entries per thread), the ready list with its collision
// flags, the histogram / goal list / expansion lists, 2 KB of small tables, the areas of the published path, validity bytes, then
// as many node records as fit.
bool layout_bulk(pdmpc_handle* h, size_t budget, int n_waves, int areas, int soup_cap, LdsLayout& L, uint32_t& nv, uint32_t& nl, uint32_t ready_cap, bool compact) {
    (void)n_waves;
    const uint32_t lk_waves = compact ? PDMPC_LK_COMPACT_WAVES : PDMPC_MAX_WAVES, lk_ready = compact ? PDMPC_LK_COMPACT_READY_CAP : 2048u, lk_per = compact ? PDMPC_LK_COMPACT_BK_PER : PDMPC_BK_PER;
    if (ready_cap > lk_ready || (uint32_t)n_waves > lk_waves) return false;
    // the regions of fixed size at the kernel's compile-time offsets (pdmpc_device.h: PDMPC_LK_*) ...
    // ... the automaton's tables and the soup behind them
    uint32_t off = pdmpc_lk_fixed(lk_waves, lk_ready, lk_per, &L);
    L.mask = off;
    off = align16(off + (uint32_t)h->mask_bytes);
    L.man_index = off;
    off = align16(off + (uint32_t)h->mi_bytes);
    L.pose = off;
    off = align16(off + (uint32_t)(h->n_man * sizeof(DevManPose)));
    L.area = off;
    if (areas) off = align16(off + (uint32_t)(h->n_man * 3 * PDMPC_VMAX * 16));
    L.soup = off;
    off = align16(off + (uint32_t)std::max(soup_cap, 1) * 16);
    L.tree16 = L.bk_hist;  // (the sampled optimizer's region: not part of this layout)
    const uint32_t min_nodes = 64 * (uint32_t)sizeof(NodeRec) + 1024;
    if ((size_t)off + min_nodes + 256 > budget) return false;
    const uint32_t rest = (uint32_t)(budget - off - 256);
    nv = std::min<uint32_t>(16384u, std::max<uint32_t>(compact ? 512u : 1024u, rest / 6));
    nv = std::min(nv, h->max_nodes) & ~15u;
    nl = std::min((rest - nv) / (uint32_t)sizeof(NodeRec), h->max_nodes);
    L.vstate = off;
    off += align16(nv);
    L.nodes = off;
    off += nl * (uint32_t)sizeof(NodeRec);
    L.total = align16(off);
    return L.total <= budget;
}

// helper workgroups serve the launches that leave CUs idle (launch_range)
bool bulk_has_helpers(const pdmpc_handle* h, int n_launch) {
    if (!h->tune.speculate || h->tune.helpers == 0) return false;
    if (n_launch > h->n_cu) return h->tune.helpers_oversub != 0;  // (the tail of a launch with more searches than CUs)
    return n_launch <= h->n_cu - 2;
}

int compute_lds_bulk(pdmpc_handle* h, int n_launch, int soup_cap) {
    // large rounds pay where helper workgroups share them; without helpers the LDS is better spent on node records
    h->bk_ready_launch = bulk_has_helpers(h, n_launch) ? h->tune.ready : std::max(256, h->tune.ready / 2);
    // Sixteen wavefronts where the kernel's registers allow four per SIMD (measured against twelve: C2 +1.5 %, C3 +1.3 %, C4 +7.5 %;
    // C5, five light searches per CU one after the other, -1.7 %: it keeps twelve and the LDS-resident nodes that go with them)
    const int cap = h->cfg.checker == PDMPC_CHECK_SAT ? PDMPC_MAX_WAVES_SAT : PDMPC_MAX_WAVES;
    // Two workgroups per CU (bulk_kernel_compact.hip: 8 wavefronts, at most half the LDS, a near list of 1 024 entries, the automaton's
    // areas in L2) for launches of more than two searches per CU — C5's class: light searches one after the other on every CU, each
    // bound by the latency of its own passes; two side by side fill each other's gaps (C5 557 -> 710 steps/s).  NOT for launches whose
    // step is one heavy search (C4, 512 searches: 94.6 -> 31 steps/s with it — half the lanes, a quarter of the near list, rounds of
    // 240 entries that are never shared).  InterX with one mask word only; falls back to the full layout if the soup does not fit.
    const bool want_compact = h->cfg.checker == PDMPC_CHECK_INTERX && h->n_words == 1 && (h->tune.compact > 0 || (h->tune.compact < 0 && n_launch > 2 * h->n_cu));
    h->compact_layout = false;
    if (want_compact) {
        LdsLayout L{};
        uint32_t nv = 0, nl = 0;
        const int waves = h->tune.waves >= 0 ? std::min(h->tune.waves, PDMPC_LK_COMPACT_WAVES) : PDMPC_LK_COMPACT_WAVES;
        const int ready = std::min(std::min(h->bk_ready_launch, 3 * PDMPC_WAVE * waves), (int)PDMPC_LK_COMPACT_READY_CAP);
        // (compact=2, debugging: the compact kernel with the whole CU's LDS, i.e. ONE workgroup per CU — its layout without the co-residency)
        // (compact=3: a little more than half — still one workgroup per CU, but with the small budget's few LDS-resident nodes)
        if (layout_bulk(h, h->tune.compact == 2 ? kLdsMax : (h->tune.compact == 3 ? kLdsMax / 2 + 4096 : (h->tune.compact == 5 ? kLdsMax / 2 - 4096 : kLdsMax / 2)), waves, 0, soup_cap, L, nv, nl, (uint32_t)ready, true)) {
            if (h->tune.compact == 5) L.total += 4096;  // (debugging: two workgroups per CU with 4 KB of slack behind the layout)
            if (h->tune.compact == 4) L.total += 4096;  // (debugging: the half-CU layout, allocated too large for two workgroups per CU)
            if (h->tune.debug_lds)
                fprintf(stderr, "pdmpc LDS layout (compact): launch %d waves %d near %u ready %d nv %u nl %u total %u\n", n_launch, waves, PDMPC_LK_COMPACT_BK_PER * (uint32_t)waves * PDMPC_WAVE, ready, nv, nl, L.total);
            h->bk_ready_launch = ready;
            h->lds = L;
            h->n_waves = waves;
            h->NL = (int)nl;
            h->NV = (int)nv;
            h->areas_in_lds = 0;
            h->compact_layout = true;
            return PDMPC_OK;
        }
    }
    const int waves = h->tune.waves >= 0 ? std::min(h->tune.waves, cap) : (n_launch > 2 * h->n_cu ? std::min(12, cap) : cap);
    for (int areas = 1; areas >= 0; --areas) {  // (the maneuver areas fall back to L2 when the soup leaves no room)
        LdsLayout L{};
        uint32_t nv = 0, nl = 0;
        const int ready = std::min(h->bk_ready_launch, 3 * PDMPC_WAVE * waves);
        if (!layout_bulk(h, kLdsMax, waves, areas, soup_cap, L, nv, nl, (uint32_t)ready, false)) continue;
        if (h->tune.debug_lds)
            fprintf(stderr, "pdmpc LDS layout: launch %d waves %d areas %d near %u ready %d nv %u nl %u total %u\n", n_launch, waves, areas, PDMPC_BK_PER * (uint32_t)waves * PDMPC_WAVE, ready,
                    nv, nl, L.total);
        h->bk_ready_launch = ready;
        h->lds = L;
        h->n_waves = waves;
        h->NL = (int)nl;
        h->NV = (int)nv;
        h->areas_in_lds = areas;
        return PDMPC_OK;
    }
    char buf[256];
    snprintf(buf, sizeof buf, "obstacle soup (%d columns) + MPA tables do not fit into %zu B of LDS", soup_cap, kLdsMax);
    return fail(PDMPC_ERR_CAPACITY, buf);
}

// LDS layout of the sampled optimizer (one wavefront per vehicle): MPA tables, reference, the wave's two shapes, offsets, obstacle
// soup, the candidate segments of one edge check, and its tree (288 nodes x (16 children + parent + trim) x 2 B).
int compute_lds_sampled(pdmpc_handle* h, int soup_cap, int cand_cap) {
    for (int areas = 1; areas >= 0; --areas) {
        LdsLayout L{};
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
        off += (2 * PDMPC_VMAX + 1) * 16;
        L.path = off;
        off += align16((PDMPC_HP_MAX + 2) * 4 + 2 * (PDMPC_HP_MAX + 1) * 4 + PDMPC_SH_WORDS * 4 + PDMPC_HP_MAX * 4);
        L.soup = off;
        off = align16(off + (uint32_t)std::max(soup_cap, 1) * 16);
        L.cand = off;
        off += align16((uint32_t)std::max(cand_cap, 1) * 4u);
        L.expand = off;
        off += (2 * PDMPC_HP_MAX * PDMPC_HP_MAX) * 8 + 16 * 16;
        L.tree16 = off;
        off += align16(288u * 18u * 2u);
        L.total = align16(off);
        if (L.total > kLdsMax / 2) continue;
        h->lds = L;
        h->n_waves = 1;
        h->NL = 0;
        h->NV = 0;
        h->areas_in_lds = areas;
        return PDMPC_OK;
    }
    return fail(PDMPC_ERR_CAPACITY, "obstacle soup + MPA tables do not fit into the LDS budget of the sampled optimizer");
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
    // The packed batch is written where it is copied from, the bank's pinned blob: [vehicles | predecessor slots | points].  The
    // points come last — their number is known once they are written — and the blob grows with its contents kept.
    size_t total_pred = 0;
    if (pred_offset)
        for (int vi = 0; vi < n; ++vi) total_pred += (size_t)std::max(0, pred_offset[vi + 1] - pred_offset[vi]);
    const size_t veh_bytes = ((size_t)std::max(n, 1) * sizeof(DevVehicle) + 15) & ~(size_t)15;
    const size_t pred_bytes = ((total_pred + 1) * sizeof(int32_t) + 15) & ~(size_t)15;
    const size_t pts_base = veh_bytes + pred_bytes;
    B.n_packed = 0;  // (a pack that fails leaves the bank empty: the batch that was in it is being overwritten)
    B.pack_failed = true;
    B.h_veh = nullptr;
    B.h_pts = nullptr;
    B.h_pred = nullptr;
    if (B.h_blob.ensure_keep(pts_base + 4096, 0)) return fail(PDMPC_ERR_HIP, "hipHostMalloc failed");
    DevVehicle* veh = (DevVehicle*)B.h_blob.p;
    int32_t* pred = (int32_t*)(B.h_blob.p + veh_bytes);
    double* pts = (double*)(B.h_blob.p + pts_base);
    size_t n_pred_out = 0, n_pts = 0;  // entries of pred / POINTS (two doubles each) written
    auto put = [&](double x, double y) {
        pts[2 * n_pts] = x;
        pts[2 * n_pts + 1] = y;
        ++n_pts;
    };
    auto set_points = [](const pdmpc_polygon_set& s) -> size_t {  // points of a (checked) set + one separator per polygon
        return s.n_polygons > 0 ? (size_t)(s.offset[s.n_polygons] - s.offset[0]) + (size_t)s.n_polygons : 0;
    };
    B.lit_cols.assign((size_t)n, 0);
    int soup_cap = 0, cand_cap = 0;
    // Slot order.  A search spins for predecessors of the same launch, so every predecessor must sit in a lower slot than its
    // successors (launch_range: forward progress of oversubscribed launches).  Callers hand the vehicles over in level order
    // (kahn.m); a batch that is not is put into level order here -- computation levels by longest path, stable within a level --
    // and pdmpc_fetch_results hands the records back in the caller's order.
    B.perm.clear();
    B.inv.clear();
    // Priority order (pdmpc_set_step_weights).  Workgroups are handed out in index order, and a launch of more searches than CUs
    // starts its later workgroups when earlier ones end: in level order a heavy search of a late level starts late — behind finished
    // searches that hold their CUs while they wait for predecessors (C4: 2.5-4 ms into a 10 ms step).  With an expected work per
    // vehicle the slots are filled by PRIORITY instead: the largest expected work among a vehicle and its descendants in the coupling
    // DAG, descending; ties by level, then by the caller's index.  A predecessor's priority is at least its successors' and its level
    // is lower, so this is a topological order too — every predecessor in a lower slot: the forward-progress argument holds
    // unchanged — and the records go back in the caller's order as for any batch the library reorders.
    const bool by_priority = h->tune.dispatch_order && (int)h->next_weights.size() == n && n > 1 && pred_offset != nullptr;
    if (pred_offset) {
        bool ordered = true;
        for (int i = 0; i < n && ordered; ++i)
            for (int q = pred_offset[i]; q < pred_offset[i + 1]; ++q) {
                const int ps = pred_index[q];
                if (ps < 0 || ps >= h->max_vehicles) return fail(PDMPC_ERR_INVALID, "predecessor slot out of range");
                if (ps < n && ps >= i) ordered = false;
            }
        if (!ordered || by_priority) {
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
            if (by_priority) {
                std::vector<double> prio((size_t)n);
                for (int i = 0; i < n; ++i) {
                    const double w = h->next_weights[(size_t)i];
                    prio[(size_t)i] = (w == w && w > 0) ? w : 0.0;
                }
                for (size_t qi = queue.size(); qi-- > 0;) {  // (reverse topological order: a vehicle after all its successors)
                    const int u = queue[qi];
                    for (int q = succ_off[u]; q < succ_off[u + 1]; ++q) prio[(size_t)u] = std::max(prio[(size_t)u], prio[(size_t)succ[(size_t)q]]);
                }
                std::stable_sort(B.perm.begin(), B.perm.end(), [&](int32_t x, int32_t y) {
                    if (prio[(size_t)x] != prio[(size_t)y]) return prio[(size_t)x] > prio[(size_t)y];
                    return level[x] < level[y];
                });
            } else {
                std::stable_sort(B.perm.begin(), B.perm.end(), [&](int32_t x, int32_t y) { return level[x] < level[y]; });
            }
            bool identity = true;
            for (int i = 0; i < n && identity; ++i) identity = B.perm[(size_t)i] == i;
            if (identity) {
                B.perm.clear();  // (the caller's order is the order wanted: raw slots are the caller's vehicles)
            } else {
                B.inv.resize((size_t)n);
                for (int sl = 0; sl < n; ++sl) B.inv[(size_t)B.perm[(size_t)sl]] = sl;
            }
        }
    }
    h->next_weights.clear();
    const bool permuted = !B.perm.empty();
    // Vehicles that hand over THE SAME ARRAYS (same pointers, same counts: the prioritization instances of an explorative step share
    // every input but the predecessor lists, PrioritizedExplorativeController.m:25-91; step_controller.cpp builds one set per distinct
    // content) share one copy of their soups in the pool: a vehicle seen before takes over the offsets of the first one.
    // (an open-addressing table over the slots that brought new arrays: a batch of 1 280 slots looks its key up 1 280 times)
    std::vector<SoupKey>& soup_keys = h->pack_soup_keys;  // key of the q-th distinct vehicle, soup_slot[q] the slot it was packed in
    std::vector<int32_t>& soup_slot = h->pack_soup_slot;
    std::vector<int32_t>& soup_table = h->pack_soup_table;  // hash -> q + 1, 0 = empty
    size_t table_size = 64;
    while (table_size < (size_t)n * 2) table_size *= 2;
    soup_keys.clear();
    soup_slot.clear();
    soup_table.assign(table_size, 0);
    auto soup_hash = [](const SoupKey& k) {
        uint64_t hsh = 1469598103934665603ull;
        const uint64_t* w = (const uint64_t*)&k;
        for (size_t q = 0; q < sizeof(SoupKey) / 8; ++q) hsh = (hsh ^ w[q]) * 1099511628211ull;
        return hsh ^ (hsh >> 29);
    };
    static_assert(sizeof(SoupKey) % 8 == 0, "SoupKey is hashed by 64-bit words");
    for (int slot_i = 0; slot_i < n; ++slot_i) {
        const int i = slot_i;  // (slot: index into the packed arrays)
        const int vi = permuted ? B.perm[(size_t)slot_i] : slot_i;  // (the caller's vehicle)
        const pdmpc_vehicle_in& v = in[vi];
        if (!v.ref_x || !v.ref_y || !v.v_ref) return fail(PDMPC_ERR_INVALID, "reference trajectory missing");
        if (v.trim0 < 1 || v.trim0 > h->n_trims) return fail(PDMPC_ERR_INVALID, "trim0 out of range");
        const bool has_fb = fallback && fallback[vi].n_polygons > 0;
        // seen before?  (looked up first: a vehicle that hands over arrays that are packed already needs neither their checks nor room)
        SoupKey key;
        std::memset(&key, 0, sizeof key);
        {
            const pdmpc_polygon_set* fbv = has_fb ? &fallback[vi] : nullptr;
            const void* ptrs[13] = {v.obstacles.offset, v.obstacles.x, v.obstacles.y, v.dynamic_obstacles.offset, v.dynamic_obstacles.x, v.dynamic_obstacles.y, v.hdv_reachable_sets.offset,
                                    v.hdv_reachable_sets.x, v.left_x, v.right_x, fbv ? fbv->offset : nullptr, fbv ? fbv->x : nullptr, fbv ? fbv->y : nullptr};
            for (int q = 0; q < 13; ++q) key.p[q] = ptrs[q];
            key.c[0] = v.obstacles.n_polygons;
            key.c[1] = v.dynamic_obstacles.n_polygons;
            key.c[2] = v.hdv_reachable_sets.n_polygons;
            key.c[3] = v.n_left;
            key.c[4] = v.n_right;
            key.c[5] = fbv ? fbv->n_polygons : 0;
        }
        size_t at_table = (size_t)soup_hash(key) & (table_size - 1);
        int seen_slot = -1;
        while (soup_table[at_table] != 0) {
            const int q = soup_table[at_table] - 1;
            if (std::memcmp(&soup_keys[(size_t)q], &key, sizeof key) == 0) {
                seen_slot = soup_slot[(size_t)q];
                break;
            }
            at_table = (at_table + 1) & (table_size - 1);
        }
        const pdmpc_vehicle_in* first_in = seen_slot >= 0 ? &in[permuted ? B.perm[(size_t)seen_slot] : seen_slot] : nullptr;
        const bool shared = first_in && v.left_y == first_in->left_y && v.right_y == first_in->right_y && v.hdv_reachable_sets.y == first_in->hdv_reachable_sets.y;
        if (!shared) {
            int rc;
            if ((rc = check_set(v.obstacles, "obstacles"))) return rc;
            if ((rc = check_set(v.dynamic_obstacles, "dynamic_obstacles"))) return rc;
            if ((rc = check_set(v.hdv_reachable_sets, "hdv_reachable_sets"))) return rc;
            if (v.dynamic_obstacles.n_polygons % Hp) return fail(PDMPC_ERR_INVALID, "dynamic_obstacles must hold n_d * Hp polygons");
            if (v.hdv_reachable_sets.n_polygons % Hp) return fail(PDMPC_ERR_INVALID, "hdv_reachable_sets must hold n_h * Hp polygons");
            if (v.n_left < 0 || v.n_right < 0 || v.n_left == 1 || v.n_right == 1)
                return fail(PDMPC_ERR_INVALID, "lanelet boundary needs 0 or >= 2 points per side");
            if (has_fb) {
                if (fallback[vi].n_polygons != Hp) return fail(PDMPC_ERR_INVALID, "fallback_shapes must hold Hp polygons per vehicle");
                if ((rc = check_set(fallback[vi], "fallback_shapes"))) return rc;
            }
            // room for everything this vehicle can add (+ the batch's trailing pad)
            const size_t most = (size_t)Hp * set_points(v.obstacles) + set_points(v.dynamic_obstacles) + set_points(v.hdv_reachable_sets) + (size_t)v.n_left + (size_t)v.n_right + 2 +
                                (has_fb ? set_points(fallback[vi]) : 0) + 2;
            if (B.h_blob.ensure_keep(pts_base + (n_pts + most) * 16, pts_base + n_pts * 16)) return fail(PDMPC_ERR_HIP, "hipHostMalloc failed");
            veh = (DevVehicle*)B.h_blob.p;
            pred = (int32_t*)(B.h_blob.p + veh_bytes);
            pts = (double*)(B.h_blob.p + pts_base);
        }
        DevVehicle& d = veh[(size_t)i];
        std::memset(&d, 0, sizeof d);
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
        d.pred_off = (int32_t)n_pred_out;
        for (int q = 0; q < n_pred; ++q) {
            const int ps = pred_index[pred_offset[vi] + q];
            if (ps < 0 || ps >= h->max_vehicles) return fail(PDMPC_ERR_INVALID, "predecessor slot out of range");
            pred[n_pred_out++] = permuted && ps < n ? B.inv[(size_t)ps] : ps;
        }
        if (shared) {
            const DevVehicle& f = veh[(size_t)seen_slot];  // (validated when it was packed)
            std::memcpy(d.lit_off, f.lit_off, sizeof d.lit_off);
            std::memcpy(d.hdv_off, f.hdv_off, sizeof d.hdv_off);
            std::memcpy(d.fb_off, f.fb_off, sizeof d.fb_off);
            d.ll_off = f.ll_off;
            d.ll_len = f.ll_len;
            B.lit_cols[i] = B.lit_cols[(size_t)seen_slot];
            const int need = (d.lit_off[Hp] - d.lit_off[0]) + Hp * n_pred * PDMPC_VMAX + (d.hdv_off[Hp] - d.hdv_off[0]) + d.ll_len;
            soup_cap = std::max(soup_cap, need);
            for (int k = 0; k < Hp; ++k)
                cand_cap = std::max(cand_cap, (d.lit_off[k + 1] - d.lit_off[k]) + n_pred * PDMPC_VMAX + (d.hdv_off[k + 1] - d.hdv_off[k]) + d.ll_len);
            continue;
        }
        if (seen_slot < 0) {  // (a key met again with other y arrays keeps its first entry, as the map did)
            soup_keys.push_back(key);
            soup_slot.push_back(i);
            soup_table[at_table] = (int32_t)soup_keys.size();
        }
        const int n_dyn = v.dynamic_obstacles.n_polygons / Hp;
        const int n_hdv = v.hdv_reachable_sets.n_polygons / Hp;
        auto append_poly = [&](const pdmpc_polygon_set& s, int p, bool sep) {
            for (int q = s.offset[p]; q < s.offset[p + 1]; ++q) put(s.x[q], s.y[q]);
            if (sep) put(qnan, qnan);
        };
        int need = 0;
        // vehicle_obstacles{k} = [static..., dynamic(:, k)...], each followed by [NaN; NaN]   vectorize_all_obstacles.m:36-62
        for (int k = 0; k < Hp; ++k) {
            d.lit_off[k] = (int32_t)n_pts;
            for (int p = 0; p < v.obstacles.n_polygons; ++p) append_poly(v.obstacles, p, true);
            for (int r = 0; r < n_dyn; ++r) append_poly(v.dynamic_obstacles, r * Hp + k, true);
            need += (int)n_pts - d.lit_off[k] + n_pred * PDMPC_VMAX;
        }
        d.lit_off[Hp] = (int32_t)n_pts;
        B.lit_cols[i] = d.lit_off[Hp] - d.lit_off[0];
        for (int k = 0; k < Hp; ++k) {
            d.hdv_off[k] = (int32_t)n_pts;
            for (int r = 0; r < n_hdv; ++r) append_poly(v.hdv_reachable_sets, r * Hp + k, true);
        }
        d.hdv_off[Hp] = (int32_t)n_pts;
        need += d.hdv_off[Hp] - d.hdv_off[0];
        // lanelet_boundary = [left, NaN, right, NaN]                                          vectorize_all_obstacles.m:27-30
        d.ll_off = (int32_t)n_pts;
        for (int q = 0; q < v.n_left; ++q) put(v.left_x[q], v.left_y[q]);
        put(qnan, qnan);
        for (int q = 0; q < v.n_right; ++q) put(v.right_x[q], v.right_y[q]);
        put(qnan, qnan);
        d.ll_len = (int32_t)n_pts - d.ll_off;
        need += d.ll_len;
        B.lit_cols[i] += d.ll_len;
        if (has_fb) {
            for (int k = 0; k < Hp; ++k) {
                d.fb_off[k] = (int32_t)n_pts;
                if (fallback[vi].offset[k + 1] - fallback[vi].offset[k] > PDMPC_VMAX)
                    return fail(PDMPC_ERR_INVALID, "fallback area has more than PDMPC_VMAX columns");
                append_poly(fallback[vi], k, false);
            }
            d.fb_off[Hp] = (int32_t)n_pts;
        } else {
            for (int k = 0; k <= Hp; ++k) d.fb_off[k] = -1;
        }
        soup_cap = std::max(soup_cap, need);
        for (int k = 0; k < Hp; ++k)
            cand_cap = std::max(cand_cap, (d.lit_off[k + 1] - d.lit_off[k]) + n_pred * PDMPC_VMAX + (d.hdv_off[k + 1] - d.hdv_off[k]) + d.ll_len);
    }
    // a trailing pad so 16-byte staged copies never run past the allocation (room: the blob's first 4096 bytes of points, or a vehicle's)
    put(qnan, qnan);
    pred[n_pred_out++] = 0;
    B.soup_cap = soup_cap + 2;
    B.cand_cap = (cand_cap + 4 + 3) & ~3;
    const size_t pts_bytes = (n_pts * 16 + 15) & ~(size_t)15;
    const size_t total = pts_base + pts_bytes;
    if (B.d_blob.ensure(total)) return fail(PDMPC_ERR_HIP, "hipMalloc failed for the batch blob");
    B.h_veh = veh;
    B.h_pred = pred;
    B.h_pts = pts;
    B.d_veh = (DevVehicle*)B.d_blob.p;
    B.d_pred = (int32_t*)(B.d_blob.p + veh_bytes);
    B.d_pts = (double*)(B.d_blob.p + pts_base);
    // one copy, not waited for: whatever the stream does next is ordered behind it, and the next pack into this bank waits (above)
    HIPCHK(hipMemcpyAsync(B.d_blob.p, B.h_blob.p, total, hipMemcpyHostToDevice, h->stream));
    B.staged_serial = h->sync_serial;
    B.n_packed = n;
    B.pack_failed = false;
    h->events_used = 0;
    h->folded_kernel_ms = 0.0;
    h->folded_launches = 0;
    std::memset(&h->stats, 0, sizeof h->stats);
    return PDMPC_OK;
}

// per-vehicle arenas for `nodes` tree nodes each (contents are scratch: every search starts from an empty tree)
const size_t kArenaBytesPerNode = sizeof(NodeRec) + 8 + 8 + 1 + 8 + 4 + 8 + 4 + 8 + 4 + 16 + 4;
int alloc_arenas(pdmpc_handle* h, uint32_t nodes) {
    nodes = (nodes + 1u) & ~1u;
    const size_t tot = (size_t)h->max_vehicles * nodes;
    h->anodes.release();
    h->akey.release();
    h->alink.release();
    h->avs.release();
    h->afark.release();
    h->afari.release();
    h->amidk.release();
    h->amidi.release();
    h->apbk.release();
    h->apbd.release();
    h->awalk.release();
    h->achild0.release();
    h->max_nodes = 0;
    int bad = 0;
    bad |= h->anodes.ensure_exact(tot) | h->akey.ensure_exact(tot) | h->alink.ensure_exact(tot) | h->avs.ensure_exact(tot) | h->afark.ensure_exact(tot) | h->afari.ensure_exact(tot);
    bad |= h->amidk.ensure_exact(tot) | h->amidi.ensure_exact(tot) | h->apbk.ensure_exact(tot) | h->apbd.ensure_exact(tot) | h->awalk.ensure_exact(2 * tot) | h->achild0.ensure_exact(tot);
    if (bad) return bad;
    h->max_nodes = nodes;
    return 0;
}

// safe == true: the recovery path after a predecessor time-out (plan_packed_growing): slices that are resident as a whole,
// no helper workgroups next to an oversubscribed launch, the default spin limit.
int launch_range(pdmpc_handle* h, int first, int count, bool safe = false) {
    if (!h) return fail(PDMPC_ERR_INVALID, "null handle");
    PackedStep& B = h->banks[h->bank];
    if (B.pack_failed) return fail(PDMPC_ERR_INVALID, "the last pack into this bank failed: nothing is packed");
    if (first < 0 || count < 0 || first + count > B.n_packed) return fail(PDMPC_ERR_INVALID, "launch range outside the packed batch");
    if (!B.perm.empty() && (first != 0 || count != B.n_packed)) return fail(PDMPC_ERR_INVALID, "range launches need a batch packed in level order (predecessors in lower slots)");
    if (count == 0) return PDMPC_OK;
    const Tuning& T = h->tune;
    const bool search = !h->sampled_launch;
    int rc = search ? compute_lds_bulk(h, count, B.soup_cap) : compute_lds_sampled(h, B.soup_cap, B.cand_cap);
    if (rc) return rc;
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
    a.arena.key = h->akey.p;
    a.arena.link = h->alink.p;
    a.arena.vstate = h->avs.p;
    a.arena.far_key = h->afark.p;
    a.arena.far_id = h->afari.p;
    a.arena.mid_key = h->amidk.p;
    a.arena.mid_id = h->amidi.p;
    a.arena.pb_key = h->apbk.p;
    a.arena.pb_d = h->apbd.p;
    a.arena.walk = h->awalk.p;
    a.arena.child0 = h->achild0.p;
    a.max_nodes = h->max_nodes;
    a.tree_size = h->d_tree_size.p;
    a.lds = h->lds;
    a.NL = h->NL;
    a.NV = h->NV;
    a.soup_cap = B.soup_cap;
    a.cand_cap = B.cand_cap;
    a.n_waves = h->n_waves;
    a.tie_count = h->d_tie_count.p;
    a.work_count = h->d_work_count.p;
    a.sampled_random = h->d_random.p;
    a.sampled_n_random = h->sampled_n_random;
    a.spin_limit = safe ? (1u << 22) : T.spin_limit;
    a.debug_tail = T.debug_tail;
    if (T.debug_progress && !h->progress) {
        if (hipHostMalloc((void**)&h->progress, (size_t)h->max_vehicles * 64 * 4, hipHostMallocMapped) != hipSuccess) h->progress = nullptr;
        if (h->progress) std::memset(h->progress, 0, (size_t)h->max_vehicles * 64 * 4);
    }
    a.progress = h->progress;
    a.speculate = T.speculate;
    a.reverse_dispatch = (!safe && T.reverse_dispatch) ? 1 : 0;

    // rounds: measured on C2 / C3 (20 / 128 searches, helpers): cap 256, ramp 4 -> 646 / 589 steps/s; 512, 2 -> 735 / 786; 1000, 2 -> 769 / 909; 1000, 1 -> 620 / 772
    const bool helped = search && !safe && bulk_has_helpers(h, count);
    a.bk_ready_cap = std::min(h->bk_ready_launch, 3 * PDMPC_WAVE * h->n_waves);  // (the verdict pass handles three entries per thread)
    a.bk_round0 = T.round0 > 0 ? T.round0 : 24;  // (C3's class: below, once the helpers are counted)
    a.bk_round = std::min(h->bk_ready_launch / 2 - 16, std::max(a.bk_round0, T.round > 0 ? T.round : (helped ? 1000 : 256)));
    a.bk_ramp = T.ramp > 0 ? T.ramp : (helped ? 2 : 4);
    a.bk_flags = (T.fast_select ? 1 : 0) | (T.lazy_verify ? 2 : 0) | ((std::min(8, std::max(1, T.poll_every)) - 1) << 2);
    a.bk_mid_min = T.mid_min;
    a.bk_mid_fill = T.mid_fill;
    a.bk_tile = T.tile > 0 ? T.tile : 256;
    if (h->compact_layout) a.bk_tile = std::min(a.bk_tile, 256);  // (a helper stages its range in the near list's room: 12 KB in the compact layout)
    a.bk_tentative = T.tentative;
    a.bk_fast_arrival = T.fast_arrival;
    a.bk_seat_nodes = std::max(1, T.seat_nodes);
    a.bk_force_tie = T.force_tie;
    a.bk_post = h->d_bk_post.p;
    a.help_board = h->d_help_board.p;
    a.help_verdict = h->d_help_verdict.p;
    a.help_finished = h->d_help_finished.p;
    a.help_fin_base = 0;
    h->launch_serial += 1;
    if (h->launch_serial == 0) h->launch_serial = 1;
    a.launch_id = h->launch_serial;
    // Helper workgroups: the trailing workgroups of the launch, on the CUs it leaves idle, check tiles of the searches' large rounds.
    // A launch with more searches than CUs gets them for its tail, when CUs fall idle while a few long searches still run (measured on
    // C4, 512 searches: none 25.6 steps/s, 32 helpers 41.5, 96: 42.8-45.9; C5, 1 280 searches: none 478 steps/s, 64 behind the searches 543,
    // 200: 549, with rounds shared from 64 nodes on 560 — the last levels' searches, which run when the CUs fall idle, are the tail of
    // the step).  In the safe mode a launch gets none: they would sit where a slice's search could run.
    a.n_searches = count;
    a.n_helpers = 0;
    if (helped) {
        if (count <= h->n_cu) {
            // every CU the launch leaves idle: a seated helper polls a word of its own, so helpers cost the searches nothing (measured on
            // C2, 20 searches: 32 helpers 1 110 steps/s, 64: 1 120, 96: 1 190, 128: 1 200, 200: 1 210, 230: 1 235; with the ticket word of
            // rounds 3-4 that all helpers polled and claimed from, 64 helpers were slower than 32 and 200 cost 40 %)
            int want = h->n_cu - count;
            if (T.helpers >= 0) want = T.helpers;
            a.n_helpers = std::max(0, std::min(want, h->n_cu - count));
            if (a.n_helpers < 2) a.n_helpers = 0;
        } else {
            a.n_helpers = 200;  // (seated helpers cost the searches nothing: measured on C4 96 -> 76.9 steps/s, 160-250 -> 77.7)
            if (T.helpers_oversub >= 0) a.n_helpers = std::min(T.helpers_oversub, 3 * h->n_cu);
            if (T.helpers >= 0) a.n_helpers = std::min(a.n_helpers, T.helpers);
        }
    }
    // ... and where do they sit?  Behind the searches they get the CUs the searches leave.  In a launch of more searches than CUs whose
    // searches wait for one another, finished searches hold their CUs until their predecessors are through, and the helpers behind them
    // start when the step is half over (C4: a helper lived 5-6 of the step's 11 ms, and the step's 10^5-node search ran most of its
    // rounds with fewer than eight seats).  Half the CUs' worth of helpers in front of the searches: C4 82.8 -> 90.2 steps/s (32: 84.2,
    // 64: 86.5, 128: 90.2, 160: 88.0, 200: 54.6).  A launch of independent searches keeps every CU for them.
    // (handles that share a device — the logical ranks of a group — launch side by side: the idle CUs are the device's, not the
    // launch's, and helper workgroups in FRONT of every launch's searches would fill the chip before any search starts)
    if (h->device_share > 1) a.n_helpers = a.n_helpers / h->device_share >= 2 ? a.n_helpers / h->device_share : 0;
    a.bk_helpers_first = 0;
    if (count > h->n_cu && a.n_helpers > 0 && h->device_share == 1) {
        int want = T.helpers_first;
        if (want < 0 && count > 2 * h->n_cu) want = 0;  // (five searches per CU, C5: the searches need every CU — 128 in front 331 steps/s, 32: 515, none: 560)
        if (want < 0) {
            int chained = 0;
            const DevVehicle* hv = h->banks[h->bank].h_veh + first;
            for (int i = 0; i < count; ++i) chained += hv[i].n_pred > 0 ? 1 : 0;
            want = 2 * chained >= count ? h->n_cu / 2 : 0;
        }
        a.bk_helpers_first = std::max(0, std::min(want, a.n_helpers));
    }
    // rounds are shared from 64 nodes on where helpers are plenty (C2: a dozen per search), from a few hundred on where there are
    // about as many helpers as searches or fewer (measured C3, 128 + 128: 64 -> 1 026 steps/s, 128-192 -> 1 070, 384 -> 986; C4, 512 + 96:
    // 64 -> 65.5, 192 -> 67, 512 -> 69; C5, 1 280 + 200 behind the searches, whose helpers only meet the medium searches of the tail: 32-128 -> 560)
    a.bk_share_min = T.share_min > 0 ? T.share_min : (a.n_helpers >= 4 * count ? 64 : (count <= h->n_cu ? 128 : (count <= 2 * h->n_cu ? 512 : 64)));
    // (sixteen wavefronts: C3 — 128 searches + 128 helpers — 1 135 steps/s with young rounds of 24 nodes and sharing from 160 on, 1 175 with 32
    // and 128; 28: 1 154, 36: 1 137.  C2 / C4 / C5 with 32: -0.6 % / -1 % / +0.6 %: they stay at 24)
    if (T.round0 < 0 && helped && ((count <= h->n_cu && a.n_helpers < 4 * count) || count > 2 * h->n_cu)) {  // (C5, 1 280 searches: 558 -> 565)
        a.bk_round0 = 32;
        a.bk_round = std::max(a.bk_round, a.bk_round0);
    }
    if (a.n_helpers > 0 && !h->boards_dirty) {
        // The boards stay closed between launches (a search closes every round it shares before it uses the verdicts, and a closed
        // ticket word offers nothing) and the count of finished searches runs on from launch to launch: nothing to clear -- two
        // memset dispatches less per launch.  A launch that ended with a watchdog status marks them dirty and the next one clears them.
        a.help_fin_base = h->help_fin_total;
        h->help_fin_total += (uint32_t)count;
    } else if (a.n_helpers > 0) {
        HIPCHK(hipMemsetAsync(h->d_help_board.p, 0, (size_t)h->max_vehicles * PDMPC_HB_WORDS * sizeof(unsigned long long), h->stream));
        HIPCHK(hipMemsetAsync(h->d_help_finished.p, 0, 16 * sizeof(uint32_t), h->stream));
        h->help_fin_total = (uint32_t)count;
        h->boards_dirty = false;
    }
    if (h->events_used == kMaxLaunchEvents) {
        // a caller that launches resident banks for ever (no pack, no pdmpc_reset_stats in between) must not make the handle hold an
        // event pair per launch: the pairs' times are folded into a sum and the pairs used again
        HIPCHK(hipStreamSynchronize(h->stream));
        for (size_t i = 0; i < h->events_used; ++i) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, h->events[i].first, h->events[i].second) == hipSuccess) h->folded_kernel_ms += t;
        }
        h->folded_launches += (int64_t)h->events_used;
        h->events_used = 0;
    }
    if (h->events_used == h->events.size()) {
        hipEvent_t e0, e1;
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        h->events.emplace_back(e0, e1);
    }
    auto& ev = h->events[h->events_used++];
    HIPCHK(hipEventRecord(ev.first, h->stream));
    h->last_launch_search = search;
    h->last_first = first;
    h->last_count = count;
    h->last_safe = safe;
    // Oversubscribed launches (more searches than CUs).  A resident search spins for predecessors of the same launch; slots are in
    // level order (pack_common sees to it), so as long as the hardware hands out workgroups in index order every predecessor was
    // dispatched before its successors and the launch cannot stall.  That order is not a documented guarantee: should a launch ever
    // stall, the watchdog (spin_limit) ends the waiting searches with an error status and plan_packed_growing plans the call again
    // with safe == true, in slices that are resident as a whole (a slice's predecessors are in it or in an earlier slice) -- forward
    // progress then needs no assumption at all.
    const int variant = h->compact_layout ? 3 : (h->cfg.checker == PDMPC_CHECK_SAT ? 2 : (h->n_words != 1 ? 1 : 0));
    auto launch_search = [&](const KernelArgs* ka, int cnt) -> int {
        std::lock_guard<std::mutex> lock(g_lds_mutex);
        uint32_t* hw = &g_lds_high_water[h->cfg.device & 63][variant];
        if (variant == 3) return pdmpc_launch_bulk_compact(ka, cnt, (void*)h->stream, hw);
        if (variant == 2) return pdmpc_launch_bulk_sat(ka, cnt, (void*)h->stream, hw);
        if (variant == 1) return pdmpc_launch_bulk_wide(ka, cnt, (void*)h->stream, hw);
        return pdmpc_launch_bulk(ka, cnt, (void*)h->stream, hw);
    };
    int lrc = 0;
    if (!search) {
        lrc = pdmpc_launch_sampled(&a, count, (void*)h->stream);
    } else if (safe && count > h->n_cu) {
        for (int done = 0; done < count && lrc == 0; done += h->n_cu) {
            KernelArgs part = a;
            part.first = first + done;
            part.n_searches = std::min(h->n_cu, count - done);
            lrc = launch_search(&part, part.n_searches);
        }
    } else {
        lrc = launch_search(&a, count);
    }
    if (lrc != 0) {
        h->boards_dirty = true;  // (the searches that were to count themselves finished never ran: the next launch starts from cleared counters)
        char buf[256];
        snprintf(buf, sizeof buf, "kernel launch failed: %s (LDS %u B)", hipGetErrorString((hipError_t)lrc), h->lds.total);
        return fail(PDMPC_ERR_HIP, buf);
    }
    HIPCHK(hipEventRecord(ev.second, h->stream));
    h->stats.lds_bytes = h->lds.total;
    h->stats.lds_nodes = h->NL;
    return PDMPC_OK;
}

}  // namespace

extern "C" {

const char* pdmpc_last_error(void) { return g_err.c_str(); }
void pdmpc_set_last_error(const char* msg) { g_err = msg ? msg : ""; }  // (group.cpp reports through the same string)
const char* pdmpc_version(void) { return "pdmpc-hip 0.1 (gfx950)"; }

int pdmpc_create(const pdmpc_config* config, pdmpc_handle** out_handle) {
    if (!config || !out_handle) return fail(PDMPC_ERR_INVALID, "null argument");
    if (config->Hp < 1 || config->Hp > PDMPC_HP_MAX) return fail(PDMPC_ERR_INVALID, "Hp must be in 1..PDMPC_HP_MAX");
    if (config->checker != PDMPC_CHECK_SAT && config->checker != PDMPC_CHECK_INTERX) return fail(PDMPC_ERR_INVALID, "unknown checker");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(PDMPC_ERR_NO_DEVICE, "no HIP device visible: this backend has no CPU fallback");
    if (config->device < 0 || config->device >= ndev) return fail(PDMPC_ERR_NO_DEVICE, "device ordinal out of range");
    ON_DEVICE(config->device);
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, config->device));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        return fail(PDMPC_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    pdmpc_handle* h = new pdmpc_handle();
    h->cfg = *config;
    h->banks.resize(1);
    {
        std::string err;
        if (!parse_tuning(getenv("PDMPC_TUNING"), h->tune, err)) {
            delete h;
            return fail(PDMPC_ERR_INVALID, err);
        }
    }
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const uint32_t want_nodes = config->max_nodes > 0 ? (uint32_t)config->max_nodes : 32768u;  // default arena: 256 x 32768 nodes, about 0.6 GB
    h->max_vehicles = config->max_vehicles > 0 ? config->max_vehicles : 256;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return fail(PDMPC_ERR_HIP, "hipStreamCreate failed");
    }
    int bad = alloc_arenas(h, want_nodes);
    bad |= h->d_out.ensure((size_t)h->max_vehicles) | h->d_flag.ensure((size_t)h->max_vehicles) | h->d_tree_size.ensure((size_t)h->max_vehicles) | h->d_tie_count.ensure(4) | h->d_work_count.ensure(16);
    bad |= h->d_help_board.ensure((size_t)h->max_vehicles * PDMPC_HB_WORDS) | h->d_help_verdict.ensure((size_t)h->max_vehicles * PDMPC_HELP_CAP) | h->d_help_finished.ensure(16);
    bad |= h->d_bk_post.ensure((size_t)h->max_vehicles * (size_t)h->tune.ready * 6);
    if (bad) {
        pdmpc_destroy(h);
        return fail(PDMPC_ERR_HIP, "hipMalloc failed for the per-vehicle arenas (lower max_nodes / max_vehicles)");
    }
    (void)hipMemsetAsync(h->d_flag.p, 0, h->d_flag.cap * sizeof(uint32_t), h->stream);
    (void)hipMemsetAsync(h->d_tree_size.p, 0, h->d_tree_size.cap * sizeof(int32_t), h->stream);
    (void)hipMemsetAsync(h->d_tie_count.p, 0, 4 * sizeof(int32_t), h->stream);
    (void)hipMemsetAsync(h->d_work_count.p, 0, 16 * sizeof(unsigned long long), h->stream);
    (void)hipMemsetAsync(h->d_out.p, 0, h->d_out.cap * sizeof(pdmpc_vehicle_out), h->stream);
    (void)hipStreamSynchronize(h->stream);
    *out_handle = h;
    return PDMPC_OK;
}

int pdmpc_destroy(pdmpc_handle* h) {
    if (!h) return PDMPC_OK;
    DeviceGuard device_guard__(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto& ev : h->events) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    h->d_mask.release();
    h->d_mi.release();
    h->d_pose.release();
    h->d_area.release();
    h->anodes.release();
    h->akey.release();
    h->alink.release();
    h->avs.release();
    h->afark.release();
    h->afari.release();
    h->amidk.release();
    h->amidi.release();
    h->apbk.release();
    h->apbd.release();
    h->awalk.release();
    h->achild0.release();
    h->d_out.release();
    h->h_out.release();
    h->h_lean.release();
    h->d_lean.release();
    h->d_flag.release();
    h->d_tree_size.release();
    h->d_tie_count.release();
    h->d_work_count.release();
    h->d_help_board.release();
    h->d_help_verdict.release();
    h->d_bk_post.release();
    h->d_help_finished.release();
    h->d_random.release();
    for (auto& b : h->banks) b.release();
    if (h->progress) (void)hipHostFree(h->progress);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PDMPC_OK;
}

int pdmpc_upload_mpa(pdmpc_handle* h, const pdmpc_mpa* mpa) {
    if (!h || !mpa) return fail(PDMPC_ERR_INVALID, "null argument");
    if (mpa->n_trims < 1 || mpa->n_trims > 1023) return fail(PDMPC_ERR_INVALID, "n_trims must be in 1..1023");
    if (mpa->Hp < h->cfg.Hp) return fail(PDMPC_ERR_INVALID, "mpa.Hp smaller than config.Hp");
    if (!mpa->transition || !mpa->maneuver_index || (mpa->n_maneuvers > 0 && !mpa->maneuvers)) return fail(PDMPC_ERR_INVALID, "null table");
    ON_DEVICE(h->cfg.device);
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
    if (!h) return fail(PDMPC_ERR_INVALID, "null handle");
    ON_DEVICE(h->cfg.device);
    return pack_common(h, n, in, nullptr, nullptr, nullptr);
}

int pdmpc_pack_step(pdmpc_handle* h, int32_t n, const pdmpc_vehicle_in* in, const int32_t* pred_offset, const int32_t* pred_index,
                    const pdmpc_polygon_set* fallback_shapes) {
    if (!h) return fail(PDMPC_ERR_INVALID, "null handle");
    ON_DEVICE(h->cfg.device);
    if (pred_offset && !pred_index) return fail(PDMPC_ERR_INVALID, "pred_index missing");
    return pack_common(h, n, in, pred_offset, pred_index, fallback_shapes);
}

int pdmpc_set_step_weights(pdmpc_handle* h, int32_t n, const double* weights) {
    if (!h || n < 0 || (n > 0 && !weights)) return fail(PDMPC_ERR_INVALID, "pdmpc_set_step_weights: bad argument");
    h->next_weights.assign(weights, weights + n);
    return PDMPC_OK;
}

int pdmpc_launch_packed(pdmpc_handle* h) {
    if (!h) return fail(PDMPC_ERR_INVALID, "null handle");
    ON_DEVICE(h->cfg.device);
    h->epoch += 1;  // a new step: results of earlier launches no longer satisfy predecessor waits
    return launch_range(h, 0, h->banks[h->bank].n_packed, h->safe_launches);
}

int pdmpc_set_device_share(pdmpc_handle* h, int32_t n_handles) {
    if (!h || n_handles < 1) return fail(PDMPC_ERR_INVALID, "pdmpc_set_device_share: bad argument");
    h->device_share = n_handles;
    return PDMPC_OK;
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
    ON_DEVICE(h->cfg.device);
    HIPCHK(hipStreamSynchronize(h->stream));
    h->events_used = 0;
    h->folded_kernel_ms = 0.0;
    h->folded_launches = 0;
    HIPCHK(hipMemsetAsync(h->d_tie_count.p, 0, 4 * sizeof(int32_t), h->stream));
    HIPCHK(hipMemsetAsync(h->d_work_count.p, 0, 16 * sizeof(unsigned long long), h->stream));
    return PDMPC_OK;
}

int pdmpc_launch_range(pdmpc_handle* h, int32_t first, int32_t count) {
    if (!h) return fail(PDMPC_ERR_INVALID, "null handle");
    ON_DEVICE(h->cfg.device);
    return launch_range(h, first, count, h->safe_launches);
}

int pdmpc_synchronize(pdmpc_handle* h) {
    if (!h) return fail(PDMPC_ERR_INVALID, "null handle");
    ON_DEVICE(h->cfg.device);
    HIPCHK(sync_stream(h));
    return PDMPC_OK;
}

int pdmpc_fetch_results(pdmpc_handle* h, int32_t n, pdmpc_vehicle_out* out) {
    if (!h || (n > 0 && !out)) return fail(PDMPC_ERR_INVALID, "null argument");
    if (n < 0 || n > h->max_vehicles) return fail(PDMPC_ERR_INVALID, "bad record count");
    ON_DEVICE(h->cfg.device);
    PackedStep& B = h->banks[h->bank];
    const bool permuted = !B.perm.empty();
    if (permuted && n != B.n_packed) return fail(PDMPC_ERR_INVALID, "a batch that pdmpc_pack_step put into level order is fetched as a whole");
    if (h->h_out.ensure((size_t)std::max(n, 1))) return fail(PDMPC_ERR_HIP, "hipHostMalloc failed");
    if (n > 0) HIPCHK(hipMemcpyAsync(h->h_out.p, h->d_out.p, (size_t)n * sizeof(pdmpc_vehicle_out), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(sync_stream(h));
    // counters + SURVEY.md 8(d) algorithmic bytes of one pass over the packed batch (read where the records landed: pinned memory, slot order)
    const pdmpc_vehicle_out* rec = h->h_out.p;
    pdmpc_stats& s = h->stats;
    const int Hp = h->cfg.Hp;
    const int m = std::min(n, B.n_packed);
    s.n_vehicles = m;
    s.nodes_popped = s.nodes_generated = s.obstacle_columns = 0;
    int64_t bytes = h->mpa_alg_bytes;
    for (int i = 0; i < m; ++i) {
        const pdmpc_vehicle_out& o = rec[i];
        const DevVehicle& d = B.h_veh[i];
        int64_t cols = B.lit_cols[i];
        for (int q = 0; q < d.n_pred; ++q) {
            const int ps = B.h_pred[d.pred_off + q];
            if (ps < n)
                for (int k = 0; k < Hp; ++k) cols += rec[ps].shape_cols[k] + 1;
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
    // ONE pass from pinned memory into the caller's array, in the caller's order (the tree_path ids are per search: nothing else refers to slots)
    if (permuted) {
        for (int sl = 0; sl < n; ++sl) out[B.perm[(size_t)sl]] = rec[sl];
    } else if (n > 0) {
        std::memcpy(out, rec, (size_t)n * sizeof(pdmpc_vehicle_out));
    }
    return PDMPC_OK;
}

namespace {
// What a caller that keeps only a few of a batch's plans needs of ALL of them (the explorative step: the choice among the
// prioritizations rests on the cost-to-come of every vehicle's final node, PrioritizedExplorativeController.m:94-112): status and
// path_nodes[Hp][4] per vehicle, as two strided copies — 12 bytes per record instead of 2.9 KB.  Caller's order.
int fetch_lean(pdmpc_handle* h, int32_t n, int32_t* status, double* cost) {
    PackedStep& B = h->banks[h->bank];
    const bool permuted = !B.perm.empty();
    if (permuted && n != B.n_packed) return fail(PDMPC_ERR_INVALID, "a batch that pdmpc_pack_step put into its own order is fetched as a whole");
    if (h->h_lean.ensure((size_t)std::max(n, 1) * 2)) return fail(PDMPC_ERR_HIP, "hipHostMalloc failed");
    if (h->d_lean.ensure((size_t)std::max(n, 1) * 2)) return fail(PDMPC_ERR_HIP, "hipMalloc failed");
    if (n > 0) {
        const int lrc = pdmpc_launch_gather_lean(h->d_out.p, n, h->cfg.Hp, h->d_lean.p, (void*)h->stream);
        if (lrc) return fail(PDMPC_ERR_HIP, "gather kernel launch failed");
        HIPCHK(hipMemcpyAsync(h->h_lean.p, h->d_lean.p, (size_t)n * 16, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(sync_stream(h));
    for (int sl = 0; sl < n; ++sl) {
        const int v = permuted ? B.perm[(size_t)sl] : sl;
        cost[v] = h->h_lean.p[2 * (size_t)sl];
        std::memcpy(&status[v], &h->h_lean.p[2 * (size_t)sl + 1], sizeof(int32_t));
    }
    return PDMPC_OK;
}
}  // namespace

int pdmpc_fetch_records_at(pdmpc_handle* h, int32_t count, const int32_t* vehicles, pdmpc_vehicle_out* out) {
    if (!h || count < 0 || (count > 0 && (!vehicles || !out))) return fail(PDMPC_ERR_INVALID, "null argument");
    ON_DEVICE(h->cfg.device);
    PackedStep& B = h->banks[h->bank];
    if (h->h_out.ensure((size_t)std::max(count, 1))) return fail(PDMPC_ERR_HIP, "hipHostMalloc failed");
    for (int i = 0; i < count; ++i) {
        const int v = vehicles[i];
        if (v < 0 || v >= B.n_packed) return fail(PDMPC_ERR_INVALID, "vehicle index outside the packed batch");
        const int sl = B.perm.empty() ? v : B.inv[(size_t)v];
        HIPCHK(hipMemcpyAsync(h->h_out.p + i, h->d_out.p + sl, sizeof(pdmpc_vehicle_out), hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(sync_stream(h));
    if (count > 0) std::memcpy(out, h->h_out.p, (size_t)count * sizeof(pdmpc_vehicle_out));
    return PDMPC_OK;
}

namespace {
// The reference's tree grows without bound (Tree.m:54-70); the arenas here are finite.  A call whose search outgrows them
// is planned again from scratch with arenas twice as large (searches are deterministic, so the vehicles that did fit
// produce the same records again) until it fits, the limit set with pdmpc_set_arena_limit is reached, or HBM runs out.
int plan_packed_growing(pdmpc_handle* h, int32_t n, pdmpc_vehicle_out* out, int32_t* lean_status = nullptr, double* lean_cost = nullptr) {
    bool safe = h->safe_launches;
    for (;;) {
        const bool dbg = h->tune.debug_host == 1;
        if (dbg) fprintf(stderr, "pdmpc: launching %d vehicles, arena %u nodes%s\n", n, h->max_nodes, safe ? " (resident slices)" : "");
        ON_DEVICE(h->cfg.device);
        h->epoch += 1;  // a new step: results of earlier launches no longer satisfy predecessor waits
        const auto t0 = std::chrono::steady_clock::now();
        int rc = launch_range(h, 0, h->banks[h->bank].n_packed, safe);
        if (rc) return rc;
        const auto t1 = std::chrono::steady_clock::now();
        rc = out ? pdmpc_fetch_results(h, n, out) : fetch_lean(h, n, lean_status, lean_cost);
        if (rc) return rc;
        h->last_us[1] += std::chrono::duration<double, std::micro>(t1 - t0).count();
        h->last_us[2] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
        if (h->tune.debug_host == 2) {  // (PDMPC_DEBUG_HOST=2: where a call's host time goes, printed by pdmpc_plan_step_literal)
            h->dbg_us[1] += std::chrono::duration<double, std::micro>(t1 - t0).count();
            h->dbg_us[2] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
            float ms = 0.f;
            if (h->events_used > 0 && hipEventElapsedTime(&ms, h->events[h->events_used - 1].first, h->events[h->events_used - 1].second) == hipSuccess) h->dbg_us[3] += 1e3 * ms;
        }
        if (dbg) fprintf(stderr, "pdmpc: fetched, status[0] %d\n", n > 0 ? (out ? out[0].status : lean_status[0]) : 0);
        bool overflow = false, timed_out = false;
        for (int i = 0; i < n; ++i) {
            const int st = out ? out[i].status : lean_status[i];
            overflow = overflow || st == PDMPC_ARENA_OVERFLOW;
            timed_out = timed_out || st == PDMPC_ERR_HIP;
        }
        if (timed_out && safe)
            return fail(PDMPC_ERR_HIP, "a search gave up waiting for a predecessor although the call was planned in resident slices without helper workgroups (records carry PDMPC_ERR_HIP)");
        if (timed_out && !safe) {
            // A search gave up waiting for a predecessor of the same launch (the kernel's watchdog): the launch was
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
        const size_t per_node = kArenaBytesPerNode;
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
    h->last_us[0] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    h->last_us[1] = h->last_us[2] = 0;
    return plan_packed_growing(h, n, out);
}

int pdmpc_plan_step(pdmpc_handle* h, int32_t n, const pdmpc_vehicle_in* in, const int32_t* pred_offset, const int32_t* pred_index,
                    const pdmpc_polygon_set* fallback_shapes, pdmpc_vehicle_out* out) {
    const auto t0 = std::chrono::steady_clock::now();
    int rc = pdmpc_pack_step(h, n, in, pred_offset, pred_index, fallback_shapes);
    if (rc) return rc;
    h->last_us[0] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    h->last_us[1] = h->last_us[2] = 0;
    return plan_packed_growing(h, n, out);
}

int pdmpc_plan_step_lean(pdmpc_handle* h, int32_t n, const pdmpc_vehicle_in* in, const int32_t* pred_offset, const int32_t* pred_index, const pdmpc_polygon_set* fallback_shapes,
                         int32_t* status, double* final_cost) {
    if (!h || n < 0 || (n > 0 && (!status || !final_cost))) return fail(PDMPC_ERR_INVALID, "null argument");
    const auto t0 = std::chrono::steady_clock::now();
    int rc = pdmpc_pack_step(h, n, in, pred_offset, pred_index, fallback_shapes);
    if (rc) return rc;
    h->last_us[0] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    h->last_us[1] = h->last_us[2] = 0;
    return plan_packed_growing(h, n, nullptr, status, final_cost);
}

int pdmpc_last_call_timing(pdmpc_handle* h, double* us3) {
    if (!h || !us3) return fail(PDMPC_ERR_INVALID, "null argument");
    for (int i = 0; i < 3; ++i) us3[i] = h->last_us[i];
    return PDMPC_OK;
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
    ON_DEVICE(h->cfg.device);
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
    ON_DEVICE(h->cfg.device);
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
    ON_DEVICE(h->cfg.device);
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
    ON_DEVICE(h->cfg.device);
    if (n > 0) HIPCHK(hipMemcpyAsync(dev_records, h->d_out.p + first, (size_t)n * sizeof(pdmpc_vehicle_out), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return PDMPC_OK;
}

int pdmpc_export_results_async(pdmpc_handle* h, int32_t first, int32_t n, void* dev_records) {
    if (!h || (n > 0 && !dev_records)) return fail(PDMPC_ERR_INVALID, "null argument");
    if (!h->banks[h->bank].perm.empty()) return fail(PDMPC_ERR_INVALID, "the packed batch was put into level order by the library: raw slots are not the caller's vehicles (pack it in level order to use the device-resident record path)");
    if (first < 0 || n < 0 || first + n > h->max_vehicles) return fail(PDMPC_ERR_INVALID, "slot range out of bounds");
    ON_DEVICE(h->cfg.device);
    if (n > 0) HIPCHK(hipMemcpyAsync(dev_records, h->d_out.p + first, (size_t)n * sizeof(pdmpc_vehicle_out), hipMemcpyDeviceToDevice, h->stream));
    return PDMPC_OK;
}

int pdmpc_stream(pdmpc_handle* h, void** hip_stream) {
    if (!h || !hip_stream) return fail(PDMPC_ERR_INVALID, "null argument");
    *hip_stream = (void*)h->stream;
    return PDMPC_OK;
}

int pdmpc_debug_counters(pdmpc_handle* h, uint64_t* out16) {
    if (!h || !out16) return fail(PDMPC_ERR_INVALID, "null argument");
    ON_DEVICE(h->cfg.device);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out16, h->d_work_count.p, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PDMPC_OK;
}

int pdmpc_get_last_stats(pdmpc_handle* h, pdmpc_stats* stats) {
    if (!h || !stats) return fail(PDMPC_ERR_INVALID, "null argument");
    ON_DEVICE(h->cfg.device);
    HIPCHK(hipStreamSynchronize(h->stream));
    double ms = h->folded_kernel_ms;
    for (size_t i = 0; i < h->events_used; ++i) {
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, h->events[i].first, h->events[i].second));
        ms += t;
    }
    h->stats.kernel_ms = ms;
    h->stats.n_launches = h->folded_launches + (int64_t)h->events_used;
    int32_t ctr[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(ctr, h->d_tie_count.p, sizeof ctr, hipMemcpyDeviceToHost));
    h->stats.queue_fallbacks = ctr[0];
    h->stats.speculation_arrivals = ctr[2];
    unsigned long long work[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpy(work, h->d_work_count.p, sizeof work, hipMemcpyDeviceToHost));
    h->stats.edge_checks = (int64_t)work[0];
    h->stats.segment_pair_tests = (int64_t)work[1];
    h->stats.kernel = h->last_launch_search ? 2 : 3;
    h->stats.nodes_processed = (int64_t)work[2];
    h->stats.rounds = (int64_t)work[3];
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
    ON_DEVICE(h->cfg.device);
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

namespace {
#define PDMPC_TREE_FRONTIER 0x40000000  /* d_tree_size marker: the arena holds the frontier kernel's raw tree (creation order differs from the reference's) */
#define PDMPC_TREE_REPLAYED 0x20000000  /* ... and the search ended on the replay through the binary heap (equal keys): its pop sequence, in arena indices, is in the mid list's array */
#define PDMPC_TREE_SIZE(sz) ((sz) & ~(PDMPC_TREE_FRONTIER | PDMPC_TREE_REPLAYED))

// The frontier kernel processes open nodes in parallel, so its arena holds the reference's tree plus some nodes the
// reference never creates, in another order.  This turns it back into the reference's tree and pop sequence, on the host
// and independently of the kernel's phase B (it sorts the popped nodes instead of counting them), for the debug read-backs
// the parity tests use.  Order (bulk_search.hpp): X is popped before Y iff X is an ancestor of Y or the largest key on
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
    HIPCHK(hipMemcpy(key.data(), h->akey.p + off, (size_t)raw_n * 8, hipMemcpyDeviceToHost));
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
    if (vehicle < 0 || vehicle >= h->max_vehicles) return fail(PDMPC_ERR_INVALID, "vehicle slot out of range");
    if (!h->banks[h->bank].inv.empty() && vehicle < h->banks[h->bank].n_packed) vehicle = h->banks[h->bank].inv[(size_t)vehicle];  // (the batch was put into level order)
    ON_DEVICE(h->cfg.device);
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
    return fail(PDMPC_ERR_INVALID, "no graph search has run in that slot");
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
    ON_DEVICE(h->cfg.device);
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
    ON_DEVICE(h->cfg.device);
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
    if (key) HIPCHK(hipMemcpy(key, h->akey.p + off, m * 8, hipMemcpyDeviceToHost));
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
    ON_DEVICE(h->cfg.device);
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
