// group.cpp — several GPUs behind the C ABI (include/pdmpc.h: pdmpc_group_*).
//
// What is exchanged.  In the reference every vehicle publishes its solved areas after it has planned and every coupled vehicle reads
// them before it plans (hlc/communication/PredictionsCommunication.m:34-63: send_message / read_message on the Predictions topic;
// called from PrioritizedController.publish_predictions, hlc/controller/prioritized/PrioritizedController.m:356-365, read back in
// consider_predecessors, :476-491).  Between GPUs that broadcast is an all-gather of the fixed-stride result records
// (pdmpc_vehicle_out, 2.9 KB per vehicle) on the handles' own streams: RCCL over xGMI, one communicator per device of ONE process
// (ncclCommInitAll), every collective issued for all devices inside ncclGroupStart / ncclGroupEnd by the calling thread.
//
// How a step is split (the twin of p-dmpc_amd/pdmpc/distributed.py, which drives the same protocol from one process per GPU
// through torch.distributed; tests/test_group.py compares the partitions):
//   - the weakly connected components of the step's coupling graph exchange nothing within the step (the reference's vehicles do
//     not even subscribe to uncoupled ones, PrioritizedController.m:208-255): a device takes whole components, longest processing
//     time first, plans them with ONE launch (hand-off between levels on the device) and one all-gather ends the step;
//   - a component that outweighs the mean load per device (or every component, PDMPC_SHARD_LEVELS) is planned by all devices level
//     by level: the level's slots block-partitioned, pdmpc_launch_range per device, all-gather, pdmpc_import_results of the other
//     devices' blocks, next level — everything enqueued on the streams, the host waits once per step.
// librccl is loaded with dlopen when the first group that needs it is created: the single-GPU library has no link-time dependency on it.
//
// The exchange sits behind a function table (struct Collective): RCCL's all-gather between the distinct devices of a group, or the
// same all-gather as peer copies ordered by events on the handles' streams (PDMPC_COLLECTIVE_COPY) — which also works between
// LOGICAL ranks that share a physical device (two handles, two streams, two sets of arenas on one GPU), so every line of the
// multi-rank protocol (slot remapping, block partition of a level, import of the other ranks' blocks) runs on a 1-GPU box
// (tests/test_group.py).  Every entry point leaves the caller's current device as it found it (DeviceScope).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/pdmpc.h"

extern "C" void pdmpc_set_last_error(const char* msg);  // api.cpp

namespace {

int gfail(int code, const std::string& msg) {
    pdmpc_set_last_error(msg.c_str());
    return code;
}

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
std::mutex g_rccl_mutex;
Rccl g_rccl;

bool load_rccl(std::string& err) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return true;
    // (a librccl.so.1 the process has loaded already — torch ships one — is the one dlopen returns: never two copies)
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
    }
    if (!lib) {
        err = std::string("librccl could not be loaded: ") + (dlerror() ? dlerror() : "not found");
        return false;
    }
    Rccl r;
    r.lib = lib;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(lib, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(lib, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(lib, "ncclAllGather");
    r.GroupStart = (decltype(r.GroupStart))dlsym(lib, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(lib, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd) {
        err = "librccl lacks ncclCommInitAll / ncclAllGather / ncclGroupStart";
        return false;
    }
    g_rccl = r;
    return true;
}

// the caller's current device, put back on every exit path (torch takes its current device from hipGetDevice)
struct DeviceScope {
    int prev = -1;
    DeviceScope() {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    }
    ~DeviceScope() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

// ---------------------------------------------------------------------------------------------------------------
// Partition (pure host logic; pdmpc_group_partition exposes it).

struct Partition {
    std::vector<std::vector<int>> parts;  // per device: the slots (caller's indices) of its whole components, in level order
    std::vector<int> shared;              // slots of the component planned by levels over all devices, in LEVEL ORDER
    std::vector<int> level_sizes;         // ... and its computation levels
    std::vector<int> level_of;            // [n] computation level (1-based) within the whole problem's coupling DAG
};

// union-find over the predecessor lists: component label per slot = smallest slot of the component (distributed.weak_components)
std::vector<int> weak_components(int n, const int32_t* off, const int32_t* idx) {
    std::vector<int> parent((size_t)n);
    std::iota(parent.begin(), parent.end(), 0);
    auto find = [&](int a) {
        while (parent[(size_t)a] != a) {
            parent[(size_t)a] = parent[(size_t)parent[(size_t)a]];
            a = parent[(size_t)a];
        }
        return a;
    };
    if (off)
        for (int s = 0; s < n; ++s)
            for (int q = off[s]; q < off[s + 1]; ++q) {
                const int p = idx[q];
                if (p < 0 || p >= n) continue;
                const int ra = find(s), rb = find(p);
                if (ra != rb) parent[(size_t)std::max(ra, rb)] = std::min(ra, rb);
            }
    std::vector<int> label((size_t)n);
    for (int s = 0; s < n; ++s) label[(size_t)s] = find(s);
    return label;
}

// computation levels by longest path (kahn.m: a vehicle's level = 1 + the highest level among its predecessors); -1 on a cycle
int levels_of(int n, const int32_t* off, const int32_t* idx, std::vector<int>& level) {
    level.assign((size_t)n, 0);
    std::vector<int> indeg((size_t)n, 0), succ_off((size_t)n + 1, 0), succ, queue;
    if (off)
        for (int i = 0; i < n; ++i)
            for (int q = off[i]; q < off[i + 1]; ++q)
                if (idx[q] >= 0 && idx[q] < n) {
                    if (idx[q] == i) return -1;
                    succ_off[(size_t)idx[q] + 1] += 1;
                    indeg[(size_t)i] += 1;
                }
    for (int i = 0; i < n; ++i) succ_off[(size_t)i + 1] += succ_off[(size_t)i];
    succ.resize((size_t)succ_off[(size_t)n]);
    std::vector<int> fill(succ_off.begin(), succ_off.end() - 1);
    if (off)
        for (int i = 0; i < n; ++i)
            for (int q = off[i]; q < off[i + 1]; ++q)
                if (idx[q] >= 0 && idx[q] < n) succ[(size_t)fill[(size_t)idx[q]]++] = i;
    for (int i = 0; i < n; ++i)
        if (indeg[(size_t)i] == 0) {
            level[(size_t)i] = 1;
            queue.push_back(i);
        }
    for (size_t qi = 0; qi < queue.size(); ++qi) {
        const int u = queue[qi];
        for (int q = succ_off[(size_t)u]; q < succ_off[(size_t)u + 1]; ++q) {
            const int w = succ[(size_t)q];
            level[(size_t)w] = std::max(level[(size_t)w], level[(size_t)u] + 1);
            if (--indeg[(size_t)w] == 0) queue.push_back(w);
        }
    }
    return (int)queue.size() == n ? 0 : -1;
}

int make_partition(int n, const int32_t* off, const int32_t* idx, const double* weights, int world, int mode, Partition& P) {
    P.parts.assign((size_t)world, {});
    P.shared.clear();
    P.level_sizes.clear();
    if (levels_of(n, off, idx, P.level_of)) return gfail(PDMPC_ERR_INVALID, "the sequential coupling graph has a cycle");
    const std::vector<int> label = weak_components(n, off, idx);
    std::vector<int> comp_ids;  // labels in ascending order
    std::vector<std::vector<int>> comp((size_t)n);
    for (int s = 0; s < n; ++s) {
        if (comp[(size_t)label[(size_t)s]].empty()) comp_ids.push_back(label[(size_t)s]);
        comp[(size_t)label[(size_t)s]].push_back(s);
    }
    std::sort(comp_ids.begin(), comp_ids.end());
    auto weight = [&](int c) {
        double w = 0;
        for (int s : comp[(size_t)c]) w += weights ? weights[s] : 1.0;
        return w;
    };
    int heavy = -1;
    if (mode == PDMPC_SHARD_LEVELS) {
        for (int s = 0; s < n; ++s) P.shared.push_back(s);
        comp_ids.clear();
    } else if (mode == PDMPC_SHARD_AUTO && world > 1 && !comp_ids.empty()) {
        // distributed.hybrid_partition: the heaviest component (ties: the smaller label) is shared if it outweighs the mean load per device
        double total = 0, wh = -1;
        for (int c : comp_ids) {
            const double w = weight(c);
            total += w;
            if (w > wh) {
                wh = w;
                heavy = c;
            }
        }
        if ((int)comp[(size_t)heavy].size() >= 2 * world && wh > total / world) {
            P.shared = comp[(size_t)heavy];
            comp_ids.erase(std::find(comp_ids.begin(), comp_ids.end(), heavy));
        }
    }
    // whole components: longest processing time first (ties: the smaller label), each to the least loaded device (ties: the lower rank)
    std::stable_sort(comp_ids.begin(), comp_ids.end(), [&](int a, int b) { return weight(a) > weight(b); });
    std::vector<double> load((size_t)world, 0.0);
    for (int c : comp_ids) {
        int r = 0;
        for (int q = 1; q < world; ++q)
            if (load[(size_t)q] < load[(size_t)r]) r = q;
        for (int s : comp[(size_t)c]) P.parts[(size_t)r].push_back(s);
        load[(size_t)r] += weight(c);
    }
    // a device's slots in a topological order of the coupling DAG: the sub-problem handed to pdmpc_pack_step then has its
    // predecessors in lower slots whatever order the caller's slots are in, so the handle does not permute it and the device-resident
    // record path (pdmpc_export_results_async) applies; group_fetch scatters back through these lists.  Without weights that is
    // level order (ascending caller index within a level); with weights, PRIORITY order — the largest expected work among a vehicle
    // and its descendants, descending, ties by level (api.cpp: pack_common does the same for a single handle, pdmpc_set_step_weights):
    // a device with more searches than CUs starts its heavy searches with the launch, not behind the searches that wait.
    std::vector<double> prio;
    if (weights && off) {
        prio.assign((size_t)n, 0.0);
        for (int s = 0; s < n; ++s) prio[(size_t)s] = (weights[s] == weights[s] && weights[s] > 0) ? weights[s] : 0.0;
        std::vector<int> by_level((size_t)n);
        std::iota(by_level.begin(), by_level.end(), 0);
        std::stable_sort(by_level.begin(), by_level.end(), [&](int a, int b) { return P.level_of[(size_t)a] > P.level_of[(size_t)b]; });
        for (int s : by_level)  // (a vehicle after all its successors)
            for (int q = off[s]; q < off[s + 1]; ++q)
                if (idx[q] >= 0 && idx[q] < n) prio[(size_t)idx[q]] = std::max(prio[(size_t)idx[q]], prio[(size_t)s]);
    }
    for (auto& p : P.parts) {
        std::sort(p.begin(), p.end());
        std::stable_sort(p.begin(), p.end(), [&](int a, int b) {
            if (!prio.empty() && prio[(size_t)a] != prio[(size_t)b]) return prio[(size_t)a] > prio[(size_t)b];
            return P.level_of[(size_t)a] < P.level_of[(size_t)b];
        });
    }
    // the shared component in level order (levels by longest path, ascending vehicle index within a level: find(levels == i))
    if (!P.shared.empty()) {
        std::stable_sort(P.shared.begin(), P.shared.end(), [&](int a, int b) { return P.level_of[(size_t)a] < P.level_of[(size_t)b]; });
        int cur = -1;
        for (int s : P.shared) {
            if (P.level_of[(size_t)s] != cur) {
                cur = P.level_of[(size_t)s];
                P.level_sizes.push_back(0);
            }
            P.level_sizes.back() += 1;
        }
    }
    return PDMPC_OK;
}

// the step problem restricted to `slots`: predecessor indices remapped to positions (predecessors outside the selection cannot occur:
// the selection is made of whole components)
struct SubProblem {
    std::vector<pdmpc_vehicle_in> in;
    std::vector<int32_t> pred_off, pred_idx;
    std::vector<pdmpc_polygon_set> fallback;
    bool any_fallback = false;
};
void make_sub(const std::vector<int>& slots, int n, const pdmpc_vehicle_in* in, const int32_t* off, const int32_t* idx, const pdmpc_polygon_set* fb, SubProblem& S) {
    std::vector<int> pos((size_t)n, -1);
    for (size_t i = 0; i < slots.size(); ++i) pos[(size_t)slots[i]] = (int)i;
    S.in.clear();
    S.pred_off.assign(1, 0);
    S.pred_idx.clear();
    S.fallback.clear();
    S.any_fallback = fb != nullptr;
    for (int s : slots) {
        S.in.push_back(in[s]);
        if (off)
            for (int q = off[s]; q < off[s + 1]; ++q)
                if (idx[q] >= 0 && idx[q] < n && pos[(size_t)idx[q]] >= 0) S.pred_idx.push_back(pos[(size_t)idx[q]]);
        S.pred_off.push_back((int32_t)S.pred_idx.size());
        if (fb) S.fallback.push_back(fb[s]);
    }
    if (S.pred_idx.empty()) S.pred_idx.push_back(0);
}

const size_t kRec = sizeof(pdmpc_vehicle_out);

}  // namespace

struct pdmpc_group;
namespace {
// the exchange between the ranks of a group: `per` records from every rank's send buffer into every rank's receive buffer
// (rank r's block at r * per), enqueued on the handles' streams
struct Collective {
    const char* name;
    int (*all_gather)(pdmpc_group* g, size_t per);
};
}  // namespace

struct pdmpc_group {
    std::vector<int> dev;
    std::vector<pdmpc_handle*> h;
    std::vector<hipStream_t> stream;
    std::vector<ncclComm_t> comm;
    const Collective* coll = nullptr;
    std::vector<hipEvent_t> ev_sent, ev_got;  // copy collective: rank r's block is ready / rank r has copied every block
    int last_bank = -1;                       // the bank whose records recv[] / shared_buf hold (pdmpc_group_fetch)
    std::vector<unsigned char*> send, recv;  // per device: its block of records / every device's block
    size_t send_cap = 0;                      // records per block the buffers hold
    unsigned char* shared_buf = nullptr;      // device 0: the records of the component planned by levels
    size_t shared_cap = 0;
    std::vector<unsigned char> host;          // read-back staging
    double timing[6] = {0, 0, 0, 0, 0, 0};
    struct Plan {                             // a packed step (pdmpc_group_pack_step): how it is split and how large its blocks are
        bool valid = false;
        int n = 0;
        Partition P;
        size_t per_w = 0;
    };
    std::vector<Plan> plans;
};

namespace {

#define GHIP(expr)                                                                                                              \
    do {                                                                                                                        \
        hipError_t e__ = (expr);                                                                                                \
        if (e__ != hipSuccess) {                                                                                                \
            char b__[512];                                                                                                      \
            snprintf(b__, sizeof b__, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__);              \
            return gfail(PDMPC_ERR_HIP, b__);                                                                                   \
        }                                                                                                                       \
    } while (0)
#define GNCCL(expr)                                                                                                             \
    do {                                                                                                                        \
        ncclResult_t r__ = (expr);                                                                                              \
        if (r__ != ncclSuccess) {                                                                                               \
            char b__[512];                                                                                                      \
            snprintf(b__, sizeof b__, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r__) : "?", __FILE__, __LINE__); \
            return gfail(PDMPC_ERR_HIP, b__);                                                                                   \
        }                                                                                                                       \
    } while (0)
#define GRC(expr)                \
    do {                         \
        const int rc__ = (expr); \
        if (rc__) return rc__;   \
    } while (0)

// HBM banks of the two sub-problems of group bank b (pdmpc_select_bank): the handles' own banks 0 .. 2047 stay the caller's
const int kGroupBanks = 1000;
inline int bank_shared(int b) { return 2048 + 2 * b; }
inline int bank_whole(int b) { return 2049 + 2 * b; }

int ensure_buffers(pdmpc_group* g, size_t per, size_t n_shared) {
    const size_t world = g->h.size();
    if (per > g->send_cap) {
        const size_t cap = per + per / 2 + 16;
        for (size_t r = 0; r < world; ++r) {
            GHIP(hipSetDevice(g->dev[r]));
            GHIP(hipStreamSynchronize(g->stream[r]));
            if (g->send[r]) (void)hipFree(g->send[r]);
            if (g->recv[r]) (void)hipFree(g->recv[r]);
            g->send[r] = g->recv[r] = nullptr;
            GHIP(hipMalloc((void**)&g->send[r], cap * kRec));
            GHIP(hipMalloc((void**)&g->recv[r], cap * kRec * world));
            GHIP(hipMemset(g->send[r], 0, cap * kRec));
            GHIP(hipMemset(g->recv[r], 0, cap * kRec * world));
        }
        g->send_cap = cap;
    }
    if (n_shared > g->shared_cap) {
        GHIP(hipSetDevice(g->dev[0]));
        GHIP(hipStreamSynchronize(g->stream[0]));
        if (g->shared_buf) (void)hipFree(g->shared_buf);
        g->shared_buf = nullptr;
        const size_t cap = n_shared + n_shared / 2 + 16;
        GHIP(hipMalloc((void**)&g->shared_buf, cap * kRec));
        g->shared_cap = cap;
    }
    return PDMPC_OK;
}

// one all-gather of `per` records per device over the whole group, on the handles' streams: RCCL ...
int all_gather_rccl(pdmpc_group* g, size_t per) {
    const size_t world = g->h.size();
    GNCCL(g_rccl.GroupStart());
    for (size_t r = 0; r < world; ++r) GNCCL(g_rccl.AllGather(g->send[r], g->recv[r], per * kRec, ncclChar, g->comm[r], g->stream[r]));
    GNCCL(g_rccl.GroupEnd());
    return PDMPC_OK;
}
// ... or peer copies.  Rank q's stream waits for every rank's block (an event per rank, recorded behind its export), copies the
// blocks into its own receive buffer, and says so; a rank's stream goes on (its next export overwrites its send buffer) once
// everybody has copied.  Nothing waits on the host.
int all_gather_copy(pdmpc_group* g, size_t per) {
    const size_t world = g->h.size();
    for (size_t r = 0; r < world; ++r) {
        GHIP(hipSetDevice(g->dev[r]));
        GHIP(hipEventRecord(g->ev_sent[r], g->stream[r]));
    }
    for (size_t q = 0; q < world; ++q) {
        GHIP(hipSetDevice(g->dev[q]));
        for (size_t r = 0; r < world; ++r) {
            if (r != q) GHIP(hipStreamWaitEvent(g->stream[q], g->ev_sent[r], 0));
            unsigned char* dst = g->recv[q] + r * per * kRec;
            if (g->dev[q] == g->dev[r])
                GHIP(hipMemcpyAsync(dst, g->send[r], per * kRec, hipMemcpyDeviceToDevice, g->stream[q]));
            else
                GHIP(hipMemcpyPeerAsync(dst, g->dev[q], g->send[r], g->dev[r], per * kRec, g->stream[q]));
        }
        GHIP(hipEventRecord(g->ev_got[q], g->stream[q]));
    }
    for (size_t r = 0; r < world; ++r) {
        GHIP(hipSetDevice(g->dev[r]));
        for (size_t q = 0; q < world; ++q)
            if (q != r) GHIP(hipStreamWaitEvent(g->stream[r], g->ev_got[q], 0));
    }
    return PDMPC_OK;
}
const Collective kCollRccl = {"rccl", all_gather_rccl};
const Collective kCollCopy = {"copy", all_gather_copy};
inline int all_gather(pdmpc_group* g, size_t per) { return g->coll->all_gather(g, per); }

using clk = std::chrono::steady_clock;
inline double ms_between(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

// partition the step, build the sub-problems and make them resident on the devices (group bank `bank`)
int group_pack(pdmpc_group* g, int bank, int n, const pdmpc_vehicle_in* in, const int32_t* off, const int32_t* idx, const pdmpc_polygon_set* fb, const double* weights, int mode) {
    const auto t0 = clk::now();
    const int world = (int)g->h.size();
    if ((size_t)bank >= g->plans.size()) g->plans.resize((size_t)bank + 1);
    pdmpc_group::Plan& L = g->plans[(size_t)bank];
    L.valid = false;
    if (g->last_bank == bank) g->last_bank = -1;
    L.n = n;
    Partition& P = L.P;
    GRC(make_partition(n, off, idx, weights, world, mode, P));
    SubProblem S;
    std::vector<SubProblem> W((size_t)world);
    if (!P.shared.empty()) make_sub(P.shared, n, in, off, idx, fb, S);
    size_t per_w = 0, per_l = 0;
    for (int r = 0; r < world; ++r) {
        if (!P.parts[(size_t)r].empty()) make_sub(P.parts[(size_t)r], n, in, off, idx, fb, W[(size_t)r]);
        per_w = std::max(per_w, P.parts[(size_t)r].size());
    }
    for (int sz : P.level_sizes) per_l = std::max(per_l, (size_t)((sz + world - 1) / world));
    L.per_w = std::max<size_t>(per_w, 1);
    const auto t1 = clk::now();
    // the shared component on every device, a device's whole components on that device
    const int nS = (int)P.shared.size();
    for (int r = 0; r < world; ++r) {
        if (nS) {
            GRC(pdmpc_select_bank(g->h[(size_t)r], bank_shared(bank)));
            GRC(pdmpc_pack_step(g->h[(size_t)r], nS, S.in.data(), S.pred_off.data(), S.pred_idx.data(), S.any_fallback ? S.fallback.data() : nullptr));
        }
        const SubProblem& w = W[(size_t)r];
        if (!w.in.empty()) {
            GRC(pdmpc_select_bank(g->h[(size_t)r], bank_whole(bank)));
            GRC(pdmpc_pack_step(g->h[(size_t)r], (int)w.in.size(), w.in.data(), w.pred_off.data(), w.pred_idx.data(), w.any_fallback ? w.fallback.data() : nullptr));
        }
        GRC(pdmpc_select_bank(g->h[(size_t)r], 0));
    }
    GRC(ensure_buffers(g, std::max<size_t>(std::max(per_w, per_l), 1), (size_t)nS));
    L.valid = true;
    g->timing[1] = ms_between(t0, t1);
    g->timing[2] = ms_between(t1, clk::now());
    return PDMPC_OK;
}

// plan the packed step: everything enqueued on the handles' streams, one wait at the end
int group_launch(pdmpc_group* g, int bank) {
    if ((size_t)bank >= g->plans.size() || !g->plans[(size_t)bank].valid) return gfail(PDMPC_ERR_INVALID, "pdmpc_group_launch: nothing packed in that bank");
    const auto t2 = clk::now();
    const int world = (int)g->h.size();
    const pdmpc_group::Plan& L = g->plans[(size_t)bank];
    const Partition& P = L.P;
    const int nS = (int)P.shared.size();
    // ---- the shared component, level by level (PredictionsCommunication.m:34-63 per level)
    if (nS) {
        for (int r = 0; r < world; ++r) {
            GRC(pdmpc_select_bank(g->h[(size_t)r], bank_shared(bank)));
            GRC(pdmpc_begin_step(g->h[(size_t)r]));
        }
        int first = 0;
        for (int size : P.level_sizes) {
            const int per = (size + world - 1) / world;
            auto block = [&](int r, int& lo, int& hi) {
                lo = std::min(first + r * per, first + size);
                hi = std::min(lo + per, first + size);
            };
            for (int r = 0; r < world; ++r) {
                int lo, hi;
                block(r, lo, hi);
                if (hi > lo) {
                    GRC(pdmpc_launch_range(g->h[(size_t)r], lo, hi - lo));
                    GRC(pdmpc_export_results_async(g->h[(size_t)r], lo, hi - lo, g->send[(size_t)r]));
                }
            }
            GRC(all_gather(g, (size_t)per));
            for (int r = 0; r < world; ++r)
                for (int q = 0; q < world; ++q) {
                    int lo, hi;
                    block(q, lo, hi);
                    if (q != r && hi > lo) GRC(pdmpc_import_results(g->h[(size_t)r], lo, hi - lo, g->recv[(size_t)r] + (size_t)q * per * kRec));
                }
            first += size;
        }
        GRC(pdmpc_export_results_async(g->h[0], 0, nS, g->shared_buf));  // (before the next launch overwrites the slots)
    }
    // ---- whole components: one launch per device, one all-gather
    bool any_whole = false;
    for (int r = 0; r < world; ++r) {
        const int nr = (int)P.parts[(size_t)r].size();
        if (!nr) continue;
        any_whole = true;
        GRC(pdmpc_select_bank(g->h[(size_t)r], bank_whole(bank)));
        GRC(pdmpc_launch_packed(g->h[(size_t)r]));
        GRC(pdmpc_export_results_async(g->h[(size_t)r], 0, nr, g->send[(size_t)r]));
    }
    if (any_whole) GRC(all_gather(g, L.per_w));
    const auto t3 = clk::now();
    for (int r = 0; r < world; ++r) GRC(pdmpc_synchronize(g->h[(size_t)r]));
    for (int r = 0; r < world; ++r) GRC(pdmpc_select_bank(g->h[(size_t)r], 0));
    g->last_bank = bank;
    g->timing[3] = ms_between(t2, t3);
    g->timing[4] = ms_between(t3, clk::now());
    return PDMPC_OK;
}

// the records of the step planned last (every device holds every block: read from device 0), in the caller's order
int group_fetch(pdmpc_group* g, int bank, int n, pdmpc_vehicle_out* out) {
    if ((size_t)bank >= g->plans.size() || !g->plans[(size_t)bank].valid) return gfail(PDMPC_ERR_INVALID, "pdmpc_group_fetch: nothing packed in that bank");
    const auto t4 = clk::now();
    const pdmpc_group::Plan& L = g->plans[(size_t)bank];
    if (n != L.n) return gfail(PDMPC_ERR_INVALID, "pdmpc_group_fetch: the bank holds another number of vehicles");
    // the gathered records are those of the bank launched last, whatever bank is asked for
    if (bank != g->last_bank) return gfail(PDMPC_ERR_INVALID, "pdmpc_group_fetch: the records on the devices are those of the bank launched last; launch this bank first");
    const Partition& P = L.P;
    const int world = (int)g->h.size(), nS = (int)P.shared.size();
    bool any_whole = false;
    for (int r = 0; r < world; ++r) any_whole = any_whole || !P.parts[(size_t)r].empty();
    GHIP(hipSetDevice(g->dev[0]));
    if (any_whole) {
        g->host.resize(L.per_w * kRec * (size_t)world);
        GHIP(hipMemcpy(g->host.data(), g->recv[0], g->host.size(), hipMemcpyDeviceToHost));
        for (int r = 0; r < world; ++r)
            for (size_t i = 0; i < P.parts[(size_t)r].size(); ++i) std::memcpy(&out[P.parts[(size_t)r][i]], g->host.data() + ((size_t)r * L.per_w + i) * kRec, kRec);
    }
    if (nS) {
        g->host.resize((size_t)nS * kRec);
        GHIP(hipMemcpy(g->host.data(), g->shared_buf, g->host.size(), hipMemcpyDeviceToHost));
        for (int i = 0; i < nS; ++i) std::memcpy(&out[P.shared[(size_t)i]], g->host.data() + (size_t)i * kRec, kRec);
    }
    g->timing[5] = ms_between(t4, clk::now());
    return PDMPC_OK;
}

}  // namespace

extern "C" {

int pdmpc_group_partition(int32_t n, const int32_t* pred_offset, const int32_t* pred_index, const double* weights, int32_t n_devices, int32_t mode, int32_t* rank_of,
                          int32_t* level_of, int32_t* block_rank) {
    if (n < 0 || n_devices < 1 || !rank_of) return gfail(PDMPC_ERR_INVALID, "pdmpc_group_partition: bad argument");
    if (mode != PDMPC_SHARD_AUTO && mode != PDMPC_SHARD_COMPONENTS && mode != PDMPC_SHARD_LEVELS) return gfail(PDMPC_ERR_INVALID, "unknown sharding mode");
    Partition P;
    GRC(make_partition(n, pred_offset, pred_index, weights, n_devices, mode, P));
    for (int v = 0; v < n; ++v) {
        rank_of[v] = -1;
        if (level_of) level_of[v] = 0;
        if (block_rank) block_rank[v] = -1;
    }
    for (int r = 0; r < n_devices; ++r)
        for (int s : P.parts[(size_t)r]) rank_of[s] = r;
    int first = 0;
    for (size_t l = 0; l < P.level_sizes.size(); ++l) {
        const int size = P.level_sizes[l], per = (size + n_devices - 1) / n_devices;
        for (int i = 0; i < size; ++i) {
            const int s = P.shared[(size_t)(first + i)];
            if (level_of) level_of[s] = (int32_t)l + 1;
            if (block_rank) block_rank[s] = i / per;
        }
        first += size;
    }
    return PDMPC_OK;
}

int pdmpc_group_create_ex(const pdmpc_config* config, int32_t n_devices, const int32_t* devices, int32_t collective, pdmpc_group** out_group) {
    if (!config || !out_group || n_devices < 1 || n_devices > 64) return gfail(PDMPC_ERR_INVALID, "pdmpc_group_create: bad argument");
    if (collective != PDMPC_COLLECTIVE_AUTO && collective != PDMPC_COLLECTIVE_RCCL && collective != PDMPC_COLLECTIVE_COPY)
        return gfail(PDMPC_ERR_INVALID, "pdmpc_group_create_ex: unknown collective");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return gfail(PDMPC_ERR_NO_DEVICE, "no HIP device visible: this backend has no CPU fallback");
    DeviceScope keep;
    pdmpc_group* g = new pdmpc_group();
    bool repeated = false;
    for (int r = 0; r < n_devices; ++r) {
        const int d = devices ? devices[r] : r;
        if (d < 0 || d >= ndev) {
            pdmpc_group_destroy(g);
            return gfail(PDMPC_ERR_NO_DEVICE, "pdmpc_group_create: device ordinal out of range");
        }
        repeated = repeated || std::find(g->dev.begin(), g->dev.end(), d) != g->dev.end();
        g->dev.push_back(d);
    }
    // ranks that share a device are logical ranks: RCCL has one rank per device, the copy collective does not care
    if (collective == PDMPC_COLLECTIVE_AUTO) {
        const char* e = getenv("PDMPC_GROUP_COLLECTIVE");
        collective = repeated ? PDMPC_COLLECTIVE_COPY : ((e && std::strcmp(e, "copy") == 0) ? PDMPC_COLLECTIVE_COPY : PDMPC_COLLECTIVE_RCCL);
    }
    if (repeated && collective == PDMPC_COLLECTIVE_RCCL) {
        pdmpc_group_destroy(g);
        return gfail(PDMPC_ERR_NO_DEVICE, "pdmpc_group_create: a device listed twice needs PDMPC_COLLECTIVE_COPY (RCCL has one rank per device)");
    }
    g->coll = collective == PDMPC_COLLECTIVE_COPY ? &kCollCopy : &kCollRccl;
    std::string err;
    if (g->coll == &kCollRccl && !load_rccl(err)) {
        pdmpc_group_destroy(g);
        return gfail(PDMPC_ERR_NO_DEVICE, err);
    }
    g->send.assign((size_t)n_devices, nullptr);
    g->recv.assign((size_t)n_devices, nullptr);
    for (int r = 0; r < n_devices; ++r) {
        pdmpc_config c = *config;
        c.device = g->dev[(size_t)r];
        pdmpc_handle* h = nullptr;
        const int rc = pdmpc_create(&c, &h);
        if (rc) {
            pdmpc_group_destroy(g);
            return rc;
        }
        g->h.push_back(h);
        const int share = (int)std::count(g->dev.begin(), g->dev.end(), g->dev[(size_t)r]);
        if (share > 1) (void)pdmpc_set_device_share(h, share);  // (logical ranks: their launches run side by side on one GPU)
        void* st = nullptr;
        (void)pdmpc_stream(h, &st);
        g->stream.push_back((hipStream_t)st);
    }
    if (g->coll == &kCollRccl) {
        g->comm.assign((size_t)n_devices, nullptr);
        const ncclResult_t nr = g_rccl.CommInitAll(g->comm.data(), n_devices, g->dev.data());
        if (nr != ncclSuccess) {
            g->comm.clear();
            pdmpc_group_destroy(g);
            return gfail(PDMPC_ERR_HIP, std::string("ncclCommInitAll failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(nr) : "?"));
        }
    } else {
        for (int r = 0; r < n_devices; ++r) {
            hipEvent_t a = nullptr, b = nullptr;
            if (hipSetDevice(g->dev[(size_t)r]) != hipSuccess || hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) {
                if (a) (void)hipEventDestroy(a);
                pdmpc_group_destroy(g);
                return gfail(PDMPC_ERR_HIP, "pdmpc_group_create: hipEventCreate failed");
            }
            g->ev_sent.push_back(a);
            g->ev_got.push_back(b);
            // peer access between the distinct devices of the group (a copy between devices without it goes through the host)
            for (int q = 0; q < n_devices; ++q)
                if (g->dev[(size_t)q] != g->dev[(size_t)r]) {
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, g->dev[(size_t)r], g->dev[(size_t)q]) == hipSuccess && can) {
                        const hipError_t pe = hipDeviceEnablePeerAccess(g->dev[(size_t)q], 0);
                        if (pe != hipSuccess) (void)hipGetLastError();  // (already enabled: fine)
                    }
                }
        }
    }
    *out_group = g;
    return PDMPC_OK;
}

int pdmpc_group_create(const pdmpc_config* config, int32_t n_devices, const int32_t* devices, pdmpc_group** out_group) {
    return pdmpc_group_create_ex(config, n_devices, devices, PDMPC_COLLECTIVE_AUTO, out_group);
}

int pdmpc_group_collective(pdmpc_group* g, int32_t* collective) {
    if (!g || !collective) return gfail(PDMPC_ERR_INVALID, "null argument");
    *collective = g->coll == &kCollCopy ? PDMPC_COLLECTIVE_COPY : PDMPC_COLLECTIVE_RCCL;
    return PDMPC_OK;
}

int pdmpc_group_destroy(pdmpc_group* g) {
    if (!g) return PDMPC_OK;
    DeviceScope keep;
    for (size_t r = 0; r < g->h.size(); ++r) (void)pdmpc_synchronize(g->h[r]);
    for (hipEvent_t e : g->ev_sent) (void)hipEventDestroy(e);
    for (hipEvent_t e : g->ev_got) (void)hipEventDestroy(e);
    for (ncclComm_t c : g->comm)
        if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
    for (size_t r = 0; r < g->send.size(); ++r) {
        if (r < g->dev.size()) (void)hipSetDevice(g->dev[r]);
        if (g->send[r]) (void)hipFree(g->send[r]);
        if (g->recv[r]) (void)hipFree(g->recv[r]);
    }
    if (g->shared_buf) {
        (void)hipSetDevice(g->dev[0]);
        (void)hipFree(g->shared_buf);
    }
    for (pdmpc_handle* h : g->h) (void)pdmpc_destroy(h);
    delete g;
    return PDMPC_OK;
}

int pdmpc_group_size(pdmpc_group* g, int32_t* n_devices) {
    if (!g || !n_devices) return gfail(PDMPC_ERR_INVALID, "null argument");
    *n_devices = (int32_t)g->h.size();
    return PDMPC_OK;
}

int pdmpc_group_handle(pdmpc_group* g, int32_t rank, pdmpc_handle** handle) {
    if (!g || !handle || rank < 0 || rank >= (int32_t)g->h.size()) return gfail(PDMPC_ERR_INVALID, "pdmpc_group_handle: bad argument");
    *handle = g->h[(size_t)rank];
    return PDMPC_OK;
}

int pdmpc_group_upload_mpa(pdmpc_group* g, const pdmpc_mpa* mpa) {
    if (!g || !mpa) return gfail(PDMPC_ERR_INVALID, "null argument");
    DeviceScope keep;
    for (pdmpc_handle* h : g->h) GRC(pdmpc_upload_mpa(h, mpa));
    return PDMPC_OK;
}

int pdmpc_group_plan_step(pdmpc_group* g, int32_t n, const pdmpc_vehicle_in* in, const int32_t* pred_offset, const int32_t* pred_index, const pdmpc_polygon_set* fallback_shapes,
                          const double* weights, int32_t mode, pdmpc_vehicle_out* out) {
    if (!g || n < 0 || (n > 0 && (!in || !out))) return gfail(PDMPC_ERR_INVALID, "pdmpc_group_plan_step: null argument");
    if (pred_offset && !pred_index) return gfail(PDMPC_ERR_INVALID, "pred_index missing");
    if (mode != PDMPC_SHARD_AUTO && mode != PDMPC_SHARD_COMPONENTS && mode != PDMPC_SHARD_LEVELS) return gfail(PDMPC_ERR_INVALID, "unknown sharding mode");
    if (n == 0) return PDMPC_OK;
    DeviceScope keep;
    // The reference's tree is unbounded (Tree.m:54-70): a step in which some search outgrew its arena is planned again with arenas twice
    // as large on every device (as pdmpc_plan_step does for one).
    for (;;) {
        const auto t0 = clk::now();
        GRC(group_pack(g, 0, n, in, pred_offset, pred_index, fallback_shapes, weights, mode));
        GRC(group_launch(g, 0));
        GRC(group_fetch(g, 0, n, out));
        g->timing[0] = ms_between(t0, clk::now());
        bool overflow = false;
        for (int i = 0; i < n; ++i) overflow = overflow || out[i].status == PDMPC_ARENA_OVERFLOW;
        if (!overflow) return PDMPC_OK;
        // (the devices' arenas may differ after a failed attempt to grow: the step fits when the SMALLEST arena holds every search)
        int32_t nodes = 0;
        for (pdmpc_handle* h : g->h) {
            int32_t nr = 0;
            GRC(pdmpc_arena_nodes(h, &nr, nullptr));
            nodes = nodes ? std::min(nodes, nr) : nr;
        }
        if ((int64_t)nodes * 2 > (1ll << 30)) return PDMPC_OK;  // (statuses tell)
        for (size_t r = 0; r < g->h.size(); ++r) {
            int32_t nr = 0;
            GRC(pdmpc_arena_nodes(g->h[r], &nr, nullptr));
            if (nr >= nodes * 2) continue;
            if (pdmpc_grow_arena(g->h[r], nodes * 2) != PDMPC_OK) {
                // no room to grow on this device: the records keep their PDMPC_ARENA_OVERFLOW statuses AND the error string says why
                char b[160];
                snprintf(b, sizeof b, "pdmpc_group_plan_step: the arenas of rank %zu (device %d) cannot grow to %d nodes per vehicle", r, g->dev[r], nodes * 2);
                pdmpc_set_last_error(b);
                return PDMPC_OK;
            }
        }
    }
}

int pdmpc_group_pack_step(pdmpc_group* g, int32_t bank, int32_t n, const pdmpc_vehicle_in* in, const int32_t* pred_offset, const int32_t* pred_index,
                          const pdmpc_polygon_set* fallback_shapes, const double* weights, int32_t mode) {
    if (!g || n < 0 || (n > 0 && !in)) return gfail(PDMPC_ERR_INVALID, "pdmpc_group_pack_step: null argument");
    if (bank < 0 || bank >= kGroupBanks) return gfail(PDMPC_ERR_INVALID, "pdmpc_group_pack_step: bank out of range");
    if (pred_offset && !pred_index) return gfail(PDMPC_ERR_INVALID, "pred_index missing");
    if (mode != PDMPC_SHARD_AUTO && mode != PDMPC_SHARD_COMPONENTS && mode != PDMPC_SHARD_LEVELS) return gfail(PDMPC_ERR_INVALID, "unknown sharding mode");
    DeviceScope keep;
    return group_pack(g, bank, n, in, pred_offset, pred_index, fallback_shapes, weights, mode);
}

int pdmpc_group_launch(pdmpc_group* g, int32_t bank) {
    if (!g || bank < 0) return gfail(PDMPC_ERR_INVALID, "pdmpc_group_launch: bad argument");
    DeviceScope keep;
    return group_launch(g, bank);
}

int pdmpc_group_fetch(pdmpc_group* g, int32_t bank, int32_t n, pdmpc_vehicle_out* out) {
    if (!g || bank < 0 || (n > 0 && !out)) return gfail(PDMPC_ERR_INVALID, "pdmpc_group_fetch: bad argument");
    DeviceScope keep;
    return group_fetch(g, bank, n, out);
}

int pdmpc_group_grow_arena(pdmpc_group* g, int32_t max_nodes) {
    if (!g || max_nodes <= 0) return gfail(PDMPC_ERR_INVALID, "pdmpc_group_grow_arena: bad argument");
    DeviceScope keep;
    for (pdmpc_handle* h : g->h) {
        int32_t nodes = 0;
        GRC(pdmpc_arena_nodes(h, &nodes, nullptr));
        if (nodes < max_nodes) GRC(pdmpc_grow_arena(h, max_nodes));
    }
    return PDMPC_OK;
}

int pdmpc_group_last_timing(pdmpc_group* g, double* ms6) {
    if (!g || !ms6) return gfail(PDMPC_ERR_INVALID, "null argument");
    for (int i = 0; i < 6; ++i) ms6[i] = g->timing[i];
    return PDMPC_OK;
}

}  // extern "C"
