// serial_search.hpp — the pop-ordered searches (device code): the block-min queue pipeline of round 1 (queue / expander /
// scout / validator waves) and the libstdc++-faithful binary-heap search that is exact for any keys.  Included by
// search_kernel.hip (the round-1 kernel, selectable with PDMPC_KERNEL=serial) and by frontier_kernel.hip, which falls back
// to the binary-heap search when two keys that decide the pop order are equal.
#pragma once
#include "../../include/pdmpc_math.h"
#include "pdmpc_device.h"

#ifdef PDMPC_PROFILE
#define PROF_N 16
#define PROF_MEMBERS unsigned long long prof_t0, prof_acc[PROF_N];
#define PROF_DECL \
    S.prof_t0 = 0; \
    for (int i__ = 0; i__ < PROF_N; ++i__) S.prof_acc[i__] = 0;
#define PROF_START S.prof_t0 = __builtin_readcyclecounter();
#define PROF_STOP(i)                                            \
    {                                                           \
        unsigned long long t1__ = __builtin_readcyclecounter(); \
        S.prof_acc[i] += t1__ - S.prof_t0;                      \
        S.prof_t0 = t1__;                                       \
    }
#define PROF_COUNT(i, v) S.prof_acc[i] += (v);
// hand-over timeline (profile build, trace buffer hijacked): slot k of hand-over j <- clock
#define PROF_TL(j, k)                                                                                              \
    if (A.trace_cap > 0 && (int)(12 * (j) + 12) <= A.trace_cap && lane == 0)                                         \
        A.pop_trace[(size_t)slot * A.trace_cap + 12 * (j) + (k)] = (int32_t)(__builtin_readcyclecounter() & 0x7FFFFFFFull);
#else
#define PROF_TL(j, k)
#define PROF_MEMBERS
#define PROF_DECL
#define PROF_START
#define PROF_STOP(i)
#define PROF_COUNT(i, v)
#endif

namespace {

#include "wave_primitives.hpp"
#include "search_state.hpp"
#include "heap_queue.hpp"
#include "blockmin_queue.hpp"
#include "edge_checks.hpp"


#define VS_UNKNOWN 0u
#define VS_VALID 1u
#define VS_INVALID 2u
#define VS_CLAIMED 4u   // block-min mode: a validator wave is evaluating the edge right now
#define VS_VALID_CS 3u  // valid, and a helper has already stored cos/sin(yaw) in the node's record (expand_node.m:50-51)
#define VS_DROPPED 5u   // block-min mode: invalid, and already taken out of the open list (counted as popped at the end)


// validity cache: 0 unknown, 1 valid, 2 invalid; the first NV nodes in LDS, the rest in HBM (same CU -> same L1)
struct VState {
    volatile lds_u8* l;
    uint8_t* g;
    uint32_t NV;
};
__device__ __forceinline__ uint32_t vs_load(const VState& v, uint32_t i0) {
    if (i0 < v.NV) return (uint32_t)v.l[i0];
    return (uint32_t)__hip_atomic_load(v.g + i0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void vs_store(const VState& v, uint32_t i0, uint32_t val) {
    if (i0 < v.NV)
        v.l[i0] = (uint8_t)val;
    else
        __hip_atomic_store(v.g + i0, (uint8_t)val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// unknown -> claimed, by exactly one of the waves that try (one lane calls): the winner evaluates the edge
__device__ __forceinline__ bool vs_claim(const VState& v, uint32_t i0) {
    const uint32_t sh = (i0 & 3u) * 8u;  // (both arrays are at least 4-byte aligned)
    if (i0 < v.NV) {
        lds_u32* w = (lds_u32*)(v.l + (i0 & ~3u));
        for (;;) {
            const uint32_t old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((old >> sh) & 0xFFu) return false;
            uint32_t expect = old;
            if (__hip_atomic_compare_exchange_strong(w, &expect, old | (VS_CLAIMED << sh), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return true;
        }
    }
    uint32_t* w = (uint32_t*)(v.g + (i0 & ~3u));
    for (;;) {
        const uint32_t old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((old >> sh) & 0xFFu) return false;
        uint32_t expect = old;
        if (__hip_atomic_compare_exchange_strong(w, &expect, old | (VS_CLAIMED << sh), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return true;
    }
}

}  // namespace

// LDS words shared between the waves of a workgroup (in the `path` region, after the offset tables)
#define SH_STATE 0     // 0 searching, 1 predecessor areas arrived (all waves meet in arrival_sync), 2 finished
#define SH_HEAP_LEN 1  // current open-list length (for the helpers' scan)
#define SH_VERSION 2   // bumped by the sequencing wave whenever the open list changed (idle helpers sleep on it)
#define SH_NNODES 3    // tree size, published by the sequencing wave before an arrival is handled
#define SH_RESTART 4   // set during verification: a node that was already expanded collides with the new areas
#define SH_PEND_LO 5   // predecessors whose areas are not in the soup yet (bit p = p-th predecessor)
#define SH_PEND_HI 6
#define SH_ARR_LO 7    // predecessors that just finished (to be incorporated by arrival_sync)
#define SH_ARR_HI 8
#define SH_CAND_VER 9  // block-min mode: bumped by the scout wave when it rewrote the candidate list
#define SH_CAND 10     // block-min mode: BM_NCAND node ids proposed for pre-validation, most urgent first (0 = none)
#define BM_NCAND 6
// Mail boxes of the block-min mode: 8-byte words {sequence number, payload} written and read with one LDS access each.
#define SH_Q2E_SEQ 16   // queue wave -> expander wave: number of the hand-over ...
#define SH_Q2E_ID 17    // ... and the popped node (1-based id) to evaluate and expand
#define SH_E2Q_SEQ 18   // expander wave -> queue wave: number of the hand-over this reply answers ...
#define SH_E2Q_FLAGS 19 // ... E2Q_* bits | children created << 8 (their keys are in the key ring, their records in the tree)
#define SH_Q_SYNC 21    // set by the queue wave while it is inside arrival_sync (the expander must not enter before: it may owe a reply)
#define SH_HINT_SEQ 22  // queue wave -> expander wave (mail box): the hand-over number SH_HINT_ID will probably be posted under ...
#define SH_HINT_ID 23   // ... and the node: the expander may evaluate and expand it ahead of time
#define SH_EAGER 24     // block-min mode: next node (0-based) the validator waves evaluate when nothing is urgent
#define SH_WORDS PDMPC_SH_WORDS
#define E2Q_VALID 1u     // the edge into the node is collision-free
#define E2Q_GOAL 2u      // ... and the node is at the horizon: the search is over
#define E2Q_OVERFLOW 4u  // the arena cannot take the node's children
#define ST_RUN 0u
#define ST_ARRIVED 1u
#define ST_DONE 2u
#define ST_TIE 3u  // block-min mode: the minimal key was not unique; every wave leaves and the search is redone on the binary heap

namespace {

__device__ __forceinline__ unsigned long long sh_load64(volatile lds_u32* sh, int lo) {
    return (unsigned long long)sh[lo] | ((unsigned long long)sh[lo + 1] << 32);
}

// mail box = two consecutive shared words at an even index (8-byte aligned): sequence number in the low half
__device__ __forceinline__ void mbox_post(volatile lds_u32* sh, int word, uint32_t seq, uint32_t payload) {
    *(volatile LDS_AS unsigned long long*)(sh + word) = ((unsigned long long)payload << 32) | seq;
}
__device__ __forceinline__ unsigned long long mbox_read(volatile lds_u32* sh, int word) { return *(volatile LDS_AS unsigned long long*)(sh + word); }

struct SpecCtx {
    volatile lds_u32* sh;
    lds_d2* l_soup;
    const lds_i32* l_soff;
    const lds_i32* l_lit;  // literal soup length per step
    const pdmpc_vehicle_out* out;
    const int32_t* pred;   // this vehicle's predecessor slots
    int32_t* counters;     // [0] tie fallbacks, [1] speculation restarts, [2] arrivals handled, [3] pops thrown away by restarts (cumulative, all vehicles)
    int n_pred, Hp;
};

// copy the solved areas of the predecessors in `arr` into their soup slots (PrioritizedController.m:476-491)
__device__ void incorporate_areas(const SpecCtx& P, unsigned long long arr, int tid) {
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    while (arr) {
        const int p = (int)__builtin_ctzll(arr);
        arr &= arr - 1;
        const pdmpc_vehicle_out* PO = P.out + P.pred[p];
        for (int idx = tid; idx < P.Hp * PDMPC_VMAX; idx += (int)blockDim.x) {
            const int k = idx / PDMPC_VMAX;
            const int v = idx - k * PDMPC_VMAX;
            // (all three loads at once: the coordinates do not wait for the column count — every v < VMAX is inside the array)
            const int cols = PO->shape_cols[k];
            const double sx = PO->shapes[k][0][v], sy = PO->shapes[k][1][v];
            d2 pt;
            pt.x = v < cols ? sx : qnan;
            pt.y = v < cols ? sy : qnan;
            P.l_soup[P.l_soff[k] + P.l_lit[k] + p * PDMPC_VMAX + v] = pt;
        }
    }
}

// Does the edge into node i0 (0-based, per lane) cross the areas of the predecessors in `arr`?  Same arithmetic as
// edge_valid / interx_check restricted to those polygons (InterX.m:63-76), one node per lane.
__device__ bool node_hits_areas(const Search& S, const CheckCtx& C, const SpecCtx& P, uint32_t i0, unsigned long long arr, bool& popped) {
    const NodeRec cn = node_load(S, i0);
    popped = (cn.packed & NODE_POPPED_BIT) != 0;
    if (!cn.parent) return false;
    const NodeRec pn = node_load(S, cn.parent - 1);
    const int m = NODE_MAN(cn.packed), ncols = NODE_COLS(cn.packed), k = NODE_K(cn.packed);
    const double c = pn.cs, s = pn.sn, pX = pn.x, pY = pn.y;
    const size_t abase = (size_t)m * 3 * PDMPC_VMAX;
    auto area_at = [&](int i) -> d2 { return C.areas_in_lds ? (d2)C.l_area[abase + i] : C.g_area[abase + i]; };
    bool hit = false;
    d2 a0 = area_at(0);
    d2 p0;
    p0.x = c * a0.x - s * a0.y + pX;
    p0.y = s * a0.x + c * a0.y + pY;
    for (int i = 0; i + 1 < ncols; ++i) {
        const d2 a1 = area_at(i + 1);
        d2 p1;
        p1.x = c * a1.x - s * a1.y + pX;
        p1.y = s * a1.x + c * a1.y + pY;
        const double dx1 = p1.x - p0.x, dy1 = p1.y - p0.y;
        const double S1 = dx1 * p0.y - dy1 * p0.x;
        unsigned long long rem = arr;
        while (rem) {
            const int p = (int)__builtin_ctzll(rem);
            rem &= rem - 1;
            const lds_d2* poly = P.l_soup + P.l_soff[k - 1] + P.l_lit[k - 1] + p * PDMPC_VMAX;
            d2 q0 = poly[0];
            for (int j = 0; j + 1 < PDMPC_VMAX; ++j) {
                const d2 q1 = poly[j + 1];
                const double dx2 = q1.x - q0.x, dy2 = q1.y - q0.y;
                const double S2 = dx2 * q0.y - dy2 * q0.x;
                const double A0 = dx1 * q0.y - dy1 * q0.x;
                const double A1 = dx1 * q1.y - dy1 * q1.x;
                const bool c1 = (A0 - S1) * (A1 - S1) < 0;
                const double B0 = p0.y * dx2 - p0.x * dy2;
                const double B1 = p1.y * dx2 - p1.x * dy2;
                const bool c2 = (B0 - S2) * (B1 - S2) < 0;
                hit = hit || (c1 && c2);
                q0 = q1;
            }
        }
        p0 = p1;
    }
    return hit;
}

// All four waves meet here when predecessors finished while this vehicle was already searching (speculation).
// The new areas enter the soup; every node whose cached edge check said "valid" is re-checked against the new areas
// only: not yet expanded -> it simply becomes invalid; already expanded -> the search so far depended on a wrong
// answer and restarts (returns true).  If no expanded node is hit, the search is exactly the one the reference would
// have run with the areas present from the start: the pop sequence only depends on the validity of popped nodes.
__device__ bool arrival_sync(const Search& S, const CheckCtx& C, const SpecCtx& P, const VState& VS, int tid) {
    __syncthreads();  // #1: nobody reads the soup or the validity cache any more
    const unsigned long long arr = sh_load64(P.sh, SH_ARR_LO);
    const uint32_t nn = P.sh[SH_NNODES];
    incorporate_areas(P, arr, tid);
    __syncthreads();  // #2
    for (uint32_t i0 = (uint32_t)tid; i0 < nn; i0 += blockDim.x) {
        const uint32_t vst = vs_load(VS, i0);
        if (vst == VS_VALID || vst == VS_VALID_CS) {
            bool popped;
            if (node_hits_areas(S, C, P, i0, arr, popped)) {
                if (popped)
                    P.sh[SH_RESTART] = 1;
                else
                    vs_store(VS, i0, VS_INVALID);
            }
        }
    }
    __syncthreads();  // #3
    const bool restart = P.sh[SH_RESTART] != 0;
    __syncthreads();  // #4: everyone has read the verdict
    if (tid == 0) {
        atomicAdd(P.counters + 2, 1);
        if (restart) {
            atomicAdd(P.counters + 1, 1);
            // back to the root before any helper looks at the open list again: node ids are about to be reused
            S.lkey[0] = 0.0;  // (block-min mode rebuilds its own queue after this returns)
            S.lid[0] = 1;
            P.sh[SH_HEAP_LEN] = 1;
            P.sh[SH_EAGER] = 1;
            P.sh[SH_NNODES] = 1;
            for (int c = 0; c < BM_NCAND; ++c) P.sh[SH_CAND + c] = 0;
            P.sh[SH_HINT_SEQ] = 0;  // (the hinted node belongs to the tree that is being thrown away)
            P.sh[SH_HINT_ID] = 0;
            P.sh[SH_VERSION] = P.sh[SH_VERSION] + 1;
            ((uint32_t*)S.gn)[15] &= ~NODE_POPPED_BIT;
            if (S.NL > 0) ((lds_u32*)S.ln)[15] = ((lds_u32*)S.ln)[15] & ~NODE_POPPED_BIT;
        }
        const unsigned long long pend = sh_load64(P.sh, SH_PEND_LO) & ~arr;
        P.sh[SH_PEND_LO] = (uint32_t)pend;
        P.sh[SH_PEND_HI] = (uint32_t)(pend >> 32);
        P.sh[SH_ARR_LO] = 0;
        P.sh[SH_ARR_HI] = 0;
        P.sh[SH_RESTART] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        P.sh[SH_STATE] = ST_RUN;
    }
    __syncthreads();  // #5
    return restart;
}

}  // namespace

// One look at the done flags of the predecessors that are still planning; if some finished, their set is posted in
// SH_ARR and the state goes ST_RUN -> ST_ARRIVED (which loses only against ST_DONE / ST_TIE).  Whole wave calls.
__device__ __forceinline__ bool poll_predecessors(const KernelArgs& A, const SpecCtx& P, volatile lds_u32* l_shared, int lane) {
    const unsigned long long pend = sh_load64(l_shared, SH_PEND_LO);
    if (!pend) return false;
    bool d = false;
    if ((pend >> lane) & 1ull) d = __hip_atomic_load(A.done_flag + P.pred[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == A.epoch;
    const unsigned long long got = __ballot(d);
    if (!got) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        l_shared[SH_ARR_LO] = (uint32_t)got;
        l_shared[SH_ARR_HI] = (uint32_t)(got >> 32);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        atomicCAS((uint32_t*)&l_shared[SH_STATE], ST_RUN, ST_ARRIVED);
    }
    return true;
}

// Everything the search loops need from the prologue (LDS carve, per-vehicle state) and what they hand to the epilogue.
struct Ctx {
    int tid, lane, wave, slot, Hp, n, nw;
    const DevVehicle* V;
    lds_mask64* l_mask;
    lds_i16* l_mi;
    lds_pose* l_pose;
    lds_f64 *l_rx, *l_ry;
    volatile lds_u32* l_shared;
    VState VS;
    lds_f64 *l_dcum, *l_term;
    lds_d2* l_chxy;
    Search S;
    CheckCtx C;
    SpecCtx P;
    pdmpc_vehicle_out* O;
    BmQueue Q;
    lds_u32* l_path;          // uint32[HP_MAX + 2] scratch of the epilogue (path nodes)
    LDS_AS unsigned char* lsm;  // base of the dynamic LDS allocation
    // results
    int status, n_popped;
    uint32_t goal, nnodes;
    bool dep_timeout;
    bool rec_valid = false, rec_written = false;  // bulk kernel: the result record has been written ahead of the publication / at all
    bool published = false;                       // bulk kernel: the done flag is out already (the record's areas are final, only counts and ids may still be written)
    unsigned long long rt_kernel_start = 0;  // s_memrealtime at kernel entry (diagnostics of the bulk kernel)
    bool path_ready = false;  // l_path already holds the nodes of the goal's path (the frontier kernel's counting pass has walked it)
#ifdef PDMPC_PROFILE
    unsigned long long rt_start;
#endif
};

struct ExpandEnv {
    lds_mask64* l_mask;
    lds_i16* l_mi;
    lds_pose* l_pose;
    lds_f64 *l_rx, *l_ry, *l_dcum, *l_term;
    lds_d2* l_chxy;
    int Hp, n, nw, lane;
};

// expand_node.m:1-91 for the popped node `cur` (1-based id, record cn, cos/sin of its yaw): creates the children's
// records (validity unknown) and calls push(mask, active, i0, f, cnt, record) once per 64-trim word of the successor mask, with
// nnodes still the index of the word's first child; the caller's push makes the children visible in its open list.
// Returns false if the arena cannot take the children (nothing is created then).
// FENCE: drain the records' HBM stores before push (needed when push makes the children visible to waves that read
// their records; the expander wave of the block-min mode drains later, before it publishes the new tree size).
template <bool FENCE, int NW, class Push>
__device__ __forceinline__ bool expand_children(const ExpandEnv& E, Search& S, const VState& VS, uint32_t cur, const NodeRec& cn, double cs, double sn,
                                                uint32_t& nnodes, Push push) {
    const int Hp = E.Hp, n = E.n, nw = NW > 0 ? NW : E.nw, lane = E.lane;  // NW > 0: mask words known at compile time
    lds_mask64* l_mask = E.l_mask;
    lds_i16* l_mi = E.l_mi;
    lds_pose* l_pose = E.l_pose;
    lds_f64 *l_rx = E.l_rx, *l_ry = E.l_ry, *l_dcum = E.l_dcum, *l_term = E.l_term;
    lds_d2* l_chxy = E.l_chxy;
    const uint32_t cpk = uni_u(cn.packed);
    const int cTrim = NODE_TRIM(cpk);  // 1-based
    const int cK = NODE_K(cpk);
    const double curX = cn.x, curY = cn.y, curYaw = cn.yaw, curG = cn.g;
    const int k_exp = cK + 1;            // :13
    const int steps_to_go = Hp - k_exp;  // :37
    const lds_mask64* mrow = l_mask + ((size_t)(k_exp - 1) * n + (cTrim - 1)) * nw;
    uint32_t total = 0;
    for (int w = 0; w < nw; ++w) total += (uint32_t)__builtin_popcountll(mrow[w]);
    total = uni_u(total);
    if (nnodes + total > S.max_nodes) {
        return false;
    }
    for (int w = 0; w < nw; ++w) {
        uint64_t mask = mrow[w];
        {
            const uint32_t lo = uni_u((uint32_t)mask), hi = uni_u((uint32_t)(mask >> 32));
            mask = ((uint64_t)hi << 32) | lo;
        }
        const int cnt = __builtin_popcountll(mask);
        const bool active = (mask >> lane) & 1ull;
        const int rank = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        double f = 0.0;
        NodeRec ch;
        uint32_t i0 = 0;
        if (active) {
            const int t2 = w * 64 + lane;  // 0-based successor trim
            const int m = (int)l_mi[(cTrim - 1) * n + t2];
            DevManPose mp;
            mp.dx = l_pose[m].dx;
            mp.dy = l_pose[m].dy;
            mp.dyaw = l_pose[m].dyaw;
            mp.n_cols = l_pose[m].n_cols;
            ch.x = cs * mp.dx - sn * mp.dy + curX;  // :53
            ch.y = sn * mp.dx + cs * mp.dy + curY;  // :54
            ch.yaw = curYaw + mp.dyaw;              // :55
            ch.cs = 0.0;
            ch.sn = 0.0;
            ch.parent = cur;
            ch.packed = (uint32_t)(t2 + 1) | ((uint32_t)k_exp << 10) | ((uint32_t)m << 15) | ((uint32_t)mp.n_cols << 27);
            i0 = nnodes + (uint32_t)rank;  // 0-based index of the child (Tree.add_nodes, Tree.m:61)
            d2 xy;
            xy.x = ch.x;
            xy.y = ch.y;
            l_chxy[rank] = xy;
        }
        wave_sync();
        // The distance terms of the cost-to-come (expand_node.m:57-61, slot 0) and of the cost-to-go (:68-73, slots 1..T),
        // one lane per (child, step): all the sqrt chains of an expansion run side by side, usually in a single pass;
        // the sums below keep the reference's order.
        const int T = steps_to_go;
        // lane layout: child r = idx & (2^sh - 1), step it = idx >> sh (no integer division; a word has <= 16
        // successors, enforced by pdmpc_upload_mpa; the reference MPAs have at most 12, mostly <= 8)
        const int sh = cnt <= 8 ? 3 : 4;
        for (int base = 0; base < ((T + 1) << sh); base += PDMPC_WAVE) {
            const int idx = base + lane;
            const int r = idx & ((1 << sh) - 1);
            const int it = idx >> sh;
            if (r < cnt && it <= T) {
                const d2 xy = l_chxy[r];
                const double ddx = xy.x - l_rx[k_exp - 1 + it], ddy = xy.y - l_ry[k_exp - 1 + it];
                const double nrm = sqrt(ddx * ddx + ddy * ddy);
                double val;
                if (it == 0) {
                    val = nrm * nrm;  // :61
                } else {
                    const double df = nrm - l_dcum[(k_exp - 1) * PDMPC_HP_MAX + (it - 1)];
                    const double m0 = (df > 0) ? df : 0.0;
                    val = m0 * m0;
                }
                l_term[r * PDMPC_HP_MAX + it] = val;
            }
        }
        wave_sync();
        if (active) {
            ch.g = curG + l_term[rank * PDMPC_HP_MAX];  // :61
            double expH = 0.0;
            for (int it = 1; it <= T; ++it) expH = expH + l_term[rank * PDMPC_HP_MAX + it];
            ch.h = expH;
            f = ch.g * 1 + expH * 1;  // GraphSearch.m:100-102
            node_store(S, i0, ch);
            vs_store(VS, i0, 0);  // validity unknown
        }
        // records (and the parent's cos/sin) must be visible to the helper waves before any id reaches the heap: LDS
        // copies are ordered by the in-order DS queue; records that only live in HBM need the stores drained
        if (FENCE && nnodes + (uint32_t)cnt > S.NL) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        wave_sync();
        PROF_STOP(5)
        push(mask, active, i0, f, cnt, ch);
        nnodes += (uint32_t)cnt;
    }
    return true;
}

// The search proper: root node, sequencing wave, helper waves, and the wait for predecessors that are still planning.
// BM = false: the libstdc++-faithful binary heap (exact for any keys).  BM = true: the block-min queue, which is only
// exact while the minimal key is unique; returns true (to every wave) if it met a tie and the search must be redone.
template <int CHECKER, bool BM, int NW>
__device__ __forceinline__ bool search_loops(const KernelArgs& A, Ctx& X) {
    const int tid = X.tid, lane = X.lane, wave = X.wave, slot = X.slot, Hp = X.Hp, n = X.n, nw = X.nw;
    const DevVehicle* __restrict__ V = X.V;
    lds_mask64* l_mask = X.l_mask;
    lds_i16* l_mi = X.l_mi;
    lds_pose* l_pose = X.l_pose;
    lds_f64 *l_rx = X.l_rx, *l_ry = X.l_ry;
    volatile lds_u32* l_shared = X.l_shared;
    const VState& VS = X.VS;
    lds_f64 *l_dcum = X.l_dcum, *l_term = X.l_term;
    lds_d2* l_chxy = X.l_chxy;
    Search& S = X.S;
    const CheckCtx& C = X.C;
    const SpecCtx& P = X.P;
    pdmpc_vehicle_out* __restrict__ O = X.O;
    bool dep_timeout = X.dep_timeout;
    (void)V;
    (void)O;
    (void)slot;

    // ---- root node (GraphSearch.m:34-46)
    uint32_t nnodes = 1;
    if (tid == 0) {
        NodeRec r;
        r.x = V->x0;
        r.y = V->y0;
        r.yaw = V->yaw0;
        r.g = 0.0;
        r.cs = 0.0;
        r.sn = 0.0;
        r.h = 0.0;
        r.parent = 0;
        r.packed = (uint32_t)V->trim0;
        node_store(S, 0, r);
        if (!BM) {
            S.lkey[0] = 0.0;
            S.lid[0] = 1;
        }
        vs_store(VS, 0, 1);  // the root has no edge: valid
        l_shared[SH_HEAP_LEN] = 1;
        l_shared[SH_NNODES] = 1;
        l_shared[SH_EAGER] = 1;
        for (int c = 0; c < BM_NCAND; ++c) l_shared[SH_CAND + c] = 0;
    }
    S.heap_len = 1;
    BmQueue& Q = X.Q;
    if (BM) {
        bm_init(Q, tid, (int)blockDim.x);
        __syncthreads();
        Q.tie = false;
        if (wave == 0) bm_push<true>(Q, lane == 0, 0u, 0.0, 0u, 1u);
    }
    __syncthreads();

    int status = PDMPC_OK;
    int n_popped = 0;
    uint32_t goal = 0;
#ifdef PDMPC_PROFILE
    const unsigned long long rt_search = __builtin_amdgcn_s_memrealtime();
#endif

    if (BM && wave == 0) {
        // ================= queue wave (block-min mode): owns the open list and the pop order ====================
        // GraphSearch.m:53-107 split over two waves.  This one pops (GraphSearch.m:55-56) and discards nodes whose edge
        // is already known to collide (:75-77) on its own; every other node goes to the expander wave, and while that
        // one evaluates the edge and creates the children (:111-196, expand_node.m), this wave removes the node from
        // the open list and finds the best of the remaining entries.  When the reply arrives, the next node to pop is
        // that entry or, if one is smaller, one of the new children — the same node the reference would pop next.
        const double inf = bm_inf();
        uint32_t nn = 1;  // tree size as far as the open list knows (children of the node in flight excluded)
        uint32_t seq = lds_load_u32(&l_shared[SH_Q2E_SEQ]);
        uint32_t waited = 0, ver_ctr = 0;
        // Entries known to collide leave the open list whenever a pop looks at their block (BmDrop).  The reference would
        // have popped and discarded such an entry X (GraphSearch.m:75-77) as soon as it was the minimum of its open
        // list, that is before the first later pop with a larger key.  X was in the open list until it was dropped, so
        // no earlier pop had a larger key; whether a later one has is settled at the end from the log of the pops'
        // keys: X counts as popped iff key(X) < max key of the pops made after it was dropped (equality: the order is
        // the heap's business -> the search is redone on the binary heap).  Keys are not monotone along the pop sequence
        // (a child can have a smaller key than its parent), hence the log instead of a comparison with the last key.
        double* pop_log = A.arena.pop_log + (size_t)slot * A.max_nodes;
        BmDrop D;
        D.on = A.drop_invalid != 0;
        D.validity = VS.l;
        D.gvalidity = A.drop_beyond_lds ? VS.g : nullptr;
        D.nv = VS.NV;
        D.invalid_code = VS_INVALID;
        D.dropped_code = VS_DROPPED;
        D.list = S.gid;
        D.stamps_end = (uint32_t*)(pop_log + A.max_nodes);
        D.stamp = 0;
        D.n = 0;
        // number of dropped entries the reference would have popped, given that the search ends after `n_real` pops
        // (whole wave; sets Q.tie on an undecidable case)
        auto dropped_pops = [&](uint32_t n_real) -> uint32_t {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (the log was written by other lanes of this wave)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // in place: pop_log[j] = max key of the pops j .. n_real - 1
            double carry = 0.0;  // keys are non-negative
            for (int32_t base = (int32_t)((n_real - 1u) & ~63u); base >= 0; base -= 64) {
                const uint32_t j = (uint32_t)base + (uint32_t)lane;
                double v = j < n_real ? pop_log[j] : 0.0;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const double w = __shfl_down(v, o);
                    if (lane + o < 64) v = w > v ? w : v;
                }
                v = carry > v ? carry : v;
                if (j < n_real) pop_log[j] = v;
                carry = lane_d(v, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            uint32_t mine = 0;
            bool same = false;
            for (uint32_t i = (uint32_t)lane; i < D.n; i += (uint32_t)PDMPC_WAVE) {
                const uint32_t t = D.stamps_end[-1 - (int)i];
                if (t < n_real) {
                    const double k = Q.gkey[D.list[i]];
                    const double later = pop_log[t];
                    mine += k < later ? 1u : 0u;
                    same = same || k == later;
                }
            }
            if (__ballot(same)) Q.tie = true;
            uint32_t total = 0;
            for (int l = 0; l < PDMPC_WAVE; ++l) total += lane_u(mine, l);
            return total;
        };
        PROF_DECL
        PROF_START
      q_again:
        D.n = 0;
        D.stamp = 0;
        // Run-ahead list: entries already taken out of the open list, in pop order (ascending keys), entry j in lane j.
        // All but the last are known to collide: they are popped and discarded (GraphSearch.m:75-77) without further
        // ado; the last one is the next node to evaluate.  While the expander wave works on a node, this wave keeps
        // popping until it holds an entry that is not known to collide, so the pops of colliding nodes overlap with the
        // expansion.  If a child of the node in flight turns out to come before some of them, those are put back.
        uint32_t ra_idx = 0xFFFFFFFFu;  // (per lane)
        double ra_key = inf;            // (per lane)
        uint32_t ra_n = 0;
        bool head_in_list = false;  // the last entry is a child that is still in the open list
        bool head_invalid = false;  // the last entry is known to collide as well (the list is full of colliding entries)
        {
            const BmFound r0 = bm_pop(Q, nn, D);  // the root
            if (lane == 0) {
                ra_idx = r0.idx;
                ra_key = r0.key;
            }
            ra_n = r0.idx != 0xFFFFFFFFu ? 1u : 0u;
        }
        for (;;) {
            // (the expander wave is idle whenever this wave is here)
            if (lds_load_u32(&l_shared[SH_STATE]) == ST_ARRIVED) {
                if (lane == 0) {
                    l_shared[SH_NNODES] = nn;
                    l_shared[SH_Q_SYNC] = 1;
                }
                const bool restart = arrival_sync(S, C, P, VS, tid);
                if (lane == 0) l_shared[SH_Q_SYNC] = 0;
                if (restart) {
                    if (lane == 0) atomicAdd(P.counters + 3, n_popped);
                    n_popped = 0;
                    nn = 1;
                    bm_init(Q, lane, PDMPC_WAVE);  // the other waves see an empty candidate list and a tree of one node meanwhile
                    bm_push<true>(Q, lane == 0, 0u, 0.0, 0u, 1u);
                    D.n = 0;
                    D.stamp = 0;
                    const BmFound r0 = bm_pop(Q, nn, D);
                    ra_idx = lane == 0 ? r0.idx : 0xFFFFFFFFu;
                    ra_key = lane == 0 ? r0.key : inf;
                    ra_n = 1;
                    head_in_list = false;
                    head_invalid = false;
                }
                continue;
            }
            if (Q.tie) {
                // the pop order is no longer certified: everybody leaves, the search is redone on the binary heap
                uint32_t old = 0;
                if (lane == 0) old = atomicCAS((uint32_t*)&l_shared[SH_STATE], ST_RUN, ST_TIE);
                if (uni_u(old) == ST_RUN) return true;
                continue;  // an arrival got in first: handle it, then try again
            }
            if (ra_n == 0) {  // pop on an empty queue returns -1 (mex.cpp:87-93)  GraphSearch.m:57-61
                status = PDMPC_EXHAUSTED;
                n_popped += (int)D.n;  // the open list ran empty: everything that was dropped has been "popped" on the way
                if (lane == 0 && D.n) {
                    atomicAdd(A.work_count + 2, (unsigned long long)D.n);
                    atomicAdd(A.work_count + 3, (unsigned long long)D.n);
                }
                break;
            }
            // the colliding entries in front: popped, discarded
            const uint32_t n_dead = ra_n - 1u;
#ifndef PDMPC_PROFILE
            if (A.trace_cap > 0 && (uint32_t)lane <= n_dead && n_popped + lane < A.trace_cap) A.pop_trace[(size_t)slot * A.trace_cap + n_popped + lane] = (int32_t)(ra_idx + 1u);
#endif
            if (D.on && (uint32_t)lane <= n_dead) pop_log[(uint32_t)n_popped + (uint32_t)lane] = ra_key;
            n_popped += (int)ra_n;
            D.stamp = (uint32_t)n_popped;
            const uint32_t cidx = lane_u(ra_idx, (int)n_dead);
            const uint32_t cur = cidx + 1u;
            ra_n = 0;
            PROF_COUNT(7, n_dead)
            PROF_STOP(0)  // loop head
            PROF_TL(seq, 9)
            if (head_invalid) {  // GraphSearch.m:75-77 without leaving this wave (a verdict that comes in after an entry
                                 // was listed is the expander's business)
                const BmFound t = bm_pop(Q, nn, D);
                if (lane == 0) {
                    ra_idx = t.idx;
                    ra_key = t.key;
                    l_shared[SH_VERSION] = ++ver_ctr;
                }
                ra_n = t.idx != 0xFFFFFFFFu ? 1u : 0u;
                head_invalid = ra_n != 0u && uni_u(vs_load(VS, t.idx)) == VS_INVALID;
                PROF_STOP(3)  // pop
                PROF_COUNT(7, 1)
                continue;
            }
            PROF_STOP(1)
            ++seq;
            PROF_TL(seq - 1, 10)
            PROF_TL(seq - 1, 0)
            if (lane == 0) mbox_post(l_shared, SH_Q2E_SEQ, seq, cur);
            if (head_in_list) bm_remove(Q, cidx, nn);
            head_in_list = false;
            PROF_STOP(2)
            // run ahead while the expander works
            for (;;) {
                const BmFound t = bm_pop(Q, nn, D);
                if (t.idx == 0xFFFFFFFFu) break;
                if ((uint32_t)lane == ra_n) {
                    ra_idx = t.idx;
                    ra_key = t.key;
                }
                ++ra_n;
                if (uni_u(vs_load(VS, t.idx)) != VS_INVALID) {
                    // most likely the next node to be handed over: the expander may start on it as soon as it is idle
                    if (lane == 0) mbox_post(l_shared, SH_HINT_SEQ, seq + 1u, t.idx + 1u);
                    break;
                }
                if (ra_n == 8u) {
                    head_invalid = true;
                    break;
                }
            }
            if (lane == 0) l_shared[SH_VERSION] = ++ver_ctr;
            PROF_STOP(3)
            PROF_TL(seq - 1, 1)
            uint32_t spins = 0;
            unsigned long long reply;
            while ((uint32_t)(reply = mbox_read(l_shared, SH_E2Q_SEQ)) != seq) {
                if (A.crowded) __builtin_amdgcn_s_sleep(6);  // leave the issue slots to the waves that have work
                if (++spins > A.spin_limit) break;  // (cannot happen: the expander always answers)
            }
            PROF_STOP(4)  // waiting for the expander
            PROF_TL(seq - 1, 2)
            const uint32_t flags = (uint32_t)(reply >> 32) & 0xFFu;
            const uint32_t cnt = (uint32_t)(reply >> 40);
            if (spins > A.spin_limit) {
                dep_timeout = true;
                status = PDMPC_EXHAUSTED;
                break;
            }
            if (flags & (E2Q_GOAL | E2Q_OVERFLOW)) {
                // the search ends with this node: the dropped entries that come before it were popped by the reference
                if (D.n) {
                    const uint32_t counted = dropped_pops((uint32_t)n_popped);
                    n_popped += (int)counted;
                    if (Q.tie) continue;
                    if (lane == 0) {
                        atomicAdd(A.work_count + 2, (unsigned long long)D.n);
                        atomicAdd(A.work_count + 3, (unsigned long long)counted);
                    }
                }
                if (flags & E2Q_GOAL)  // :81-90
                    goal = cur;
                else
                    status = PDMPC_ARENA_OVERFLOW;
                break;
            }
            if (cnt) {
                // the children are nodes nn .. nn + cnt - 1; their keys sit in the ring.  Make them visible, then see
                // where the smallest of them falls among the entries of the run-ahead list.
                for (uint32_t base = 0; base < cnt; base += (uint32_t)PDMPC_WAVE) {
                    const bool active = base + (uint32_t)lane < cnt;
                    const uint32_t i0 = nn + base + (uint32_t)lane;
                    const double f = active ? *(volatile lds_f64*)&Q.kring[i0 & Q.kr_mask] : inf;
                    bm_push<false>(Q, active, i0, f, nn + base, nn + (base + PDMPC_WAVE < cnt ? base + PDMPC_WAVE : cnt));
                    const double limit = ra_n ? lane_d(ra_key, (int)ra_n - 1) : inf;  // the key of the list's last entry
                    if (__ballot(active && f == limit)) Q.tie = true;
                    const unsigned long long better = __ballot(f < limit);
                    if (better) {
                        int l = __builtin_ctzll(better);
                        if (better & (better - 1ull)) {  // several: the smallest of them
                            const double mn = wave_min_d(f);
                            const unsigned long long at = __ballot(f == mn);
                            if (at & (at - 1ull)) Q.tie = true;
                            l = __builtin_ctzll(at);
                        }
                        const double ck = lane_d(f, l);
                        const bool listed = (uint32_t)lane < ra_n;
                        if (__ballot(listed && ra_key == ck)) Q.tie = true;
                        const uint32_t pos = (uint32_t)__builtin_popcountll(__ballot(listed && ra_key < ck));
                        // entries pos .. ra_n - 1 come after the child: back into the open list (a listed child that is
                        // still in the open list just drops off the list)
                        const bool back = listed && (uint32_t)lane >= pos && !(head_in_list && (uint32_t)lane == ra_n - 1u);
                        bm_unpop_lanes(Q, back, ra_idx, ra_key, nn + cnt);
                        if ((uint32_t)lane == pos) {
                            ra_idx = nn + base + (uint32_t)l;
                            ra_key = ck;
                        }
                        ra_n = pos + 1u;
                        head_in_list = true;
                        head_invalid = false;
                    }
                }
                nn += cnt;
            }
            PROF_STOP(5)  // children made visible
            PROF_TL(seq - 1, 8)
        }
#ifdef PDMPC_PROFILE
        if (lane == 0) {
            for (int i = 0; i < 8; ++i) ((double*)O->shapes[PDMPC_HP_MAX - 1])[i] = (double)S.prof_acc[i];  // unused tail of the record
        }
#endif
        // finished — but predecessors that are still planning may yet invalidate what we found
        for (;;) {
            const uint32_t st = lds_load_u32(&l_shared[SH_STATE]);
            if (st == ST_ARRIVED) {
                if (lane == 0) {
                    l_shared[SH_NNODES] = nn;
                    l_shared[SH_Q_SYNC] = 1;
                }
                const bool restart = arrival_sync(S, C, P, VS, tid);
                if (lane == 0) l_shared[SH_Q_SYNC] = 0;
                if (restart) {
                    if (lane == 0) atomicAdd(P.counters + 3, n_popped);
                    n_popped = 0;
                    goal = 0;
                    status = PDMPC_OK;
                    nn = 1;
                    bm_init(Q, lane, PDMPC_WAVE);
                    Q.tie = false;
                    bm_push<true>(Q, lane == 0, 0u, 0.0, 0u, 1u);
                    goto q_again;
                }
                continue;
            }
            if (sh_load64(l_shared, SH_PEND_LO) == 0ull) {
                uint32_t old = 0;
                if (lane == 0) old = atomicCAS((uint32_t*)&l_shared[SH_STATE], ST_RUN, ST_DONE);
                if (uni_u(old) == ST_RUN) break;
                continue;
            }
            __builtin_amdgcn_s_sleep(8);
            if (++waited > A.spin_limit) {  // a predecessor never finished: give up on it (reported as an error status)
                dep_timeout = true;
                if (lane == 0) {
                    l_shared[SH_PEND_LO] = 0;
                    l_shared[SH_PEND_HI] = 0;
                }
            }
        }
        nnodes = lds_load_u32(&l_shared[SH_NNODES]);  // the expander wave's count (it is idle: all replies are in)
    } else if (BM && wave == 1) {
        // ================= expander wave (block-min mode) ========================================================
        // Takes the popped nodes the queue wave hands over: eval_edge_exact (GraphSearch.m:111-196) unless a validator
        // wave has the verdict already, then expand_node.m.  Children: records into the tree, keys into the key ring and
        // HBM; the queue wave learns how many there are and makes them visible in the open list.
        ExpandEnv EE;
        EE.l_mask = l_mask;
        EE.l_mi = l_mi;
        EE.l_pose = l_pose;
        EE.l_rx = l_rx;
        EE.l_ry = l_ry;
        EE.l_dcum = l_dcum;
        EE.l_term = l_term;
        EE.l_chxy = l_chxy;
        EE.Hp = Hp;
        EE.n = n;
        EE.nw = nw;
        EE.lane = lane;
        uint32_t seen = lds_load_u32(&l_shared[SH_Q2E_SEQ]);
        uint32_t ver_e = 0;
        PROF_DECL
        PROF_START
        // Speculation: when the queue wave already knows which node it will most likely hand over next (the head of its
        // run-ahead list), this wave evaluates and expands that node as soon as it is idle and keeps the outcome to
        // itself: children records and keys are written (beyond the published tree size nobody looks), the reply is held
        // back.  If the node is then handed over, the reply goes out at once; if another node comes (a child of the
        // previous one came first), the tree size is wound back and the work is redone for the right node.
        bool spec_valid = false;
        uint32_t spec_id = 0, spec_flags = 0, spec_cnt = 0, spec_n0 = 0;
        for (;;) {
            const unsigned long long post = mbox_read(l_shared, SH_Q2E_SEQ);
            const uint32_t sq = (uint32_t)post;
            uint32_t cur;
            bool spec = false, answered = false;
            uint32_t flags = 0, cnt = 0;
            if (sq != seen) {
                seen = sq;
                PROF_STOP(8)  // idle
                PROF_TL(sq - 1, 3)
                cur = (uint32_t)(post >> 32);
                if (spec_valid) {
                    spec_valid = false;
                    if (spec_id == cur) {  // guessed right: the answer is ready
                        flags = spec_flags;
                        cnt = spec_cnt;
                        PROF_COUNT(13, 1)
                        answered = true;
                    } else {
                        nnodes = spec_n0;  // guessed wrong: the speculative children are forgotten
                    }
                }
            } else {
                const uint32_t state = lds_load_u32(&l_shared[SH_STATE]);
                if (state == ST_DONE || state == ST_TIE) break;
                // nothing handed over.  An arrival is joined only once the queue wave is in it: before that it may
                // still hand a node over and wait for the reply.
                if (state == ST_ARRIVED && lds_load_u32(&l_shared[SH_Q_SYNC]) != 0) {
                    if (spec_valid) {  // the verdict may change with the new areas
                        spec_valid = false;
                        nnodes = spec_n0;
                    }
                    if (arrival_sync(S, C, P, VS, tid)) nnodes = 1;
                    continue;
                }
                const unsigned long long hint = mbox_read(l_shared, SH_HINT_SEQ);
                if (!spec_valid && A.speculate_expansion && (uint32_t)hint == seen + 1u) {
                    cur = (uint32_t)(hint >> 32);
                    spec = true;
                } else {
                    if (A.crowded) __builtin_amdgcn_s_sleep(6);  // leave the issue slots to the waves that have work
                    continue;
                }
            }
            const uint32_t n_before = nnodes;
            if (!answered) {
            const uint32_t c0 = cur - 1;
            const uint32_t vs = uni_u(vs_load(VS, c0));
            bool valid;
            if (vs == VS_UNKNOWN || vs == VS_CLAIMED) {
                valid = edge_valid<CHECKER>(S, C, cur, lane);
                if (lane == 0) vs_store(VS, c0, valid ? VS_VALID : VS_INVALID);
                PROF_COUNT(15, 1)
            } else {
                valid = (vs == VS_VALID || vs == VS_VALID_CS);
            }
            PROF_STOP(9)  // validity
            PROF_COUNT(14, 1)
            PROF_TL(seen - 1 + (spec ? 1 : 0), 5)
            if (valid) {
                flags = E2Q_VALID;
                const NodeRec cn = node_load(S, c0);  // same record in every lane
                const uint32_t cpk = uni_u(cn.packed);
                if (lane == 0) node_mark_popped(S, c0, cpk);  // a later arrival that hits this node forces a restart
                PROF_TL(seen - 1 + (spec ? 1 : 0), 6)
                if (NODE_K(cpk) == Hp) {
                    flags |= E2Q_GOAL;
                } else {
                    double sn, cs;
                    if (vs == VS_VALID_CS) {  // a validator already evaluated expand_node.m:50-51 for this node
                        cs = cn.cs;
                        sn = cn.sn;
                    } else {
                        pdmpc_sincos(cn.yaw, &sn, &cs);  // expand_node.m:50-51
                        if (lane == 0) node_store_cs(S, c0, cs, sn);
                    }
                    const uint32_t n0 = nnodes;
                    const bool fits = expand_children<false, NW>(EE, S, VS, cur, cn, cs, sn, nnodes, [&](uint64_t mask, bool active, uint32_t i0, double f, int ccnt, const NodeRec&) {
                        (void)mask;
                        (void)ccnt;
                        if (active) {
                            Q.kring[i0 & Q.kr_mask] = f;
                            Q.gkey[i0] = f;
                        }
                    });
                    if (!fits) flags |= E2Q_OVERFLOW;
                    cnt = nnodes - n0;
                    PROF_TL(seen - 1 + (spec ? 1 : 0), 7)
                }
            }
            }
            if (spec) {
                spec_valid = true;
                spec_id = cur;
                spec_flags = flags;
                spec_cnt = cnt;
                spec_n0 = n_before;
                PROF_STOP(11)  // speculative work
                continue;
            }
            if (lane == 0) {
                mbox_post(l_shared, SH_E2Q_SEQ, seen, flags | (cnt << 8));  // (after the key stores of all lanes: LDS order)
                PROF_TL(seen - 1, 4)
            }
            if (cnt) {
                // The queue wave needed the keys (in the ring) and the count only.  The scout and validator waves read
                // the children's records: those that live in HBM only must have arrived before the new tree size shows.
                if (nnodes > S.NL) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) {
                    l_shared[SH_NNODES] = nnodes;
                    l_shared[SH_VERSION] = 0x80000000u | ++ver_e;
                }
            }
            PROF_STOP(10)  // expansion + reply
        }
#ifdef PDMPC_PROFILE
        if (lane == 0) {
            for (int i = 8; i < 16; ++i) ((double*)O->shapes[PDMPC_HP_MAX - 1])[i] = (double)S.prof_acc[i];
        }
#endif
    } else if (BM && wave == 2) {
        // ================= scout wave (block-min mode) ==========================================================
        // Lists the nodes that will be popped next and whose edge nobody has evaluated yet: the unknown entries with
        // the smallest keys among the four best blocks of the best group.  Its reads race with the sequencing wave's
        // updates; a stale or torn view only costs a useless proposal (every id below the published tree size is a
        // complete node, and edge validity is a pure function of the tree and the soups).  Also polls the predecessors.
        const double inf = bm_inf();
        uint32_t last_ver = 0xFFFFFFFFu, cand_ver = 0;
        for (;;) {
            const uint32_t state = lds_load_u32(&l_shared[SH_STATE]);
            if (state == ST_DONE || state == ST_TIE) break;
            if (state == ST_ARRIVED) {
                (void)arrival_sync(S, C, P, VS, tid);
                continue;
            }
            if (poll_predecessors(A, P, l_shared, lane)) continue;
            const uint32_t ver = lds_load_u32(&l_shared[SH_VERSION]);
            if (ver == last_ver) __builtin_amdgcn_s_sleep(8);  // nothing popped or pushed since the last scan: rescan lazily
            last_ver = ver;
            const uint32_t nn = lds_load_u32(&l_shared[SH_NNODES]);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            uint32_t g = 0;
            bool any = true;
            if (nn > 4096u) {
                const double v2 = *(volatile lds_f64*)&Q.m2[lane];
                const double mn2 = wave_min_d(v2);
                any = mn2 < inf;
                if (any) g = (uint32_t)__builtin_ctzll(__ballot(v2 == mn2));
            }
            double v1 = any ? *(volatile lds_f64*)&Q.m1[g * 64u + (uint32_t)lane] : inf;
            double kk[4];
            uint32_t blk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                kk[j] = inf;
                blk[j] = 0;
                const double mn = wave_min_d(v1);
                if (mn < inf) {
                    const uint32_t bl = (uint32_t)__builtin_ctzll(__ballot(v1 == mn));
                    if ((uint32_t)lane == bl) v1 = inf;
                    const uint32_t b = g * 64u + bl;
                    const uint32_t idx = b * 64u + (uint32_t)lane;
                    blk[j] = b;
                    double k = inf;
                    if (idx < nn) {
                        if (b * 64u + Q.kr_mask + 1u >= nn)
                            k = *(volatile lds_f64*)&Q.kring[idx & Q.kr_mask];
                        else
                            k = __hip_atomic_load(Q.gkey + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (k < inf && vs_load(VS, idx) != VS_UNKNOWN) k = inf;
                    }
                    kk[j] = k;
                }
            }
            uint32_t my = 0;  // lane c keeps candidate c
#pragma unroll
            for (int c = 0; c < BM_NCAND; ++c) {
                double loc = kk[0];
                int js = 0;
#pragma unroll
                for (int j = 1; j < 4; ++j) {
                    if (kk[j] < loc) {
                        loc = kk[j];
                        js = j;
                    }
                }
                const double mn = wave_min_d(loc);
                uint32_t id = 0;
                if (mn < inf) {
                    const int l = __builtin_ctzll(__ballot(loc == mn));
                    uint32_t bsel = blk[0];
#pragma unroll
                    for (int j = 1; j < 4; ++j)
                        if (js == j) bsel = blk[j];
                    id = lane_u(bsel, l) * 64u + (uint32_t)l + 1u;
                    if (lane == l) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (js == j) kk[j] = inf;
                    }
                }
                if (lane == c) my = id;
            }
            if (lane < BM_NCAND) l_shared[SH_CAND + lane] = my;
            if (lane == 0) l_shared[SH_CAND_VER] = ++cand_ver;
        }
    } else if (BM && wave != 0) {
        // ================= validator waves (block-min mode) =====================================================
        // Take the most urgent proposal nobody has claimed yet, evaluate its edge, publish the verdict (and cos/sin of
        // the node's yaw for the expansion).  Same guarantees as the helper waves of the binary-heap mode.
        uint32_t seen = 0xFFFFFFFFu;
        for (;;) {
            const uint32_t state = lds_load_u32(&l_shared[SH_STATE]);
            if (state == ST_DONE || state == ST_TIE) break;
            if (state == ST_ARRIVED) {
                (void)arrival_sync(S, C, P, VS, tid);
                continue;
            }
            if (wave - 3 >= A.n_validators) {  // (tuning knob: this wave sits the search out)
                __builtin_amdgcn_s_sleep(32);
                continue;
            }
            const uint32_t cver = lds_load_u32(&l_shared[SH_CAND_VER]);
            uint32_t id = 0;
            if (lane < BM_NCAND) id = l_shared[SH_CAND + lane];
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const bool unknown = id != 0 && vs_load(VS, id - 1) == VS_UNKNOWN;
            unsigned long long b = __ballot(unknown);
            uint32_t target = 0;
            while (b && !target) {
                const int l = __builtin_ctzll(b);
                b &= b - 1;
                const uint32_t cid = lane_u(id, l);
                uint32_t won = 0;
                if (lane == 0) won = vs_claim(VS, cid - 1) ? 1u : 0u;
                if (uni_u(won)) target = cid;
            }
            bool more = false;
            if (!target && A.eager_validation) {
                // Nothing urgent: evaluate the edges of the tree in the order the nodes were created.  Nodes are popped
                // long after they are created (tools/pop_age.py), so by the time the queue wave meets them the verdict is
                // in: colliding ones are dropped from the open list on the side, the others expand without waiting.
                const uint32_t c = lds_load_u32(&l_shared[SH_EAGER]);
                if (c < lds_load_u32(&l_shared[SH_NNODES])) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    more = true;
                    uint32_t won = 0;
                    if (lane == 0) won = (atomicCAS((uint32_t*)&l_shared[SH_EAGER], c, c + 1u) == c && vs_claim(VS, c)) ? 1u : 0u;
                    if (uni_u(won)) target = c + 1u;
                }
            }
            if (target) {
                const bool ok = edge_valid<CHECKER>(S, C, target, lane);
                uint32_t verdict = ok ? VS_VALID : VS_INVALID;
                if (ok) {
                    const NodeRec tn = node_load(S, target - 1);
                    if (NODE_K(uni_u(tn.packed)) < Hp) {
                        double sn, cs;
                        pdmpc_sincos(tn.yaw, &sn, &cs);
                        if (lane == 0) node_store_cs(S, target - 1, cs, sn);
                        if (target - 1 >= S.NL) __threadfence_block();
                        verdict = VS_VALID_CS;
                    }
                }
                if (lane == 0) vs_store(VS, target - 1, verdict);
            } else if (more) {
                // (lost the race for that node: try the next one)
            } else {
                // nothing to do until the scout rewrites the list
                uint32_t naps = 0;
                while (cver == seen && lds_load_u32(&l_shared[SH_CAND_VER]) == cver && lds_load_u32(&l_shared[SH_STATE]) == ST_RUN && naps < 64u) {
                    __builtin_amdgcn_s_sleep(2);
                    ++naps;
                }
                seen = cver;
            }
        }
    } else if (wave != 0) {
        // ================= helper waves: pre-validate the nodes near the top of the open list ==================
        // Any id read from the heap is a fully written node (the sequencing wave publishes ids after the records);
        // edge validity is a pure function, so evaluating it early, twice, or for a node that is never popped
        // cannot change the search.  Results land in the shared validity cache (1 = valid, 2 = invalid).
        const int share = wave - 1;
        uint32_t iter = 0;
        for (;;) {
            const uint32_t state = lds_load_u32(&l_shared[SH_STATE]);
            if (state == ST_DONE) break;
            if (state == ST_ARRIVED) {
                (void)arrival_sync(S, C, P, VS, tid);
                continue;
            }
            // the last helper wave also polls the pending predecessors' done flags (every 4th round and while idle)
            if (wave == (int)(blockDim.x >> 6) - 1 && (iter++ & 3u) == 0) {
                if (poll_predecessors(A, P, l_shared, lane)) continue;
            }
            const uint32_t ver = lds_load_u32(&l_shared[SH_VERSION]);
            const uint32_t hl = lds_load_u32(&l_shared[SH_HEAP_LEN]);
            const uint32_t K = hl < (uint32_t)PDMPC_WAVE ? hl : (uint32_t)PDMPC_WAVE;
            uint32_t id = 0;
            if ((uint32_t)lane < K) id = *(volatile lds_u32*)&S.lid[lane];
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const bool unknown = id != 0 && vs_load(VS, id - 1) == VS_UNKNOWN;
            unsigned long long b = __ballot(unknown);
            int skip = share;
            uint32_t target = 0;
            while (b) {
                const int l = __builtin_ctzll(b);
                b &= b - 1;
                if (skip == 0) {
                    target = lane_u(id, l);
                    break;
                }
                --skip;
            }
            if (target) {
                const bool ok = edge_valid<CHECKER>(S, C, target, lane);
                uint32_t verdict = ok ? VS_VALID : VS_INVALID;
                if (ok) {
                    // the sequencer will expand this node if it pops it: take cos/sin(yaw) off its critical path
                    const NodeRec tn = node_load(S, target - 1);
                    if (NODE_K(uni_u(tn.packed)) < Hp) {
                        double sn, cs;
                        pdmpc_sincos(tn.yaw, &sn, &cs);
                        if (lane == 0) node_store_cs(S, target - 1, cs, sn);
                        if (target - 1 >= S.NL) __threadfence_block();
                        verdict = VS_VALID_CS;
                    }
                }
                if (lane == 0) vs_store(VS, target - 1, verdict);
            } else {
                // nothing to validate in this view of the open list: sleep until the sequencing wave changes it (one LDS
                // word per poll, so idle helpers do not compete with the sequencer for LDS bandwidth)
                uint32_t naps = 0;
                while (lds_load_u32(&l_shared[SH_VERSION]) == ver && lds_load_u32(&l_shared[SH_STATE]) == ST_RUN && naps < 64u) {
                    __builtin_amdgcn_s_sleep(4);
                    ++naps;
                }
            }
        }
    } else {
        // ================= sequencing wave: GraphSearch.m:53-107 =================================================
        PROF_DECL
        PROF_START
        uint32_t waited = 0, ver_ctr = 0;
      search_again:
        for (;;) {
            if (lds_load_u32(&l_shared[SH_STATE]) == ST_ARRIVED) {
                // predecessors finished while we were searching: fold their areas in; restart only if an already
                // expanded node turns out to collide with them
                if (lane == 0) l_shared[SH_NNODES] = nnodes;
                if (arrival_sync(S, C, P, VS, tid)) {
                    S.heap_len = 1;
                    nnodes = 1;
                    n_popped = 0;
                    continue;
                }
            }
            if (S.heap_len == 0) {  // pop on an empty queue returns -1 (mex.cpp:87-93)  GraphSearch.m:57-61
                status = PDMPC_EXHAUSTED;
                break;
            }
            PROF_STOP(7)
            const uint32_t cur = uni_u(*(volatile lds_u32*)&S.lid[0]);  // 1-based node id (HL >= 1: the top is always in LDS)
            heap_pop(S);
            if (lane == 0) {
                l_shared[SH_HEAP_LEN] = S.heap_len;
                l_shared[SH_VERSION] = ++ver_ctr;
            }
            PROF_STOP(0)
            if (A.trace_cap > 0 && n_popped < A.trace_cap && lane == 0) A.pop_trace[(size_t)slot * A.trace_cap + n_popped] = (int32_t)cur;
            ++n_popped;
            const uint32_t c0 = cur - 1;

            // ---- eval_edge_exact (GraphSearch.m:111-196): from the validity cache if a helper got there first
            const uint32_t vs = uni_u(vs_load(VS, c0));
            bool valid;
            if (vs == VS_UNKNOWN) {
                valid = edge_valid<CHECKER>(S, C, cur, lane);
                if (lane == 0) vs_store(VS, c0, valid ? VS_VALID : VS_INVALID);
                PROF_COUNT(13, 1)
            } else {
                valid = (vs == VS_VALID || vs == VS_VALID_CS);
            }
            PROF_STOP(2)
            if (!valid) continue;  // GraphSearch.m:75-77

            const NodeRec cn = node_load(S, c0);  // same record in every lane
            const uint32_t cpk = uni_u(cn.packed);
            if (lane == 0) node_mark_popped(S, c0, cpk);  // a later arrival that hits this node forces a restart
            if (NODE_K(cpk) == Hp) {  // :81-90
                goal = cur;
                break;
            }

            // ---- expand_node.m:1-91
            const double curYaw = cn.yaw;
            double sn, cs;
            PROF_STOP(3)
            if (vs == VS_VALID_CS) {  // a helper already evaluated expand_node.m:50-51 for this node
                cs = cn.cs;
                sn = cn.sn;
            } else {
                pdmpc_sincos(curYaw, &sn, &cs);  // expand_node.m:50-51
                if (lane == 0) node_store_cs(S, c0, cs, sn);
            }
            PROF_STOP(4)
            {
                ExpandEnv EE;
                EE.l_mask = l_mask;
                EE.l_mi = l_mi;
                EE.l_pose = l_pose;
                EE.l_rx = l_rx;
                EE.l_ry = l_ry;
                EE.l_dcum = l_dcum;
                EE.l_term = l_term;
                EE.l_chxy = l_chxy;
                EE.Hp = Hp;
                EE.n = n;
                EE.nw = nw;
                EE.lane = lane;
                // pq.push(new_open_nodes, new_open_values): one at a time in ascending trim order (mex.cpp:67-72)
                const bool fits = expand_children<true, NW>(EE, S, VS, cur, cn, cs, sn, nnodes, [&](uint64_t mask, bool active, uint32_t i0, double f, int cnt, const NodeRec&) {
                    (void)active;
                    (void)i0;
                    uint64_t mm = mask;
                    uint32_t r = 0;
                    while (mm) {
                        const int l = __builtin_ctzll(mm);
                        mm &= mm - 1;
                        const double fk = lane_d(f, l);
                        heap_push(S, nnodes + r + 1, fk);
                        ++r;
                    }
                    if (lane == 0) {
                        l_shared[SH_HEAP_LEN] = S.heap_len;
                        l_shared[SH_NNODES] = nnodes + (uint32_t)cnt;
                        l_shared[SH_VERSION] = ++ver_ctr;
                    }
                });
                if (!fits) {
                    status = PDMPC_ARENA_OVERFLOW;
                    break;
                }
                PROF_STOP(6)
            }
        }
#ifdef PDMPC_PROFILE
        if (lane == 0)
        {
            for (int i = 0; i < PROF_N; ++i) ((double*)O->shapes[PDMPC_HP_MAX - 1])[i] = (double)S.prof_acc[i];  // unused tail of the record
            const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
            O->path_nodes[PDMPC_HP_MAX][0] = (double)(rt_search - X.rt_start);  // 100 MHz ticks: prologue + predecessor wait
            O->path_nodes[PDMPC_HP_MAX][1] = (double)(rt_end - rt_search);    // search
            O->path_nodes[PDMPC_HP_MAX][2] = (double)X.rt_start;
            O->path_nodes[PDMPC_HP_MAX][3] = (double)rt_end;
        }
#endif
        // finished — but predecessors that are still planning may yet invalidate what we found
        for (;;) {
            const uint32_t st = lds_load_u32(&l_shared[SH_STATE]);
            if (st == ST_ARRIVED) {
                if (lane == 0) l_shared[SH_NNODES] = nnodes;
                if (arrival_sync(S, C, P, VS, tid)) {
                    S.heap_len = 1;
                    nnodes = 1;
                    n_popped = 0;
                    goal = 0;
                    status = PDMPC_OK;
                    goto search_again;
                }
                continue;
            }
            if (sh_load64(l_shared, SH_PEND_LO) == 0ull) {
                uint32_t old = 0;
                if (lane == 0) old = atomicCAS((uint32_t*)&l_shared[SH_STATE], ST_RUN, ST_DONE);
                if (uni_u(old) == ST_RUN) break;
                continue;
            }
            __builtin_amdgcn_s_sleep(8);
            if (++waited > A.spin_limit) {  // a predecessor never finished: give up on it (reported as an error status)
                dep_timeout = true;
                if (lane == 0) {
                    l_shared[SH_PEND_LO] = 0;
                    l_shared[SH_PEND_HI] = 0;
                }
            }
        }
    }
    X.status = status;
    X.n_popped = n_popped;
    X.goal = goal;
    X.nnodes = nnodes;
    X.dep_timeout = dep_timeout;
    return BM && lds_load_u32(&l_shared[SH_STATE]) == ST_TIE;  // (helper waves; the sequencing wave returned above)
}

// Prologue shared by the kernels: carves the LDS allocation (offsets from KernelArgs::lds), stages the MPA tables, the vehicle
// record and its obstacle soups, initialises the result record and the predecessor bookkeeping, and fills the context X.
__device__ __forceinline__ void search_prologue(const KernelArgs& A, Ctx& X, LDS_AS unsigned char* lsm, bool interx) {
    const int tid = threadIdx.x;
    const int lane = tid & (PDMPC_WAVE - 1);
    const int wave = uni_i(tid >> 6);
    const int slot = A.first + (A.reverse_dispatch ? A.n_searches - 1 - (int)blockIdx.x : (int)blockIdx.x);
    const int Hp = A.Hp;
    const int n = A.n_trims;
    const int nw = A.n_words;
    const DevVehicle* __restrict__ V = A.veh + slot;
#ifdef PDMPC_PROFILE
    const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- LDS carve
    lds_mask64* l_mask = (lds_mask64*)(lsm + A.lds.mask);
    lds_i16* l_mi = (lds_i16*)(lsm + A.lds.man_index);
    lds_pose* l_pose = (lds_pose*)(lsm + A.lds.pose);
    lds_f64* l_rx = (lds_f64*)(lsm + A.lds.ref);
    lds_f64* l_ry = l_rx + PDMPC_HP_MAX;
    lds_f64* l_dtv = l_ry + PDMPC_HP_MAX;
    lds_u32* l_path = (lds_u32*)(lsm + A.lds.path);
    lds_i32* l_soff = (lds_i32*)(l_path + PDMPC_HP_MAX + 2);  // soup offsets [Hp+1], hdv offsets [Hp+1]
    lds_i32* l_hoff = l_soff + PDMPC_HP_MAX + 1;
    volatile lds_u32* l_shared = (volatile lds_u32*)(l_hoff + PDMPC_HP_MAX + 1);
    lds_i32* l_lit = (lds_i32*)(l_shared + SH_WORDS);  // literal soup length per step
    lds_d2* l_soup = (lds_d2*)(lsm + A.lds.soup);
    VState VS;
    VS.l = (volatile lds_u8*)(lsm + A.lds.vstate);
    VS.NV = (uint32_t)A.NV;
    lds_f64* l_dcum = (lds_f64*)(lsm + A.lds.expand);             // [HP_MAX][HP_MAX] cumulative dt*v_ref per (k_exp, t)
    lds_f64* l_term = l_dcum + PDMPC_HP_MAX * PDMPC_HP_MAX;       // [16 children][HP_MAX] cost-to-go terms
    lds_d2* l_chxy = (lds_d2*)(l_term + 16 * PDMPC_HP_MAX);       // [16] child positions

    Search S;
    S.ln = (lds_d2*)(lsm + A.lds.nodes);
    S.lkey = (lds_f64*)(lsm + A.lds.heap_key);
    S.lid = (lds_u32*)(lsm + A.lds.heap_id);
    S.NL = (uint32_t)A.NL;
    S.HL = (uint32_t)A.HL;
    S.max_nodes = A.max_nodes;
    S.lane = lane;
    S.pl = make_pop_lane(lane);
    const size_t voff = (size_t)slot * A.max_nodes;
    S.gn = A.arena.nodes + voff;
    S.gkey = A.arena.heap_key + voff;
    S.gid = A.arena.heap_id + voff;
    VS.g = A.arena.vstate + voff;

    CheckCtx C;
    C.l_area = (const lds_d2*)(lsm + A.lds.area);
    C.g_area = (const d2*)A.man_area;
    C.l_soup = l_soup;
    C.l_soff = l_soff;
    C.l_hoff = l_hoff;
    C.areas_in_lds = A.areas_in_lds;
    C.Hp = Hp;
    C.checker = A.checker;
    C.sh = (lds_d2*)(lsm + A.lds.shape) + wave * (2 * PDMPC_VMAX + 1);
    C.tally = (LDS_AS unsigned long long*)(C.sh + 2 * PDMPC_VMAX);  // [0] edge checks, [1] segment pairs (this wave)
    if (lane == 0) {
        C.tally[0] = 0;
        C.tally[1] = 0;
    }
    C.cand = (lds_u32*)(lsm + A.lds.cand) + (size_t)wave * A.cand_cap;

    pdmpc_vehicle_out* __restrict__ O = A.out + slot;

    // ---- prologue 1: stage MPA tables (coalesced 16-byte copies; the host pads every table to 16 B)
    {
        const int mask_bytes = Hp * n * nw * 8;
        stage16(l_mask, A.succ_mask, (mask_bytes + 15) / 16, tid);
        stage16(l_mi, A.man_index, (n * n * 2 + 15) / 16, tid);
        stage16(l_pose, A.man_pose, A.n_man * 2, tid);
        if (A.areas_in_lds) stage16(lsm + A.lds.area, A.man_area, A.n_man * 3 * PDMPC_VMAX, tid);
    }
    // ---- prologue 2: vehicle record, result record defaults
    if (tid < Hp) {
        l_rx[tid] = V->ref_x[tid];
        l_ry[tid] = V->ref_y[tid];
        l_dtv[tid] = A.dt * V->v_ref[tid];  // options.dt_seconds * iter.v_ref(k)   expand_node.m:70
    }
    if (tid >= PDMPC_WAVE && tid < PDMPC_WAVE + Hp) {
        // d_traveled_max of expand_node.m:66-70 for every expansion step k_exp = tid - 63 (1-based): the running sum
        // dt*v_ref(k_exp+1) + ... in the reference's order, so the bits match the in-loop accumulation
        const int k_exp = tid - PDMPC_WAVE + 1;
        double d = 0.0;
        for (int it = 1; it <= Hp - k_exp; ++it) {
            d = d + A.dt * V->v_ref[k_exp + it - 1];
            l_dcum[(k_exp - 1) * PDMPC_HP_MAX + (it - 1)] = d;
        }
    }
    if (tid == 0) {
        for (int i = 0; i < SH_WORDS; ++i) l_shared[i] = 0;
    }
    {
        // zero the record; y_predicted starts as NaN (ControlResultsInfo.m:40)
        double* od = (double*)O;
        const int nd = (int)(sizeof(pdmpc_vehicle_out) / 8);
        const int y0 = (int)(offsetof(pdmpc_vehicle_out, y_predicted) / 8);
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        for (int i = tid; i < nd; i += (int)blockDim.x) od[i] = (i >= y0 && i < y0 + PDMPC_HP_MAX * 3) ? qnan : 0.0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // later result stores hit the same bytes from other lanes
    }
    __syncthreads();

    // ---- prologue 3: obstacle soup of every step: [literal polygons + NaN][predecessor areas padded to VMAX]
    const int n_pred = V->n_pred;
    const int pred_cols = n_pred * PDMPC_VMAX;
    {
        int off = 0;
        for (int k = 0; k < Hp; ++k) {
            const int a = V->lit_off[k], b = V->lit_off[k + 1];
            if (tid == 0) {
                l_soff[k] = off;
                l_lit[k] = b - a;
            }
            stage16(l_soup + off, (const d2*)A.points + a, b - a, tid);
            off += (b - a) + pred_cols;
        }
        if (tid == 0) l_soff[Hp] = off;
        for (int k = 0; k < Hp; ++k) {
            const int a = V->hdv_off[k], b = V->hdv_off[k + 1];
            if (tid == 0) l_hoff[k] = off;
            stage16(l_soup + off, (const d2*)A.points + a, b - a, tid);
            off += (b - a);
        }
        if (tid == 0) l_hoff[Hp] = off;
        // lanelet soup last
        stage16(l_soup + off, (const d2*)A.points + V->ll_off, V->ll_len, tid);
        if (tid == 0) l_path[PDMPC_HP_MAX + 1] = (uint32_t)off;
    }
    __syncthreads();
    C.ll_base = uni_i((int)l_path[PDMPC_HP_MAX + 1]);
    C.ll_len = uni_i(V->ll_len);

    // ---- prologue 4: predecessors (PrioritizedController.m:476-491).  Their soup slots start as NaN (no obstacle).
    // Predecessors that have already finished are incorporated now; the others are "pending": the search starts
    // without them and arrival_sync() folds them in when they finish (speculation, see arrival_sync).
    SpecCtx P;
    P.sh = l_shared;
    P.l_soup = l_soup;
    P.l_soff = l_soff;
    P.l_lit = l_lit;
    P.out = A.out;
    P.pred = A.pred + V->pred_off;
    P.counters = A.tie_count;
    P.n_pred = n_pred;
    P.Hp = Hp;
    bool dep_timeout = false;
    // the arrival re-check (node_hits_areas) implements the InterX predicate; the convex/SAT checker (circle scenario,
    // a handful of vehicles) simply waits for its predecessors as the reference does
    const bool speculate = A.speculate && n_pred <= 64 && interx;
    if (n_pred > 0) {
        const d2 nanpt = d2{__longlong_as_double(0x7ff8000000000000LL), __longlong_as_double(0x7ff8000000000000LL)};
        for (int idx = tid; idx < Hp * pred_cols; idx += (int)blockDim.x) {
            const int k = idx / pred_cols;
            l_soup[l_soff[k] + l_lit[k] + (idx - k * pred_cols)] = nanpt;
        }
        if (!speculate) {
            // blocking wait (more than 64 predecessors, or speculation switched off)
            for (int p = 0; p < n_pred; ++p) {
                uint32_t spins = 0;
                while (__hip_atomic_load(A.done_flag + P.pred[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != A.epoch) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > A.spin_limit) {
                        dep_timeout = true;
                        break;
                    }
                }
            }
        }
        unsigned long long ready = 0;
        if (wave == 0) {
            bool d = false;
            if (lane < n_pred) d = __hip_atomic_load(A.done_flag + P.pred[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == A.epoch;
            ready = __ballot(d);
            if (!speculate) ready = (n_pred >= 64) ? ~0ull : ((1ull << n_pred) - 1ull);
            if (lane == 0) {
                const unsigned long long all = (n_pred >= 64) ? ~0ull : ((1ull << n_pred) - 1ull);
                const unsigned long long pend = speculate ? (all & ~ready) : 0ull;
                l_shared[SH_PEND_LO] = (uint32_t)pend;
                l_shared[SH_PEND_HI] = (uint32_t)(pend >> 32);
                l_shared[SH_ARR_LO] = (uint32_t)ready;
                l_shared[SH_ARR_HI] = (uint32_t)(ready >> 32);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (!speculate) {  // n_pred may exceed 64: incorporate everything directly
            const double qnan = __longlong_as_double(0x7ff8000000000000LL);
            for (int idx = tid; idx < Hp * pred_cols; idx += (int)blockDim.x) {
                const int k = idx / pred_cols;
                const int r = idx - k * pred_cols;
                const int p = r / PDMPC_VMAX;
                const int v = r - p * PDMPC_VMAX;
                const pdmpc_vehicle_out* PO = A.out + P.pred[p];
                const int cols = PO->shape_cols[k];
                d2 pt;
                pt.x = qnan;
                pt.y = qnan;
                if (v < cols) {
                    pt.x = PO->shapes[k][0][v];
                    pt.y = PO->shapes[k][1][v];
                }
                l_soup[l_soff[k] + l_lit[k] + r] = pt;
            }
        } else {
            incorporate_areas(P, sh_load64(l_shared, SH_ARR_LO), tid);
        }
        __syncthreads();
        if (tid == 0) {
            l_shared[SH_ARR_LO] = 0;
            l_shared[SH_ARR_HI] = 0;
        }
    }

    X.tid = tid;
    X.lane = lane;
    X.wave = wave;
    X.slot = slot;
    X.Hp = Hp;
    X.n = n;
    X.nw = nw;
    X.V = V;
    X.l_mask = l_mask;
    X.l_mi = l_mi;
    X.l_pose = l_pose;
    X.l_rx = l_rx;
    X.l_ry = l_ry;
    X.l_shared = l_shared;
    X.VS = VS;
    X.l_dcum = l_dcum;
    X.l_term = l_term;
    X.l_chxy = l_chxy;
    X.S = S;
    X.C = C;
    X.P = P;
    X.O = O;
    X.dep_timeout = dep_timeout;
#ifdef PDMPC_PROFILE
    X.rt_start = rt_start;
#endif
    X.Q.kring = (lds_f64*)(lsm + A.lds.heap_key);  // the block-min queue lives where the binary heap would
    X.Q.m1 = X.Q.kring + A.bm_kr;
    X.Q.pbits = (lds_u64*)(X.Q.m1 + A.bm_nb);
    X.Q.m2 = X.Q.m1 + 2 * A.bm_nb;
    X.Q.gkey = S.gkey;
    X.Q.kr_mask = (uint32_t)A.bm_kr - 1u;
    X.Q.nb_max = (uint32_t)A.bm_nb;
    X.Q.tie = false;
    X.l_path = l_path;
    X.lsm = lsm;
}

// Epilogue shared by the kernels (first wave only; the others have returned): the result record (GraphSearch.m:58-59, 82-89;
// return_path_to.m; return_path_area.m) and the publication of the done flag.  l_path[i] = node (1-based index into this
// vehicle's arena) of step i along the selected path; ref_ids (may be null: the same numbers) = the ids those nodes carry in the
// reference's tree, which is what info.tree_path reports.
__device__ __forceinline__ void search_epilogue(const KernelArgs& A, Ctx& X, const lds_u32* ref_ids) {
    const int lane = X.lane, Hp = X.Hp, slot = X.slot;
    const DevVehicle* __restrict__ V = X.V;
    pdmpc_vehicle_out* __restrict__ O = X.O;
    lds_u32* l_path = X.l_path;
    const CheckCtx& C = X.C;
    bool dep_timeout;
    Search S;
    const int status = X.status;
    const int n_popped = X.n_popped;
    const uint32_t goal = X.goal;
    const uint32_t nnodes = X.nnodes;
    dep_timeout = X.dep_timeout;
    S = X.S;

    // ---- results (GraphSearch.m:58-59, 82-89; return_path_to.m; return_path_area.m), sequencing wave only
    if (goal) {
        // path_to_root (Tree.m:44-52), reversed
        if (lane == 0 && !X.path_ready) {
            uint32_t nd = goal;
            for (int i = Hp; i >= 0; --i) {
                l_path[i] = nd;
                nd = node_parent(S, nd - 1);
            }
        }
        wave_sync();
        if (lane <= Hp) {
            const uint32_t nd = l_path[lane];
            const NodeRec r = node_load(S, nd - 1);
            O->tree_path[lane] = (int32_t)(ref_ids ? ref_ids[lane] : nd);
            double* row = O->path_nodes[lane];  // NodeInfo.m:5-13
            row[0] = r.x;
            row[1] = r.y;
            row[2] = r.yaw;
            row[3] = (double)NODE_TRIM(r.packed);
            row[4] = r.g;
            row[5] = r.h;
            row[6] = (double)NODE_K(r.packed);
            row[7] = 1.0;
            if (lane >= 1) {
                O->y_predicted[lane - 1][0] = r.x;
                O->y_predicted[lane - 1][1] = r.y;
                O->y_predicted[lane - 1][2] = r.yaw;
                O->predicted_trims[lane - 1] = (int32_t)NODE_TRIM(r.packed);
            }
        }
        // shapes along the path: same arithmetic as at pop time (GraphSearch.m:158-160), so the same bits
        for (int idx = lane; idx < Hp * PDMPC_VMAX; idx += PDMPC_WAVE) {
            const int i = idx / PDMPC_VMAX + 1;
            const int v = idx - (i - 1) * PDMPC_VMAX;
            const NodeRec pr = node_load(S, l_path[i - 1] - 1);
            const NodeRec cr = node_load(S, l_path[i] - 1);
            const int m = NODE_MAN(cr.packed);
            const int ncols = NODE_COLS(cr.packed);
            if (v == 0) O->shape_cols[i - 1] = ncols;
            if (v < ncols) {
                const d2 a = C.g_area[(size_t)m * 3 * PDMPC_VMAX + v];
                O->shapes[i - 1][0][v] = pr.cs * a.x - pr.sn * a.y + pr.x;
                O->shapes[i - 1][1][v] = pr.sn * a.x + pr.cs * a.y + pr.y;
            }
        }
    } else if (V->fb_off[0] >= 0) {
        // exhausted: publish the caller-supplied fallback areas so successors of this launch avoid them
        // (PrioritizedController.m:568-616, 678-718)
        for (int idx = lane; idx < Hp * PDMPC_VMAX; idx += PDMPC_WAVE) {
            const int k = idx / PDMPC_VMAX;
            const int v = idx - k * PDMPC_VMAX;
            const int a = V->fb_off[k], b = V->fb_off[k + 1];
            const int cols = (b - a < PDMPC_VMAX) ? (b - a) : PDMPC_VMAX;
            if (v == 0) O->shape_cols[k] = cols;
            if (v < cols) {
                O->shapes[k][0][v] = A.points[2 * (size_t)(a + v)];
                O->shapes[k][1][v] = A.points[2 * (size_t)(a + v) + 1];
            }
        }
    }
    if (lane == 0) {
        // (device-side tally of plans that are not planning results: bench.py reads it after its timed replay, whose launches it does not fetch)
        if (dep_timeout || (status != PDMPC_OK && status != PDMPC_EXHAUSTED)) atomicAdd(A.work_count + 6, 1ull);
        O->status = dep_timeout ? PDMPC_ERR_HIP : status;
        O->n_expanded = (int32_t)nnodes;
        O->n_popped = n_popped;
        O->n_hp = Hp;
        if (!ref_ids) A.tree_size[slot] = (int32_t)nnodes;  // (the frontier kernel has stored its raw node count)
    }
    // ---- publish: plain stores -> this wave's vmcnt(0) -> lane-0 agent release -> flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_sync();
    if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(A.done_flag + slot, A.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
