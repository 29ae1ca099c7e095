// search_common.hpp — what the graph search's workgroups share besides the search itself (device code, included by
// bulk_search.hpp): the validity bytes, the shared words of the predecessor bookkeeping, the copy of a predecessor's solved areas
// into the obstacle soup, the search context, and the prologue — LDS carve, MPA tables, vehicle record, obstacle soups
// ([literal polygons + NaN][predecessors' areas padded to VMAX] per step, vectorize_all_obstacles.m:36-62), result record defaults,
// predecessors that have finished already (PrioritizedController.m:476-491).
#pragma once
#include "../../include/pdmpc_math.h"
#include "pdmpc_device.h"

#define PROF_MEMBERS

namespace {

#include "wave_primitives.hpp"
#include "search_state.hpp"
#include "heap_queue.hpp"
#include "edge_checks.hpp"

#define VS_UNKNOWN 0u
#define VS_VALID 1u
#define VS_INVALID 2u

// validity of a node's edge: 0 never evaluated, 1 collision-free, 2 colliding (6: parked, bulk_search.hpp); the first NV nodes in LDS, the rest in HBM (same CU -> same L1)
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

}  // namespace

// LDS words shared between the waves of a workgroup (in the `path` region, after the offset tables); words 1, 2, 4 and 9-25 belong to
// the search (bulk_search.hpp), words 26 and up to its rounds (frontier_common.hpp)
#define SH_STATE 0     // ST_RUN searching, ST_ARRIVED predecessors have finished (SH_ARR says who)
#define SH_NNODES 3    // tree size
#define SH_PEND_LO 5   // predecessors whose areas are not in the soup yet (bit p = p-th predecessor)
#define SH_PEND_HI 6
#define SH_ARR_LO 7    // predecessors that just finished (to be copied into the soup)
#define SH_ARR_HI 8
#define SH_WORDS PDMPC_SH_WORDS
#define ST_RUN 0u
#define ST_ARRIVED 1u

namespace {

__device__ __forceinline__ unsigned long long sh_load64(volatile lds_u32* sh, int lo) {
    return (unsigned long long)sh[lo] | ((unsigned long long)sh[lo + 1] << 32);
}

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
__device__ __forceinline__ void incorporate_areas(const SpecCtx& P, unsigned long long arr, int tid) {
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

}  // namespace

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
};

struct ExpandEnv {
    lds_mask64* l_mask;
    lds_i16* l_mi;
    lds_pose* l_pose;
    lds_f64 *l_rx, *l_ry, *l_dcum, *l_term;
    lds_d2* l_chxy;
    int Hp, n, nw, lane;
};

// Prologue shared by the kernels: carves the LDS allocation (offsets from KernelArgs::lds), stages the MPA tables, the vehicle
// record and its obstacle soups, initialises the result record and the predecessor bookkeeping, and fills the context X.
__device__ __forceinline__ void search_prologue(const KernelArgs& A, Ctx& X, LDS_AS unsigned char* lsm) {
    const int tid = threadIdx.x;
    const int lane = tid & (PDMPC_WAVE - 1);
    const int wave = uni_i(tid >> 6);
    const int blk = (int)blockIdx.x - A.bk_helpers_first;  // (helper workgroups in front: bulk_body)
    const int slot = A.first + (A.reverse_dispatch ? A.n_searches - 1 - blk : blk);
    const int Hp = A.Hp;
    const int n = A.n_trims;
    const int nw = A.n_words;
    const DevVehicle* __restrict__ V = A.veh + slot;

    // ---- LDS carve
    lds_mask64* l_mask = (lds_mask64*)(lsm + A.lds.mask);
    lds_i16* l_mi = (lds_i16*)(lsm + A.lds.man_index);
    lds_pose* l_pose = (lds_pose*)(lsm + A.lds.pose);
    lds_f64* l_rx = (lds_f64*)(lsm + PDMPC_LK_REF);
    lds_f64* l_ry = l_rx + PDMPC_HP_MAX;
    lds_f64* l_dtv = l_ry + PDMPC_HP_MAX;
    lds_u32* l_path = (lds_u32*)(lsm + PDMPC_LK_PATH);
    lds_i32* l_soff = (lds_i32*)(l_path + PDMPC_HP_MAX + 2);  // soup offsets [Hp+1], hdv offsets [Hp+1]
    lds_i32* l_hoff = l_soff + PDMPC_HP_MAX + 1;
    volatile lds_u32* l_shared = (volatile lds_u32*)(l_hoff + PDMPC_HP_MAX + 1);
    lds_i32* l_lit = (lds_i32*)(l_shared + SH_WORDS);  // literal soup length per step
    lds_d2* l_soup = (lds_d2*)(lsm + A.lds.soup);
    VState VS;
    VS.l = (volatile lds_u8*)(lsm + A.lds.vstate);
    VS.NV = (uint32_t)A.NV;
    lds_f64* l_dcum = (lds_f64*)(lsm + PDMPC_LK_EXPAND);             // [HP_MAX][HP_MAX] cumulative dt*v_ref per (k_exp, t)
    lds_f64* l_term = l_dcum + PDMPC_HP_MAX * PDMPC_HP_MAX;       // [16 children][HP_MAX] cost-to-go terms
    lds_d2* l_chxy = (lds_d2*)(l_term + 16 * PDMPC_HP_MAX);       // [16] child positions

    Search S;
    S.ln = (lds_d2*)(lsm + A.lds.nodes);
    S.lkey = nullptr;  // (the binary heap of a replay lives where the open set's LDS part was: bulk_search.hpp, bk_replay)
    S.lid = nullptr;
    S.NL = (uint32_t)A.NL;
    S.HL = 0;
    S.max_nodes = A.max_nodes;
    S.lane = lane;
    S.pl = make_pop_lane(lane);
    S.heap_len = 0;
    const size_t voff = (size_t)slot * A.max_nodes;
    S.gn = A.arena.nodes + voff;
    S.gkey = A.arena.key + voff;
    S.gid = A.arena.far_id + voff;
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
    C.sh = (lds_d2*)(lsm + PDMPC_LK_SHAPE) + wave * (2 * PDMPC_VMAX + 1);
    C.tally = (LDS_AS unsigned long long*)(C.sh + 2 * PDMPC_VMAX);  // [0] edge checks, [1] segment pairs (this wave)
    if (lane == 0) {
        C.tally[0] = 0;
        C.tally[1] = 0;
    }
    C.cand = (lds_u32*)(lsm + PDMPC_LK_CAND) + (size_t)wave * A.cand_cap;

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
    // without them and folds them in when they finish (bulk_search.hpp: the copy at a round boundary, then the verification).
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
    const bool speculate = A.speculate && n_pred <= 64;
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
    X.l_path = l_path;
    X.lsm = lsm;
}
