// search_kernel.hip — the optimal graph search: one workgroup of 8 or 16 wavefronts plans one vehicle.
//
// The kernel restates, for gfx950, the reference's optimal graph search:
//   GraphSearch.do_graph_search      hlc/optimizer/graph_search/GraphSearch.m:23-107
//   GraphSearch.eval_edge_exact      GraphSearch.m:111-196
//   expand_node                      graph_search/expand_node.m:1-91
//   are_constraints_satisfied_sat    graph_search/are_constraints_satisfied_sat.m:1-68  (+ intersect_sat.m,
//                                    common/intersect_lanelet_boundary.m)                       -> edge_checks.hpp
//   are_constraints_satisfied_interx graph_search/are_constraints_satisfied_interx.m:1-39 (+ InterX.m:48-110)
//   std::priority_queue semantics    priority_queue/priority_queue_interface_mex.cpp:19-31      -> heap_queue.hpp (exact for any
//                                    keys), blockmin_queue.hpp (exact while the minimal key is unique; certifies it)
//   return_path_to / return_path_area / Tree.path_to_root
//   the hand-over of solved areas between vehicles of one time step (PrioritizedController.m:476-491) -> arrival_sync
//
// Mapping to the hardware (DESIGN.md section 3 has the full picture):
//   * The search is a dependent chain (pop -> check -> expand -> push) and one wavefront issues about one instruction
//     per 9.5 cycles, so the chain is bound by its instruction count.  The work is therefore split over wave roles:
//     wave 0 owns the open list and the pop order, wave 1 evaluates and expands the popped nodes, wave 2 looks for the
//     nodes that will be popped next, the remaining waves evaluate their edges ahead of time — the urgent ones first,
//     then every node of the tree in creation order.  Entries whose edge is known to collide leave the open list on the
//     side when a pop looks at their block instead of being popped and discarded one by one (GraphSearch.m:75-77); the
//     pop count the reference would report is reconstructed exactly at the end.  Waves talk through a few LDS words
//     (8-byte mail boxes with sequence numbers; LDS keeps a wave's accesses in program order).
//   * MPA tables, the vehicle's obstacle "soup", the open list's index (block minima, popped bits, recent keys), the
//     validity bytes and the first tree nodes live in LDS; every node is also written to HBM as one 64-byte record (the
//     children of an expansion are one contiguous coalesced store), every key as 8 bytes.
//   * Every LDS pointer carries address_space(3) in its type: a generic pointer would compile to flat_* accesses, which on
//     gfx9 make the wave wait for all its outstanding HBM stores.
//   * Edge check (InterX): pass 1 evaluates C2 (which obstacle-segment lines cut the vehicle's area) one lane per
//     obstacle segment and compacts the few survivors; pass 2 evaluates C1 only for those.  Skipping C1 where C2 is
//     false changes no result: the reference tests C1 & C2 (InterX.m:72-76).
//   * Floating point: every evaluated expression keeps the reference's operation order; compiled with
//     -ffp-contract=off (no FMA), IEEE sqrt/div, sin/cos from include/pdmpc_math.h: bit-identical to the oracle.
#include <hip/hip_runtime.h>

#include "serial_search.hpp"


// The kernel body, specialised at compile time on the constraint checker so each variant carries only its own
// collision code (the search is instruction-cache and issue bound: smaller is faster).
template <int CHECKER, int NW>
__device__ __forceinline__ void search_body(const KernelArgs& A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Ctx X;
    search_prologue(A, X, (LDS_AS unsigned char*)smem, CHECKER == PDMPC_CHECK_INTERX);
    const int tid = X.tid, lane = X.lane, wave = X.wave;
    volatile lds_u32* l_shared = X.l_shared;
    bool tie = false;
    if (A.queue_mode == PDMPC_QUEUE_BLOCKMIN) {
        tie = search_loops<CHECKER, true, NW>(A, X);
        if (tie) {  // (uniform over the workgroup) start over with the exact open list; areas that arrived so far stay in the soup
            __syncthreads();
            if (tid == 0) {
                l_shared[SH_STATE] = ST_RUN;
                l_shared[SH_ARR_LO] = 0;
                l_shared[SH_ARR_HI] = 0;
                l_shared[SH_RESTART] = 0;
                atomicAdd(A.tie_count, 1);
            }
            __syncthreads();
        }
    }
    if (A.queue_mode != PDMPC_QUEUE_BLOCKMIN || tie) (void)search_loops<CHECKER, false, NW>(A, X);
    if (lane == 0) {
        atomicAdd(A.work_count + 0, X.C.tally[0]);
        atomicAdd(A.work_count + 1, X.C.tally[1]);
    }
    __syncthreads();
    if (wave != 0) return;
    search_epilogue(A, X, nullptr);
}

// Debug/unit-test kernel: drives the device open list with a command script (op 0: push (id, key), op 1: pop) exactly
// like oracle_pq_script drives the reference's std::priority_queue; out receives the popped ids (-1 on empty).
// stats[0] = shader cycles spent in pops, stats[1] = number of pops, stats[2] = cycles in pushes, stats[3] = pushes.
extern "C" __global__ __launch_bounds__(PDMPC_WAVE) void pdmpc_heap_script_kernel(const int32_t* op, const int32_t* id, const double* key, int n,
                                                                                 int32_t* out, unsigned long long* stats, double* gkey, uint32_t* gid, int HL) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Search S;
    S.lkey = (lds_f64*)smem;
    S.lid = (lds_u32*)(smem + (size_t)HL * 8);
    S.gkey = gkey;
    S.gid = gid;
    S.HL = (uint32_t)HL;
    S.heap_len = 0;
    S.lane = threadIdx.x;
    S.pl = make_pop_lane(S.lane);
    S.ln = nullptr;
    S.gn = nullptr;
    S.NL = 0;
    S.max_nodes = 0;
    int n_out = 0;
    unsigned long long c_pop = 0, c_push = 0, n_pop = 0, n_push = 0;
    for (int i = 0; i < n; ++i) {
        const int o = uni_i(op[i]);
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (o == 0) {
            heap_push(S, (uint32_t)uni_i(id[i]), uni_d(key[i]));
            c_push += __builtin_readcyclecounter() - t0;
            ++n_push;
        } else {
            int32_t r = -1;
            if (S.heap_len > 0) {
                double k0;
                uint32_t i0;
                heap_load<false>(S, 0, true, k0, i0);
                r = (int32_t)uni_u(i0);
                heap_pop(S);
            }
            c_pop += __builtin_readcyclecounter() - t0;
            ++n_pop;
            if (S.lane == 0) out[n_out] = r;
            ++n_out;
        }
    }
    if (S.lane == 0) {
        stats[0] = c_pop;
        stats[1] = n_pop;
        stats[2] = c_push;
        stats[3] = n_push;
    }
}

// Debug/unit-test kernel for the block-min queue: op 0 pushes the next node (ids 1, 2, 3, ... in script order) with key[i],
// op 1 pops.  Consecutive pushes (up to 16) go in as one batch, like the children of an expansion.  out receives the
// popped ids (-1 on empty); stats[0..3] as in the heap script kernel, stats[4] = 1 if a pop saw a tied minimum.
extern "C" __global__ __launch_bounds__(PDMPC_WAVE) void pdmpc_bm_script_kernel(const int32_t* op, const double* key, int n, int32_t* out,
                                                                               unsigned long long* stats, double* gkey, int KR, int NB) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    BmQueue Q;
    Q.kring = (lds_f64*)smem;
    Q.m1 = (lds_f64*)(smem + (size_t)KR * 8);
    Q.pbits = (lds_u64*)(Q.m1 + NB);
    Q.m2 = Q.m1 + 2 * NB;
    Q.gkey = gkey;
    Q.kr_mask = (uint32_t)KR - 1u;
    Q.nb_max = (uint32_t)NB;
    Q.tie = false;
    bm_init(Q, lane, PDMPC_WAVE);
    __syncthreads();
    uint32_t nn = 0;
    int n_out = 0;
    unsigned long long c_pop = 0, c_push = 0, n_pop = 0, n_push = 0;
    int i = 0;
    // the clock is read with nothing of the script's own memory traffic in flight (loads of op/key, the store to out)
#define BM_CLOCK(t, w)                          \
    __builtin_amdgcn_s_waitcnt(w);              \
    __builtin_amdgcn_sched_barrier(0);          \
    const unsigned long long t = __builtin_readcyclecounter(); \
    __builtin_amdgcn_sched_barrier(0);
    while (i < n) {
        const int o = uni_i(op[i]);
        if (o == 0) {
            int cnt = 1;
            while (cnt < 16 && i + cnt < n && uni_i(op[i + cnt]) == 0) ++cnt;
            const bool active = lane < cnt;
            const double f = active ? key[i + lane] : 0.0;
            BM_CLOCK(t0, 0)
            bm_push<true>(Q, active, nn + (uint32_t)lane, f, nn, nn + (uint32_t)cnt);
            BM_CLOCK(t1, 0xC07F)  // lgkmcnt(0): the queue's HBM stores are fire-and-forget
            nn += (uint32_t)cnt;
            c_push += t1 - t0;
            n_push += (unsigned long long)cnt;
            i += cnt;
        } else {
            // a run of consecutive pops is timed as a whole (no clock reads in between)
            int run = 1;
            while (i + run < n && uni_i(op[i + run]) == 1) ++run;
            BM_CLOCK(t0, 0)
            for (int r = 0; r < run; ++r) {
                const uint32_t cur = bm_pop(Q, nn).idx;
                if (lane == 0) out[n_out] = cur == 0xFFFFFFFFu ? -1 : (int32_t)(cur + 1u);
                ++n_out;
            }
            BM_CLOCK(t1, 0xC07F)  // lgkmcnt(0): the queue's HBM stores are fire-and-forget
            c_pop += t1 - t0;
            n_pop += (unsigned long long)run;
            i += run;
        }
    }
    if (lane == 0) {
        stats[0] = c_pop;
        stats[1] = n_pop;
        stats[2] = c_push;
        stats[3] = n_push;
        stats[4] = Q.tie ? 1ull : 0ull;
    }
}

extern "C" int pdmpc_launch_bm_script(const int32_t* op, const double* key, int n, int32_t* out, unsigned long long* stats, double* gkey, int KR,
                                      int NB, void* stream) {
    const size_t lds = (size_t)KR * 8 + (size_t)NB * 16 + 64 * 8;
    hipError_t e = hipFuncSetAttribute((const void*)pdmpc_bm_script_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pdmpc_bm_script_kernel, dim3(1), dim3(PDMPC_WAVE), lds, (hipStream_t)stream, op, key, n, out, stats, gkey, KR, NB);
    return (int)hipGetLastError();
}

extern "C" int pdmpc_launch_heap_script(const int32_t* op, const int32_t* id, const double* key, int n, int32_t* out, unsigned long long* stats,
                                        double* gkey, uint32_t* gid, int HL, void* stream) {
    const size_t lds = (size_t)HL * 12 + 16;
    hipError_t e = hipFuncSetAttribute((const void*)pdmpc_heap_script_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pdmpc_heap_script_kernel, dim3(1), dim3(PDMPC_WAVE), lds, (hipStream_t)stream, op, id, key, n, out, stats, gkey, gid, HL);
    return (int)hipGetLastError();
}

// one successor-mask word (MPAs of at most 64 trims: all of the reference's but the "realistic" one) / any number
extern "C" __global__ __launch_bounds__(PDMPC_MAX_THREADS) void pdmpc_search_kernel(const KernelArgs A) { search_body<PDMPC_CHECK_INTERX, 1>(A); }
extern "C" __global__ __launch_bounds__(PDMPC_MAX_THREADS) void pdmpc_search_kernel_sat(const KernelArgs A) { search_body<PDMPC_CHECK_SAT, 1>(A); }
extern "C" __global__ __launch_bounds__(PDMPC_MAX_THREADS) void pdmpc_search_kernel_wide(const KernelArgs A) { search_body<PDMPC_CHECK_INTERX, 0>(A); }
extern "C" __global__ __launch_bounds__(PDMPC_MAX_THREADS) void pdmpc_search_kernel_sat_wide(const KernelArgs A) { search_body<PDMPC_CHECK_SAT, 0>(A); }

// The same kernels compiled for six wavefronts per SIMD (at most 80 VGPRs instead of 90): two workgroups of twelve
// wavefronts fit on a CU.  For launches with more workgroups than CUs (C4: 26.3 steps/s against 23.8 with the regular
// build at ten wavefronts; C5: 216.7 against 185 at eight).  With a CU to itself a workgroup is 2 % faster in the regular
// build.
#define PDMPC_DENSE __attribute__((amdgpu_waves_per_eu(6, 6)))
extern "C" __global__ __launch_bounds__(PDMPC_MAX_THREADS) PDMPC_DENSE void pdmpc_search_kernel_dense(const KernelArgs A) { search_body<PDMPC_CHECK_INTERX, 1>(A); }
extern "C" __global__ __launch_bounds__(PDMPC_MAX_THREADS) PDMPC_DENSE void pdmpc_search_kernel_sat_dense(const KernelArgs A) { search_body<PDMPC_CHECK_SAT, 1>(A); }
extern "C" __global__ __launch_bounds__(PDMPC_MAX_THREADS) PDMPC_DENSE void pdmpc_search_kernel_wide_dense(const KernelArgs A) { search_body<PDMPC_CHECK_INTERX, 0>(A); }
extern "C" __global__ __launch_bounds__(PDMPC_MAX_THREADS) PDMPC_DENSE void pdmpc_search_kernel_sat_wide_dense(const KernelArgs A) { search_body<PDMPC_CHECK_SAT, 0>(A); }

extern "C" int pdmpc_launch_search(const KernelArgs* args, int count, void* stream) {
    if (count <= 0) return 0;
    typedef void (*kernel_t)(const KernelArgs);
    const bool interx = args->checker == PDMPC_CHECK_INTERX, one_word = args->n_words == 1;
    kernel_t fn = interx ? (one_word ? pdmpc_search_kernel : pdmpc_search_kernel_wide) : (one_word ? pdmpc_search_kernel_sat : pdmpc_search_kernel_sat_wide);
    if (args->dense) fn = interx ? (one_word ? pdmpc_search_kernel_dense : pdmpc_search_kernel_wide_dense) : (one_word ? pdmpc_search_kernel_sat_dense : pdmpc_search_kernel_sat_wide_dense);
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)args->lds.total);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fn, dim3(count), dim3(PDMPC_WAVE * args->n_waves), args->lds.total, (hipStream_t)stream, *args);
    return (int)hipGetLastError();
}
