// debug_kernels.hip — the graph search's building blocks on their own: the libstdc++-faithful binary heap driven by a command script
// (pdmpc_debug_heap_script; heap_queue.hpp is what a search with equal keys ends on, bulk_search.hpp: bk_replay), and the collision
// primitives (pdmpc_debug_edge_check): one wavefront per case runs the same
// device functions the search kernels inline (edge_checks.hpp), so the reference's known-answer vectors
// (tests/unittests/hlc/intersect_unittest.m:8-54) and random polygon pairs can be checked on the device directly.
//   mode 0  InterX(a, b) with isReturnPoints = false            (graph_search/InterX.m:48-103; b may hold NaN separators)
//   mode 1  intersect_sat(a, b)                                  (graph_search/intersect_sat.m:1-42; b: one polygon or a 2-point segment)
//   mode 2  intersect_lanelet_boundary(a, [left, NaN, right, NaN]) (optimizer/common/intersect_lanelet_boundary.m:1-56)
#include <hip/hip_runtime.h>

#include "../../include/pdmpc_math.h"
#include "pdmpc_device.h"

#define PROF_MEMBERS
namespace {
#include "wave_primitives.hpp"
#include "search_state.hpp"
#include "heap_queue.hpp"
#include "edge_checks.hpp"
}  // namespace

#define DBG_MAX_B 1024

extern "C" __global__ __launch_bounds__(PDMPC_WAVE) void pdmpc_edge_check_kernel(int mode, int n_cases, const int32_t* a_off, const double* a_x, const double* a_y,
                                                                                const int32_t* b_off, const double* b_x, const double* b_y, int32_t* hit) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[(2 * PDMPC_VMAX + DBG_MAX_B + 2) * 16 + (DBG_MAX_B + 8) * 4];
    const int lane = threadIdx.x;
    const int c = blockIdx.x;
    if (c >= n_cases) return;
    LDS_AS unsigned char* lsm = (LDS_AS unsigned char*)smem;
    lds_d2* sh2 = (lds_d2*)lsm;                       // shape A in [0, VMAX), shape B (= A) in [VMAX, 2 VMAX)
    lds_d2* soup = sh2 + 2 * PDMPC_VMAX;              // the second operand
    const int a0 = a_off[c], na = a_off[c + 1] - a0, b0 = b_off[c], nb = b_off[c + 1] - b0;
    for (int i = lane; i < na; i += PDMPC_WAVE) {
        d2 p;
        p.x = a_x[a0 + i];
        p.y = a_y[a0 + i];
        sh2[i] = p;
        sh2[PDMPC_VMAX + i] = p;
    }
    for (int i = lane; i < nb; i += PDMPC_WAVE) {
        d2 p;
        p.x = b_x[b0 + i];
        p.y = b_y[b0 + i];
        soup[i] = p;
    }
    wave_sync();
    // the wave-wide forms (one lane per segment / per axis: the sampled optimizer's kernel) ...
    bool r = false;
    if (mode == 0)
        r = interx_check(sh2, na, soup, 0, nb, 0, 0, 0, 0, lane);
    else if (mode == 1)
        r = sat_pair_wave(sh2, na, soup, nb, lane);
    else
        r = sat_boundary_wave(sh2, na, soup, nb, lane);
    // ... and the per-lane forms of the graph search's check items (bulk_search.hpp, bk_check_items): a lane tests its pair alone
    bool r2 = false;
    {
        d2 pt[PDMPC_VMAX];
#pragma unroll
        for (int i = 0; i < PDMPC_VMAX; ++i) pt[i] = i < na ? (d2)sh2[i] : d2{0.0, 0.0};
        bool mine = false;
        if (mode == 0) {
            if (na >= 2)
                for (int j = lane; j + 1 < nb; j += PDMPC_WAVE) mine = mine || interx_segment_n<PDMPC_VMAX>(pt, na - 1, soup[j], soup[j + 1]);
        } else if (mode == 1) {
            if (lane == 0) mine = sat_pair_lane(pt, na, soup, nb);
        } else {
            double min_x, max_x, min_y, max_y;
            sat_area_bbox(pt, na, min_x, max_x, min_y, max_y);
            for (int j = lane; j + 1 < nb; j += PDMPC_WAVE) mine = mine || sat_boundary_segment_lane(pt, na, min_x, max_x, min_y, max_y, soup[j], soup[j + 1]);
        }
        r2 = wave_any(mine);
    }
    if (lane == 0 && r != r2) {  // (the two forms evaluate the same expressions: must never happen; reported as 2 + the per-lane answer)
        hit[c] = 2 + (r2 ? 1 : 0);
        return;
    }
    if (lane == 0) hit[c] = r ? 1 : 0;
}

extern "C" int pdmpc_launch_edge_check(int mode, int n_cases, const int32_t* a_off, const double* a_x, const double* a_y, const int32_t* b_off, const double* b_x,
                                       const double* b_y, int32_t* hit, void* stream) {
    if (n_cases <= 0) return 0;
    hipLaunchKernelGGL(pdmpc_edge_check_kernel, dim3(n_cases), dim3(PDMPC_WAVE), 0, (hipStream_t)stream, mode, n_cases, a_off, a_x, a_y, b_off, b_x, b_y, hit);
    return (int)hipGetLastError();
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


extern "C" int pdmpc_launch_heap_script(const int32_t* op, const int32_t* id, const double* key, int n, int32_t* out, unsigned long long* stats,
                                        double* gkey, uint32_t* gid, int HL, void* stream) {
    const size_t lds = (size_t)HL * 12 + 16;
    hipError_t e = hipFuncSetAttribute((const void*)pdmpc_heap_script_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pdmpc_heap_script_kernel, dim3(1), dim3(PDMPC_WAVE), lds, (hipStream_t)stream, op, id, key, n, out, stats, gkey, gid, HL);
    return (int)hipGetLastError();
}

