// debug_kernels.hip — the collision primitives on their own (pdmpc_debug_edge_check): one wavefront per case runs the same
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
