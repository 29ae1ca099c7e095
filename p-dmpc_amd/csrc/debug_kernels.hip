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
    bool r = false;
    if (mode == 0)
        r = interx_check(sh2, na, soup, 0, nb, 0, 0, 0, 0, lane);
    else if (mode == 1)
        r = sat_pair_wave(sh2, na, soup, nb, lane);
    else
        r = sat_boundary_wave(sh2, na, soup, nb, lane);
    if (lane == 0) hit[c] = r ? 1 : 0;
}

extern "C" int pdmpc_launch_edge_check(int mode, int n_cases, const int32_t* a_off, const double* a_x, const double* a_y, const int32_t* b_off, const double* b_x,
                                       const double* b_y, int32_t* hit, void* stream) {
    if (n_cases <= 0) return 0;
    hipLaunchKernelGGL(pdmpc_edge_check_kernel, dim3(n_cases), dim3(PDMPC_WAVE), 0, (hipStream_t)stream, mode, n_cases, a_off, a_x, a_y, b_off, b_x, b_y, hit);
    return (int)hipGetLastError();
}
