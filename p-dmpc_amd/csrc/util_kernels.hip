// util_kernels.hip — small device-side helpers of the host API (no search logic).
//
// pdmpc_gather_lean_kernel: status and cost-to-come of the final node (path_nodes[Hp][4]) of every result record into one compact
// array of (double, int32, pad) — what the explorative step's choice among the prioritizations reads of ALL plans
// (PrioritizedExplorativeController.m:94-112) — so that one contiguous 16-byte-per-plan copy replaces the read-back of the 2.9 KB
// records (a strided hipMemcpy2D of 8-byte rows costs a DMA descriptor per row: 1 280 rows took 1.3 ms).
#include <hip/hip_runtime.h>

#include "pdmpc_device.h"

extern "C" __global__ void pdmpc_gather_lean_kernel(const pdmpc_vehicle_out* __restrict__ out, int n, int Hp, double* __restrict__ lean) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    lean[2 * (size_t)i] = out[i].path_nodes[Hp][4];
    int32_t* st = (int32_t*)(lean + 2 * (size_t)i + 1);
    st[0] = out[i].status;
    st[1] = 0;
}

extern "C" int pdmpc_launch_gather_lean(const pdmpc_vehicle_out* out, int n, int Hp, double* lean, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pdmpc_gather_lean_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, n, Hp, lean);
    return (int)hipGetLastError();
}
