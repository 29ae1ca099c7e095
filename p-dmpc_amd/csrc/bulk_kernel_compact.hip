// bulk_kernel_compact.hip — the InterX / one-mask-word search kernel built for TWO workgroups per CU: its fixed LDS regions are sized
// for 8 wavefronts, a ready list of 512 entries and a near list of 1 024 (pdmpc_device.h: PDMPC_LK_WAVES / PDMPC_LK_READY_CAP / PDMPC_BK_PER), the launch leaves the
// automaton's areas in L2, and the register budget is that of four wavefronts per SIMD — 2 x 8 wavefronts and 2 x 80 KB per CU.
// For launches of more than two searches per CU (api.cpp: compute_lds_bulk; C5: the 64 prioritizations of a step in one launch,
// PrioritizedExplorativeController.m:25-91): light searches are bound by the latency of their own passes, and two of them side by side
// on a CU fill each other's gaps.
#define PDMPC_LK_WAVES 8
#define PDMPC_LK_READY_CAP 512u
#define PDMPC_BK_PER 2
#define PDMPC_BULK_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#include <hip/hip_runtime.h>
#ifdef PDMPC_COMPACT_FULL_BARRIER
// (experiment: every workgroup barrier also waits for this wave's global stores and loads)
__device__ __forceinline__ void pdmpc_full_barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (invalidates this CU's L1)
}
#define __syncthreads() pdmpc_full_barrier()
#endif
#include "bulk_search.hpp"

PDMPC_BULK_KERNEL(pdmpc_bulk_kernel_compact, pdmpc_launch_bulk_compact, 1, PDMPC_CHECK_INTERX, PDMPC_LK_COMPACT_WAVES)
