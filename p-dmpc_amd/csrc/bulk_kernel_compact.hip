// bulk_kernel_compact.hip — the InterX / one-mask-word search kernel built for TWO workgroups per CU: its fixed LDS regions are sized
// for 8 wavefronts and a ready list of 1 024 entries (pdmpc_device.h: PDMPC_LK_WAVES / PDMPC_LK_READY_CAP), the launch leaves the
// automaton's areas in L2, and the register budget is that of four wavefronts per SIMD — 2 x 8 wavefronts and 2 x 80 KB per CU.
// For launches of more searches than CUs (api.cpp: launch_range): a finished search that waits for its predecessors then holds half a
// CU, not a whole one, and twice as many searches are resident from the start (PrioritizedSequentialController.m:77-94 is the level
// loop those waits stand for).
#define PDMPC_LK_WAVES 8
#define PDMPC_LK_READY_CAP 1024u
#define PDMPC_BULK_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#include "bulk_search.hpp"

PDMPC_BULK_KERNEL(pdmpc_bulk_kernel_compact, pdmpc_launch_bulk_compact, 1, PDMPC_CHECK_INTERX, PDMPC_LK_COMPACT_WAVES)
