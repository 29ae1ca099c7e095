// Microbenchmark / platform check: do two workgroups that each take half the CU's LDS (80 KB of 160 KB) keep their LDS to themselves
// when they share a CU?  Every workgroup fills its dynamic LDS with a pattern of its own, keeps rewriting and verifying it for a few
// milliseconds (so that the workgroups of a launch of 2 x CUs workgroups are co-resident), and counts the words that did not read back.
// Also reports how many workgroups saw a co-resident partner (same CU id within the same time window).
// build: hipcc --offload-arch=gfx950 -O2 -o ubench_lds_coresidency tools/ubench_lds_coresidency.hip ; run: ./ubench_lds_coresidency [lds_bytes] [threads]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k(unsigned* bad, unsigned* cu_of, unsigned long long* t0t1, int words, int iters) {
    extern __shared__ unsigned lds[];
    const unsigned tag = 0x9e3779b9u * (blockIdx.x + 1u);
    unsigned hw = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) {
        cu_of[blockIdx.x] = hw;
        t0t1[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    }
    unsigned errs = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = tag ^ (unsigned)(i * 2654435761u) ^ (unsigned)it;
        __syncthreads();
        for (int rep = 0; rep < 4; ++rep) {
            for (int i = threadIdx.x; i < words; i += blockDim.x) {
                const int j = (i * 37 + rep * 101) % words;  // (read words other threads wrote)
                if (lds[j] != (tag ^ (unsigned)(j * 2654435761u) ^ (unsigned)it)) ++errs;
            }
            __builtin_amdgcn_s_sleep(20);
        }
        __syncthreads();
    }
    if (errs) atomicAdd(bad, errs);
    if (threadIdx.x == 0) t0t1[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
}

int main(int argc, char** argv) {
    const int lds_bytes = argc > 1 ? atoi(argv[1]) : 81664, threads = argc > 2 ? atoi(argv[2]) : 512, blocks = argc > 3 ? atoi(argv[3]) : 512;
    unsigned *bad, *cu;
    unsigned long long* tt;
    hipMalloc(&bad, 4);
    hipMalloc(&cu, 4 * blocks);
    hipMalloc(&tt, 16 * blocks);
    hipMemset(bad, 0, 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds_bytes, 0, bad, cu, tt, lds_bytes / 4, 60);
    hipError_t e = hipDeviceSynchronize();
    unsigned hb = 0;
    std::vector<unsigned> hcu(blocks);
    std::vector<unsigned long long> htt(2 * blocks);
    hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    hipMemcpy(hcu.data(), cu, 4 * blocks, hipMemcpyDeviceToHost);
    hipMemcpy(htt.data(), tt, 16 * blocks, hipMemcpyDeviceToHost);
    int paired = 0;
    for (int a = 0; a < blocks; ++a)
        for (int b = 0; b < blocks; ++b)
            if (a != b && ((hcu[a] ^ hcu[b]) & 0x0ff00f00u) == 0 /* same SE / SH? / CU id bits: see below */) {
                if (htt[2 * a] < htt[2 * b + 1] && htt[2 * b] < htt[2 * a + 1]) {
                    ++paired;
                    break;
                }
            }
    printf("lds %d B x %d threads x %d workgroups: status %s, words that did not read back: %u, workgroups that overlapped in time with another one on the same hw id bits: %d\n", lds_bytes, threads,
           blocks, hipGetErrorString(e), hb, paired);
    printf("sample HW_ID: %08x %08x %08x %08x\n", hcu[0], hcu[1], hcu[256 % blocks], hcu[257 % blocks]);
    return hb != 0;
}
