// Microbenchmark: instruction-level parallelism inside one wavefront on gfx950 (independent VALU chains, two interleaved DPP reductions).
// hipcc --offload-arch=gfx950 -O3 -o ubench2 tools/ubench_ilp.hip && ./ubench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(unsigned long long* out, uint32_t* sink, int n) {
    const int lane = threadIdx.x;
    uint32_t a = lane, b = lane * 3, c = lane * 5, d = lane * 7, e = lane * 11, f = lane * 13, g = lane * 17, h = lane * 19, p = lane + 1;
    unsigned long long t0, t1;
    // 1: 16 dependent v_add
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(p));
    }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[0] = t1 - t0;
    // 2: 16 v_add as 2 independent chains interleaved
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(p));
    }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[1] = t1 - t0;
    // 3: 16 v_add as 4 independent chains
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(p));
    }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[2] = t1 - t0;
    // 4: 16 v_add as 8 independent chains
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            asm volatile("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(p));
    }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[3] = t1 - t0;
    // 5: two DPP reductions interleaved (independent), no nops
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        asm volatile(
            "s_nop 1\n\t"
            "v_min_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t v_min_u32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t s_nop 0\n\t"
            "v_min_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t v_min_u32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t s_nop 0\n\t"
            "v_min_u32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t v_min_u32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t s_nop 0\n\t"
            "v_min_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t v_min_u32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\t s_nop 0\n\t"
            "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t v_min_u32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t s_nop 0\n\t"
            "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t v_min_u32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t s_nop 1"
            : "+v"(a), "+v"(b));
        a = (uint32_t)__builtin_amdgcn_readlane((int)a, 63) + lane + i;
        b = (uint32_t)__builtin_amdgcn_readlane((int)b, 63) ^ (lane * 3 + i);
    }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[4] = t1 - t0;
    // 6: 8 independent s_add
    uint32_t s1 = __builtin_amdgcn_readfirstlane(a), s2 = s1 + 1;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("s_add_u32 %0, %0, 3\n\ts_add_u32 %1, %1, 5" : "+s"(s1), "+s"(s2));
    }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[5] = t1 - t0;
    // 7: interleave valu + salu (independent)
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, 3" : "+v"(a), "+s"(s1) : "v"(p));
    }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[6] = t1 - t0;
    sink[lane] = a + b + c + d + e + f + g + h + s1 + s2;
}
int main() {
    unsigned long long* out; uint32_t* sink;
    hipMalloc(&out, 16 * 8); hipMalloc(&sink, 64 * 4);
    const int n = 2000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, sink, n); hipDeviceSynchronize(); }
    unsigned long long h[16];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = {"v_add dependent (per instr)", "v_add 2 chains (per instr)", "v_add 4 chains (per instr)", "v_add 8 chains (per instr)", "two interleaved DPP reductions + readlanes (per pair)", "s_add 2 chains (per instr)", "v_add + s_add interleaved (per instr)"};
    const double div[] = {16, 16, 16, 16, 1, 16, 16};
    for (int i = 0; i < 7; ++i) printf("%-55s %8.2f cycles\n", names[i], (double)h[i] / n / div[i]);
    return 0;
}
