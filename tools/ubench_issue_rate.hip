// Microbenchmark: what one wavefront alone on a gfx950 SIMD pays per dependent instruction, LDS round trip, DPP reduction.
// hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench_issue_rate.hip && ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define LDS_AS __attribute__((address_space(3)))
__device__ __forceinline__ uint32_t red_nop(uint32_t v) {
    asm volatile(
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t s_nop 1"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ uint32_t red_rowonly(uint32_t v) {
    asm volatile(
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t s_nop 1"
        : "+v"(v));
    return v;
}
__global__ void k(unsigned long long* out, uint32_t* sink, int n) {
    extern __shared__ unsigned char smem[];
    LDS_AS uint32_t* l = (LDS_AS uint32_t*)smem;
    const int lane = threadIdx.x;
    l[lane] = (lane * 7 + 3) & 63;
    __syncthreads();
    uint32_t v = lane * 2654435761u;
    unsigned long long t0, t1;
    // 1: full reduce + readlane, dependent
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) { v = red_nop(v ^ (uint32_t)i); v = (uint32_t)__builtin_amdgcn_readlane((int)v, 63) + lane; }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[0] = t1 - t0;
    // 2: row-only reduce + readlane
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) { v = red_rowonly(v ^ (uint32_t)i); v = (uint32_t)__builtin_amdgcn_readlane((int)v, 15) + lane; }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[1] = t1 - t0;
    // 3: LDS pointer chase
    uint32_t p = lane;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) { p = l[p & 63]; }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[2] = t1 - t0;
    // 4: ballot + ctz + readlane chain
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) { unsigned long long b = __ballot((v & 63u) == (uint32_t)lane || lane == 63); int q = __builtin_ctzll(b); v = (uint32_t)__builtin_amdgcn_readlane((int)v, q) + lane + i; }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[3] = t1 - t0;
    // 5: dependent v_add chain (16 adds per iter)
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(p));
    }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[4] = t1 - t0;
    // 6: dependent f64 min chain (16 per iter)
    double d = (double)v;
    double e = (double)p;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d) : "v"(e));
    }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[5] = t1 - t0;
    // 7: s_nop 1 x16
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("s_nop 1");
    }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[6] = t1 - t0;
    // 8: LDS atomic min to one address + read back
    LDS_AS unsigned long long* a = (LDS_AS unsigned long long*)(smem + 1024);
    if (lane == 0) *a = ~0ull;
    __syncthreads();
    unsigned long long r = 0;
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        __hip_atomic_fetch_min(a, (unsigned long long)(v + lane + r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        r = *(volatile LDS_AS unsigned long long*)a;
    }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[7] = t1 - t0;
    // 9: scalar chain s_add x16
    uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s));
    }
    t1 = __builtin_readcyclecounter();
    if (lane == 0) out[8] = t1 - t0;
    sink[lane] = v + p + (uint32_t)d + (uint32_t)r + s;
}
int main() {
    unsigned long long* out; uint32_t* sink;
    hipMalloc(&out, 16 * 8); hipMalloc(&sink, 64 * 4);
    const int n = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, out, sink, n);
        hipDeviceSynchronize();
    }
    unsigned long long h[16];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = {"full u32 reduce (6 dpp + nops) + readlane", "row reduce (4 dpp) + readlane", "LDS dependent read", "ballot+ctz+readlane", "v_add_u32 (per instr)", "v_min_f64 (per instr)", "s_nop 1 (per instr)", "LDS atomic min same addr + read", "s_add_u32 (per instr)"};
    const double div[] = {1, 1, 1, 1, 16, 16, 16, 1, 16};
    for (int i = 0; i < 9; ++i) printf("%-45s %8.1f cycles\n", names[i], (double)h[i] / n / div[i]);
    return 0;
}
