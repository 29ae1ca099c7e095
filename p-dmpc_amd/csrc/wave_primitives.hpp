// Wavefront-level primitives: uniform broadcasts, LDS pointer types, staging (device code, included by search_kernel.hip inside its anonymous namespace).
#pragma once

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uni_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ double uni_d(double v) {
    uint64_t u = (uint64_t)__double_as_longlong(v);
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ double lane_d(double v, int lane_uniform) {
    uint64_t u = (uint64_t)__double_as_longlong(v);
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, lane_uniform);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), lane_uniform);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ uint32_t lane_u(uint32_t v, int lane_uniform) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane_uniform); }
__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }
__device__ __forceinline__ bool is_nan(double v) { return v != v; }

// ---------------------------------------------------------------------------------------------------
// LDS pointers carry their address space in the type so every access is a ds_* instruction (a generic pointer that
// may be LDS or HBM compiles to slower flat_* accesses).
#define LDS_AS __attribute__((address_space(3)))
typedef LDS_AS d2 lds_d2;
typedef LDS_AS double lds_f64;
typedef LDS_AS uint32_t lds_u32;
typedef LDS_AS int32_t lds_i32;
typedef LDS_AS int16_t lds_i16;
typedef LDS_AS uint8_t lds_u8;
typedef LDS_AS uint64_t lds_mask64;
typedef LDS_AS DevManPose lds_pose;

// copy `count` 16-byte elements HBM -> LDS, thread-strided over the whole workgroup (coalesced)
__device__ __forceinline__ void stage16(LDS_AS void* dst_lds, const void* src, int count, int tid) {
    lds_d2* d = (lds_d2*)dst_lds;
    const d2* s = (const d2*)src;
    for (int i = tid; i < count; i += (int)blockDim.x) d[i] = s[i];
}

// order this wave's LDS/HBM writes before its later reads (same wave: the hardware keeps DS order; this stops the
// compiler from moving accesses)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t lds_load_u32(const volatile lds_u32* p) { return *p; }
