// Per-vehicle search state: the tree's node records (LDS for the first NL nodes, HBM for all) and the open-list arrays of the binary heap (device code, included by search_kernel.hip inside its anonymous namespace).
#pragma once

// per-lane constants of the sift-down rounds (computed once per kernel)
struct PopLane {
    int d;                       // level below the hole: 1..5 (6 for the unused lanes 62, 63)
    uint32_t q;                  // position inside the level
    unsigned long long ancmask;  // bits (node index n = lane + 2, heap order, hole = node 1) of the node and its ancestors
};
__device__ __forceinline__ PopLane make_pop_lane(int lane) {
    PopLane L;
    L.d = 31 - __builtin_clz((uint32_t)lane + 2u);
    L.q = (uint32_t)lane + 2u - (1u << L.d);
    unsigned long long m = 0;
    for (uint32_t a = (uint32_t)lane + 2u; a >= 2u; a >>= 1) m |= 1ull << a;
    L.ancmask = (lane < 62) ? m : ~0ull;  // lanes 62, 63 never match
    return L;
}

// Per-vehicle search state.  l* point into LDS, g* into this vehicle's HBM slices.
struct Search {
    lds_d2* ln;  // NodeRec[NL] as 4 x double2 each
    NodeRec* gn;
    uint32_t NL, max_nodes;
    lds_f64* lkey;
    lds_u32* lid;
    double* gkey;
    uint32_t* gid;
    uint32_t HL;
    uint32_t heap_len;
    int lane;
    PopLane pl;
    PROF_MEMBERS
};

union NodeBits {
    NodeRec r;
    d2 q[4];
    __device__ NodeBits() {}
};

__device__ __forceinline__ NodeRec node_load(const Search& S, uint32_t i0) {
    NodeBits u;
    if (i0 < S.NL) {
        lds_d2* p = S.ln + 4 * (size_t)i0;
        u.q[0] = p[0];
        u.q[1] = p[1];
        u.q[2] = p[2];
        u.q[3] = p[3];
    } else {
        const d2* p = (const d2*)(S.gn + i0);
        u.q[0] = p[0];
        u.q[1] = p[1];
        u.q[2] = p[2];
        u.q[3] = p[3];
    }
    return u.r;
}
__device__ __forceinline__ void node_store(const Search& S, uint32_t i0, const NodeRec& r) {
    NodeBits u;
    u.r = r;
    d2* g = (d2*)(S.gn + i0);
    g[0] = u.q[0];
    g[1] = u.q[1];
    g[2] = u.q[2];
    g[3] = u.q[3];
    if (i0 < S.NL) {
        lds_d2* p = S.ln + 4 * (size_t)i0;
        p[0] = u.q[0];
        p[1] = u.q[1];
        p[2] = u.q[2];
        p[3] = u.q[3];
    }
}
__device__ __forceinline__ void node_store_cs(const Search& S, uint32_t i0, double cs, double sn) {
    d2 v;
    v.x = cs;
    v.y = sn;
    ((d2*)(S.gn + i0))[2] = v;
    if (i0 < S.NL) S.ln[4 * (size_t)i0 + 2] = v;
}
__device__ __forceinline__ uint32_t node_parent(const Search& S, uint32_t i0) {
    d2 v;
    if (i0 < S.NL)
        v = S.ln[4 * (size_t)i0 + 3];
    else
        v = ((const d2*)(S.gn + i0))[3];
    return (uint32_t)((uint64_t)__double_as_longlong(v.y) & 0xffffffffull);
}
