// Block-min open list (device code, included by search_kernel.hip inside its anonymous namespace).
//
// The reference's open list is a std::priority_queue (priority_queue_interface_mex.cpp:23-29): which of several
// entries with EQUAL minimal key it pops depends on the layout of its binary heap.  As long as the minimal key is
// unique at every pop, the pop sequence is a function of the key set alone and any exact priority queue reproduces
// it.  This queue is built for one 64-lane wavefront and certifies that uniqueness at every pop; the first tie makes
// the search start over on the libstdc++-faithful binary heap (heap_push / heap_pop), so results never depend on it.
// Measured on the closed-loop C2 workload: 0 tied pops in 311 235 (tools/tie_frequency.py).
//
// Layout: the entry of node i (0-based tree index; nodes are pushed exactly once, in index order) is key[i].  Keys are
// never overwritten: pbits[b] holds the popped bits of block b = nodes [64 b, 64 b + 64).  m1[b] = min key of the
// entries of block b that are still in the queue; m2[g] = min of m1[64 g .. 64 g + 64), maintained once the tree has
// more than 4096 nodes.
//   push of up to 64 consecutive nodes   one key store per lane + one or two LDS atomic-min per lane
//   pop                                  up to three dependent 64-wide loads (m2, m1 of the best group, keys of the best
//                                        block), one wave min-reduction to find the minimum, two ballots to locate it,
//                                        one or two reductions for the new m1[b] and m2[g]; sets the popped bit
//   un-pop                               clears the popped bit, one or two LDS atomic-min
// Keys of the KR most recent nodes sit in an LDS ring (slot = i & (KR - 1)); every key is also written to HBM, which is
// where blocks older than the ring are read from.  Keys are non-negative finite doubles (sums of squares), so their
// bit patterns order like unsigned integers: the atomics are ds_min_u64.
#pragma once

typedef LDS_AS unsigned long long lds_u64;

#define BM_INF_BITS 0x7FF0000000000000ull

struct BmQueue {
    lds_f64* kring;  // [KR]
    lds_f64* m1;     // [nb_max]
    lds_u64* pbits;  // [nb_max] bit e of word b set: node 64 b + e has been popped (its key stays where it is)
    lds_f64* m2;     // [64]
    double* gkey;    // [max_nodes]
    uint32_t kr_mask;
    uint32_t nb_max;
    bool tie;       // a pop found its minimal key more than once: the pop order is not certified any more
};

// Dropping entries whose edge is already known to collide (GraphSearch.m:75-77 would pop and discard them one by one):
// whenever a pop looks at a block, the entries of that block with validity byte `invalid_code` leave the open list along
// with the popped node (same read of the keys, same reduction for the new block minimum).  Their node indices go to
// `list`, the number of pops made so far (`stamp`) to `stamps` (filled from the back); the caller works out at the end
// which of them the reference would have popped (see search_kernel.hip).
struct BmDrop {
    bool on;
    volatile LDS_AS uint8_t* validity;  // one byte per node: the first nv nodes in LDS ...
    uint8_t* gvalidity;                 // ... the others in HBM (nullptr: nodes beyond nv are not dropped)
    uint32_t nv;
    uint32_t invalid_code, dropped_code;
    uint32_t* list;
    uint32_t* stamps_end;  // stamp of entry i at stamps_end[-1 - i]
    uint32_t stamp;
    uint32_t n;
};

__device__ __forceinline__ double bm_inf() { return __longlong_as_double((long long)BM_INF_BITS); }

// unsigned minimum over the 64 lanes; lane 63 of the result holds it.  v_min_u32 takes its DPP-permuted operand
// directly, so a step is one VALU instruction (plus the two wait states a DPP read of a just-written VGPR needs).
__device__ __forceinline__ uint32_t wave_min_u32_lane63(uint32_t v) {
    asm volatile(
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"  // every lane: minimum of its row of 16
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"  // rows 1, 3 take in rows 0, 2
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"  // rows 2, 3 take in lane 31
        "s_nop 1"
        : "+v"(v));
    return v;
}
// minimum over the 64 lanes of non-negative doubles (no NaNs), uniform result: their bit patterns order like unsigned
// integers, so the high words are reduced first; if a single lane holds the minimal high word (the usual case) its low
// word completes the answer, otherwise the low words of those lanes are reduced as well
__device__ __forceinline__ double wave_min_d(double x) {
    const uint64_t u = (uint64_t)__double_as_longlong(x);
    const uint32_t hi = (uint32_t)(u >> 32), lo = (uint32_t)u;
    const uint32_t mh = (uint32_t)__builtin_amdgcn_readlane((int)wave_min_u32_lane63(hi), 63);
    const unsigned long long holders = __ballot(hi == mh);
    uint32_t ml;
    if ((holders & (holders - 1ull)) == 0ull)
        ml = (uint32_t)__builtin_amdgcn_readlane((int)lo, __builtin_ctzll(holders));
    else
        ml = (uint32_t)__builtin_amdgcn_readlane((int)wave_min_u32_lane63(hi == mh ? lo : 0xFFFFFFFFu), 63);
    return __longlong_as_double((long long)(((uint64_t)mh << 32) | ml));
}

// Two independent unsigned minima at once: the DPP steps of one reduction fill the wait states of the other (measured:
// 160 cycles for the pair against 144 for one, tools/ubench_ilp.hip).
__device__ __forceinline__ void wave_min2_u32_lane63(uint32_t& a, uint32_t& b) {
    asm volatile(
        "s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_min_u32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(a), "+v"(b));
}
// low word of the minimum whose high word is mh (held by the lanes in `holders`)
__device__ __forceinline__ uint32_t wave_min_low(uint32_t hi, uint32_t lo, uint32_t mh, unsigned long long holders) {
    if ((holders & (holders - 1ull)) == 0ull) return (uint32_t)__builtin_amdgcn_readlane((int)lo, __builtin_ctzll(holders));
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_min_u32_lane63(hi == mh ? lo : 0xFFFFFFFFu), 63);
}
// minima of x and of y over the 64 lanes (non-negative doubles, no NaNs), uniform results
__device__ __forceinline__ void wave_min2_d(double x, double y, double& mx, double& my) {
    const uint64_t ux = (uint64_t)__double_as_longlong(x), uy = (uint64_t)__double_as_longlong(y);
    const uint32_t xh = (uint32_t)(ux >> 32), xl = (uint32_t)ux, yh = (uint32_t)(uy >> 32), yl = (uint32_t)uy;
    uint32_t a = xh, b = yh;
    wave_min2_u32_lane63(a, b);
    const uint32_t mxh = (uint32_t)__builtin_amdgcn_readlane((int)a, 63), myh = (uint32_t)__builtin_amdgcn_readlane((int)b, 63);
    const uint32_t mxl = wave_min_low(xh, xl, mxh, __ballot(xh == mxh));
    const uint32_t myl = wave_min_low(yh, yl, myh, __ballot(yh == myh));
    mx = __longlong_as_double((long long)(((uint64_t)mxh << 32) | mxl));
    my = __longlong_as_double((long long)(((uint64_t)myh << 32) | myl));
}

__device__ __forceinline__ void bm_init(BmQueue& Q, int tid, int nthreads) {
    const double inf = bm_inf();
    for (uint32_t i = (uint32_t)tid; i < Q.nb_max; i += (uint32_t)nthreads) {
        Q.m1[i] = inf;
        Q.pbits[i] = 0ull;
    }
    for (uint32_t i = (uint32_t)tid; i < 64u; i += (uint32_t)nthreads) Q.m2[i] = inf;
}

// Entries written into the ring by the expanding wave but not yet pushed by the queue's owner may already have
// recycled the ring slots of the oldest nodes: a block counts as resident in the ring only if it starts at least
// BM_INFLIGHT nodes inside it (the owner's tree size lags the expander's by at most one expansion).
#define BM_INFLIGHT 128u

__device__ __forceinline__ bool bm_in_ring(const BmQueue& Q, uint32_t block, uint32_t nn) { return block * 64u + Q.kr_mask + 1u >= nn + BM_INFLIGHT; }

// Make the nodes i0 (0-based, consecutive over the active lanes, all >= nn_before) with keys f visible to bm_pop.
// STORE_KEYS = false: the keys are already in the ring and in HBM (written by the expanding wave), only the block and
// group minima are updated.  Whole wave calls; nn_after = tree size after this batch.
template <bool STORE_KEYS>
__device__ __forceinline__ void bm_push(BmQueue& Q, bool active, uint32_t i0, double f, uint32_t nn_before, uint32_t nn_after) {
    // entering a second group for the first time: m2 is not maintained while there is a single group
    if (nn_before <= 4096u && nn_after > 4096u) {
        const int lane = (int)(threadIdx.x & 63u);
        const double v = wave_min_d(Q.m1[lane]);
        if (lane == 0) Q.m2[0] = v;
    }
    if (active) {
        if (STORE_KEYS) {
            Q.kring[i0 & Q.kr_mask] = f;
            Q.gkey[i0] = f;
        }
        const unsigned long long bits = (unsigned long long)__double_as_longlong(f);
        __hip_atomic_fetch_min((lds_u64*)&Q.m1[i0 >> 6], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (nn_after > 4096u) __hip_atomic_fetch_min((lds_u64*)&Q.m2[i0 >> 12], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// keys of block b, one per lane (+inf for nodes that do not exist yet or have been popped)
__device__ __forceinline__ double bm_block_keys(const BmQueue& Q, uint32_t b, uint32_t nn, bool in_ring, unsigned long long& popped) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t idx = b * 64u + lane;
    popped = Q.pbits[b];
    double k;
    if (in_ring) {
        k = Q.kring[idx & Q.kr_mask];
    } else {
        k = Q.gkey[idx];
        // wait here, not after the join: there the wait would also be paid on the LDS path, where it only drains this
        // wave's earlier HBM stores (gfx9 counts loads and stores in the same vmcnt)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    }
    if (idx >= nn || ((popped >> lane) & 1ull)) k = bm_inf();  // (slots of nodes that do not exist yet hold stale keys)
    return k;
}

struct BmFound {
    uint32_t idx;  // 0-based node index; 0xFFFFFFFF: the queue is empty
    double key;
};

// Remove node idx (which must be in the queue): key -> +inf, block and group minima recomputed.
__device__ __forceinline__ void bm_remove(BmQueue& Q, uint32_t idx, uint32_t nn) {
    const uint32_t lane = threadIdx.x & 63u;
    const double inf = bm_inf();
    const uint32_t b = idx >> 6, e = idx & 63u;
    const bool in_ring = bm_in_ring(Q, b, nn);
    unsigned long long popped;
    const double k = bm_block_keys(Q, b, nn, in_ring, popped);
    const double new1 = wave_min_d(lane == e ? inf : k);
    if (lane == e) {
        Q.pbits[b] = popped | (1ull << e);  // (the queue's owner is the only writer of the popped bits)
        Q.m1[b] = new1;
    }
    if (nn > 4096u) {
        const uint32_t g = b >> 6;
        const double v1 = Q.m1[g * 64u + lane];
        const double new2 = wave_min_d(lane == (b & 63u) ? new1 : v1);
        if (lane == 0) Q.m2[g] = new2;
    }
}

// Put popped nodes back, one per active lane (their keys are still in place): the inverse of pops that turned out to be
// premature.  Whole wave calls.
__device__ __forceinline__ void bm_unpop_lanes(BmQueue& Q, bool active, uint32_t idx, double key, uint32_t nn) {
    if (active) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(key);
        __hip_atomic_fetch_and(&Q.pbits[idx >> 6], ~(1ull << (idx & 63u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_min((lds_u64*)&Q.m1[idx >> 6], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (nn > 4096u) __hip_atomic_fetch_min((lds_u64*)&Q.m2[idx >> 12], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// Pop: the entry with the minimal key is found and removed with a single look at its block.  Returns idx 0xFFFFFFFF if
// the queue is empty.  Sets Q.tie if the minimum was not unique.
__device__ __forceinline__ BmFound bm_pop(BmQueue& Q, uint32_t nn, BmDrop& D) {
    const uint32_t lane = threadIdx.x & 63u;
    const double inf = bm_inf();
    BmFound r;
    r.idx = 0xFFFFFFFFu;
    r.key = inf;
    const bool multi = nn > 4096u;
    uint32_t g = 0;
    double v1, mn1;
    if (multi) {
        const double v2 = Q.m2[lane];
        mn1 = wave_min_d(v2);  // the minimum of the best group is the minimum of one of its blocks: no second reduction
        if (!(mn1 < inf)) return r;
        const unsigned long long b2 = __ballot(v2 == mn1);
        if (b2 & (b2 - 1ull)) Q.tie = true;
        g = (uint32_t)__builtin_ctzll(b2);
        v1 = Q.m1[g * 64u + lane];
    } else {
        v1 = Q.m1[lane];  // m1 is +inf beyond the last block
        mn1 = wave_min_d(v1);
        if (!(mn1 < inf)) return r;
    }
    const unsigned long long b1 = __ballot(v1 == mn1);
    if (b1 & (b1 - 1ull)) Q.tie = true;
    const uint32_t bl = (uint32_t)__builtin_ctzll(b1);
    const uint32_t b = g * 64u + bl;
    const uint32_t idx = b * 64u + lane;
    uint32_t vcode = 0;
    if (D.on) {
        if (idx < D.nv)
            vcode = D.validity[idx];
        else if (D.gvalidity && idx < nn)  // (one coalesced byte load, in flight together with the block's keys)
            vcode = (uint32_t)__hip_atomic_load(D.gvalidity + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    unsigned long long popped;
    const double k = bm_block_keys(Q, b, nn, bm_in_ring(Q, b, nn), popped);
    const unsigned long long b0 = __ballot(k == mn1);
    if (b0 & (b0 - 1ull)) Q.tie = true;
    const uint32_t e = (uint32_t)__builtin_ctzll(b0);
    r.idx = b * 64u + e;
    r.key = mn1;
    // entries of this block that are known to collide leave with it
    const bool dead = D.on && k < inf && lane != e && vcode == D.invalid_code;
    const unsigned long long dmask = __ballot(dead);
    if (dmask) {
        if (dead) {
            if (idx < D.nv)
                D.validity[idx] = (uint8_t)D.dropped_code;
            else
                __hip_atomic_store(D.gvalidity + idx, (uint8_t)D.dropped_code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t at = D.n + __builtin_amdgcn_mbcnt_hi((uint32_t)(dmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dmask, 0u));
            D.list[at] = idx;
            D.stamps_end[-1 - (int)at] = D.stamp;
        }
        D.n += (uint32_t)__builtin_popcountll(dmask);
    }
    const bool gone = lane == e || dead;
    // remove: popped bit, m1[b] = min of the rest, m2[g] = min over the group's blocks = the smaller of m1[b] and the best
    // of the other blocks (two independent reductions, interleaved)
    double new1;
    if (multi) {
        double others;
        wave_min2_d(gone ? inf : k, lane == bl ? inf : v1, new1, others);
        if (lane == 0) Q.m2[g] = new1 < others ? new1 : others;
    } else {
        new1 = wave_min_d(gone ? inf : k);
    }
    if (lane == e) {
        Q.pbits[b] = popped | (1ull << e) | dmask;  // (the queue's owner is the only writer of the popped bits)
        Q.m1[b] = new1;
    }
    return r;
}
__device__ __forceinline__ BmFound bm_pop(BmQueue& Q, uint32_t nn) {
    BmDrop none;
    none.on = false;
    none.validity = nullptr;
    none.gvalidity = nullptr;
    none.nv = 0;
    none.invalid_code = none.dropped_code = 0;
    none.list = nullptr;
    none.stamps_end = nullptr;
    none.stamp = 0;
    none.n = 0;
    return bm_pop(Q, nn, none);
}
