// The libstdc++-faithful binary heap (priority_queue_interface_mex.cpp:19-31): exact for any keys, including the tie order among equal ones (device code, included by search_kernel.hip inside its anonymous namespace).
#pragma once

// per-lane heap access (index may differ per lane).  LDSONLY: the caller knows every index is < HL.  Otherwise the
// LDS part and the HBM part are two separately predicated accesses (no generic pointers).
template <bool LDSONLY>
__device__ __forceinline__ void heap_load(const Search& S, uint32_t idx, bool valid, double& k, uint32_t& id) {
    k = 0.0;
    id = 0;
    if (LDSONLY) {
        if (valid) {
            k = S.lkey[idx];
            id = S.lid[idx];
        }
    } else {
        const bool inl = valid && idx < S.HL;
        const bool ing = valid && idx >= S.HL;
        if (inl) {
            k = S.lkey[idx];
            id = S.lid[idx];
        }
        if (ing) {
            // spilled entries: L1-bypassing (sc1) loads, so an entry this wave stored a moment ago is read from L2, where
            // the wave's in-order write-through store has already landed; no fence / store drain needed between heap phases
            k = __hip_atomic_load(S.gkey + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            id = __hip_atomic_load(S.gid + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
template <bool LDSONLY>
__device__ __forceinline__ void heap_store(const Search& S, uint32_t idx, double k, uint32_t id) {
    if (LDSONLY || idx < S.HL) {
        S.lkey[idx] = k;
        S.lid[idx] = id;
    } else {
        S.gkey[idx] = k;
        S.gid[idx] = id;
    }
}
// order one phase's heap writes before the next phase's reads (LDS: in-order DS queue per wave; HBM spill:
// same-CU L1, needs the stores drained)
template <bool LDSONLY>
__device__ __forceinline__ void heap_fence() {
    if (!LDSONLY) __threadfence_block();  // measured: free next to the L2 round trips of the spilled levels
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // compiler ordering only
    __builtin_amdgcn_wave_barrier();
}

// neighbour lane's value (lane ^ 1) through DPP quad_perm [1,0,3,2]: no LDS round trip
__device__ __forceinline__ double swap_pair_d(double v) {
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, 0xB1, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), 0xB1, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo));
}

// libstdc++ __push_heap(first, hole, top = 0, value) with comp(a, b) = a.key > b.key
// (priority_queue_interface_mex.cpp:23-29): the value climbs while the parent's key is
// STRICTLY greater.  Lane t fetches ancestor t of the hole; a ballot finds the first ancestor that stays.
template <bool LDSONLY>
__device__ __forceinline__ void heap_sift_up(Search& S, uint32_t hole, double key, uint32_t id) {
    const uint32_t h1 = hole + 1;
    const int nlev = 31 - __builtin_clz(h1);  // number of ancestors, <= 17 for a 128k heap
    const int lane = S.lane;
    const bool valid = lane < nlev;
    const uint32_t anc = valid ? ((h1 >> (lane + 1)) - 1u) : 0u;
    double k;
    uint32_t i;
    heap_load<LDSONLY>(S, anc, valid, k, i);
    const bool gt = valid && (k > key);
    const unsigned long long b = __ballot(gt);
    const int cnt = (~b == 0ull) ? 64 : (int)__builtin_ctzll(~b);  // ancestors that move down one level
    if (lane < cnt) heap_store<LDSONLY>(S, (h1 >> lane) - 1u, k, i);
    if (lane == 0) heap_store<LDSONLY>(S, (h1 >> cnt) - 1u, key, id);
}

__device__ __forceinline__ void heap_push(Search& S, uint32_t id, double key) {
    const uint32_t hole = S.heap_len;
    S.heap_len = hole + 1;
    if (hole < S.HL) {
        heap_sift_up<true>(S, hole, key, id);
        heap_fence<true>();
    } else {
        heap_sift_up<false>(S, hole, key, id);
        heap_fence<false>();
    }
}

// std::pop_heap + pop_back: libstdc++ __pop_heap -> __adjust_heap(first, 0, len, value) -> __push_heap.
// The hole always sinks to a leaf, choosing the right child unless key[right] > key[left].  Each round fetches the
// five levels below the hole (2 + 4 + 8 + 16 + 32 = 62 entries, lane l holds sub-tree node l + 2 in heap order, so
// siblings are lanes l ^ 1 and the children of lane l are lanes 2l + 2, 2l + 3).  Every lane decides whether it is
// the child its parent would step to; a ballot + five scalar steps follow the chain from the hole; the lanes on
// the chain store their entry one level up in one instruction.
// one round of the sift-down: returns the new hole
template <bool LDSONLY>
__device__ __forceinline__ uint32_t heap_pop_round(Search& S, uint32_t hole, uint32_t half, const PopLane& L, double& moved_key) {
    const uint32_t idx = ((hole + 1u) << L.d) - 1u + L.q;
    const uint32_t pidx = (idx - 1u) >> 1;
    const bool step_ok = pidx < half;  // the parent has two children (adjust_heap loop condition)
    double k;
    uint32_t i;
    heap_load<LDSONLY>(S, idx, step_ok, k, i);
    const double ks = swap_pair_d(k);  // sibling's key
    // right child (odd lane) is stepped to unless key[right] > key[left]; left child (even lane) iff key[right] > key[left]
    const unsigned long long gt_self = __ballot(k > ks);   // on an odd lane: key[right] > key[left]
    const unsigned long long gt_sib = __ballot(ks > k);    // on an even lane: key[right] > key[left]
    const unsigned long long ODD = 0xAAAAAAAAAAAAAAAAull;
    const unsigned long long pref = ((~gt_self) & ODD) | (gt_sib & ~ODD);
    // bit n of Q: sub-tree node n (= lane + 2) is the child its parent steps to
    const unsigned long long Q = (pref & __ballot(step_ok)) << 2;
    const bool on = (Q & L.ancmask) == L.ancmask;  // the node and all its ancestors are stepped to: it is on the chain
    const unsigned long long pathmask = __ballot(on);
    if (on) heap_store<LDSONLY>(S, pidx, k, i);
    const int last = 63 - (int)__builtin_clzll(pathmask);  // deepest chain lane (the chain is never empty: hole < half)
    moved_key = lane_d(k, last);  // the entry that now sits in the parent of the new hole
    return lane_u(idx, last);
}

template <bool LDSONLY>
__device__ __forceinline__ void heap_pop_impl(Search& S, uint32_t len) {
    const int lane = S.lane;
    double vkey;
    uint32_t vid;
    heap_load<LDSONLY>(S, len, true, vkey, vid);
    const uint32_t half = (len - 1) >> 1;
    uint32_t hole = 0;
    double parent_key = 0.0;  // key of the entry that moved into the parent of the current hole
    while (hole < half) {
        // the five levels below `hole` end at index 32 * (hole + 1) + 30
        if (LDSONLY || ((hole + 1u) << 5) + 30u < S.HL)
            hole = heap_pop_round<true>(S, hole, half, S.pl, parent_key);
        else
            hole = heap_pop_round<false>(S, hole, half, S.pl, parent_key);
    }
    if ((len & 1u) == 0 && hole == ((len - 2u) >> 1)) {  // lone left child at the bottom
        const uint32_t child = 2u * (hole + 1u);
        double ck;
        uint32_t cid;
        heap_load<LDSONLY>(S, child - 1u, true, ck, cid);
        if (lane == 0) heap_store<LDSONLY>(S, hole, ck, cid);
        parent_key = uni_d(ck);
        hole = child - 1u;
    }
    vkey = uni_d(vkey);
    vid = uni_u(vid);
    // __push_heap(first, hole, 0, value): the value climbs only while parent.key > value.key; the parent of the hole holds
    // the entry that just moved up, whose key is still in a register -> the common "stays put" case needs no memory read
    if (hole == 0 || !(parent_key > vkey)) {
        if (lane == 0) heap_store<LDSONLY>(S, hole, vkey, vid);
        heap_fence<LDSONLY>();
        return;
    }
    heap_fence<LDSONLY>();
    heap_sift_up<LDSONLY>(S, hole, vkey, vid);
    heap_fence<LDSONLY>();
}

__device__ __forceinline__ void heap_pop(Search& S) {
    const uint32_t len = S.heap_len - 1;  // length after the pop
    S.heap_len = len;
    if (len == 0) return;
    if (len < S.HL)
        heap_pop_impl<true>(S, len);
    else
        heap_pop_impl<false>(S, len);
}
