// frontier_common.hpp — device code of the round-based search (bulk_search.hpp) that is not the round loop itself: shared-word
// indices, LDS counter helpers, the list partition and histogram, the walks that decide whether an open node comes before the goal
// candidate, goal-candidate resolution, and phase B (the reference's counts and ids).  Included after search_common.hpp.
// (The name is history: rounds 2-3 had a "frontier kernel" — one node per wavefront — next to this code; docs/HISTORY.md.)
#pragma once
#include <type_traits>

// frontier words in the shared block (indices >= 32; the serial search uses the words below)
#define FR_NNODES 32    // tree size (atomic reservation of node indices)
#define FR_RD_HEAD 33   // ready list: next entry to claim
#define FR_RD_TAIL 34   // ready list: entries reserved
#define FR_VLIST_N 35   // arrival handling: entries of the list of collision-free nodes that are being re-checked
#define FR_NEAR_N 36
#define FR_FAR_N 37
#define FR_FLAGS 38     // FRF_*
#define FR_BEST_ID 40   // best goal candidate so far (1-based node, 0 = none)
#define FR_SEL_BIN 41   // result of fr_select: bin ...
#define FR_SEL_CUM 42   // ... and the number of entries up to and including it
#define FR_SEL2_BIN 43  // second selection of the same histogram (spill boundary) ...
#define FR_SEL2_CUM 44  // ... and its count
#define FR_ROUNDS 45
#define FR_PROCESSED 60
#define FR_BEST_B1 46   // (64 bit) largest key on the best candidate's path
#define FR_NEAR_MIN 48  // (64 bit) exact minimum key of near
#define FR_NEAR_MAX 50  // (64 bit) upper bound of near's keys
#define FR_FAR_MIN 52   // (64 bit) exact minimum key of far
#define FR_FAR_MAX 54   // (64 bit) upper bound of far's keys
#define FR_L_FAR 58     // (64 bit) children with key > this go to far
#define FR_PATH_FOR 61  // the goal candidate whose path is in the relevance tables (0: none)
#define FR_EVER_INVAL 63 // set once a late arrival has invalidated a node of this search
#define FR_SLOWEST 30   // (64 bit, words 30-31 of the serial block: unused by both searches) debugging: slowest node
#define FR_GOAL_N 39     // goal candidates of the running round (entries of goal_list)
#define FR_ROUND_B1 28   // (64 bit, words 28-29 of the serial block: unused by both searches) smallest path maximum among them
#define FR_GOAL_CAP 2048
#define FR_JOIN_MAX 26    // (64 bit, words 26-27 of the serial block: unused by both searches) largest key among the round's entries
#define FR_DEAD 57      // open entries dropped because an ancestor was invalidated
#define FR_HELP_CLOSED 56 // shared round: entries of the shared part the helpers claimed before the owner closed it
#define FR_DROPPED 62   // open entries dropped because they come after the best candidate (restored if that one is invalidated)
#define FRF_OVERFLOW 1u
#define FRF_TIE 2u
#define FRF_INVALIDATED 4u
#define FRF_BUG 8u
#define FRF_GOALS_LOST 16u  // bulk kernel: a round had more goal candidates than its list holds: every collision-free node at the horizon is offered again
#define FR_SCRATCH 64   // 64 scratch words behind the shared block (targets of the lanes that only take part pro forma, see sh_add_uniform)
#define FR_NBINS 2048
#define FR_READY_CAP 1536

// words of a helper workgroup's shared block
#define HS_CMD 0     // 0 nothing found, 1 work, 2 every search has finished
#define HS_SLOT 1
#define HS_FIRST 2
#define HS_COUNT 3
#define HS_MASK_LO 4
#define HS_MASK_HI 5
#define HS_TICKET 6
#define HS_EXPAND 7  // the claimed round wants its collision-free entries expanded
#define HS_RUN_BASE 8   // first node index of the block the run's children get
#define HS_RUN_TOTAL 9  // children of the run

namespace {

typedef LDS_AS unsigned long long lds_u64s;

// live counters for debugging (host-mapped memory, PDMPC_DEBUG_PROGRESS=1): stage = where the workgroup is
#define FR_PROGRESS(stage)                                                                    \
    if (A.progress && threadIdx.x == 0) {                                                     \
        volatile uint32_t* pg__ = A.progress + (size_t)X.slot * 64;                         \
        pg__[0] = sh[FR_ROUNDS];                                                              \
        pg__[1] = sh[FR_PROCESSED];                                                           \
        pg__[2] = sh[FR_NNODES];                                                              \
        pg__[3] = sh[FR_NEAR_N];                                                              \
        pg__[4] = sh[FR_FAR_N];                                                               \
        pg__[5] = sh[FR_FLAGS];                                                               \
        pg__[6] = sh[FR_BEST_ID];                                                             \
        pg__[7] = (stage);                                                                    \
        pg__[8] = sh[FR_VLIST_N];                                                             \
        pg__[9] = sh[FR_RD_HEAD];                                                             \
        pg__[10] = sh[FR_RD_TAIL];                                                            \
        pg__[11] += 1u;                                                                       \
    }

__device__ __forceinline__ double sh_ld_d(volatile lds_u32* sh, int w) { return __longlong_as_double((long long)*(volatile lds_u64s*)(sh + w)); }
__device__ __forceinline__ void sh_st_d(volatile lds_u32* sh, int w, double v) { *(volatile lds_u64s*)(sh + w) = (unsigned long long)__double_as_longlong(v); }
// keys are non-negative finite doubles (sums of squares): their bit patterns order like unsigned integers
__device__ __forceinline__ void sh_min_d(volatile lds_u32* sh, int w, double v) {
    __hip_atomic_fetch_min((lds_u64s*)(sh + w), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void sh_max_d(volatile lds_u32* sh, int w, double v) {
    __hip_atomic_fetch_max((lds_u64s*)(sh + w), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// Smallest / largest key a wave has met (mn = +inf, mx = 0 where a lane met none) folded into two shared words with one
// atomic each: sixty-four lanes doing the same-address LDS atomic themselves are served one after the other.  Whole wave calls.
__device__ __forceinline__ void sh_minmax_wave(volatile lds_u32* sh, int wmin, int wmax, double mn, double mx, int lane) {
#pragma unroll
    for (int o = PDMPC_WAVE / 2; o > 0; o >>= 1) {
        const double a = __shfl_xor(mn, o), c = __shfl_xor(mx, o);
        mn = a < mn ? a : mn;
        mx = c > mx ? c : mx;
    }
    if (lane == 0) {
        sh_min_d(sh, wmin, mn);
        sh_max_d(sh, wmax, mx);
    }
}
__device__ __forceinline__ uint32_t sh_add(volatile lds_u32* sh, int w, uint32_t v) {
    return __hip_atomic_fetch_add((lds_u32*)(sh + w), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// A wave-wide counter update whose old value every lane needs.  The obvious `if (lane == 0) old = atomic(...); old =
// readfirstlane(old)` puts a divergent branch in front of a wave-uniform value: hipcc 7.2 threads the lanes that skip the
// branch past it and lets them run the code that follows — wave-wide node processing — apart from lane 0.  So there is no
// branch: EVERY lane issues the LDS atomic, lane 0 on the counter, the others on a scratch word of their own (distinct
// addresses: one pass through the LDS), and lane 0's result is broadcast.
__device__ __forceinline__ uint32_t sh_add_uniform(volatile lds_u32* sh, int w, uint32_t v, int lane) {
    lds_u32* p = lane == 0 ? (lds_u32*)(sh + w) : (lds_u32*)(sh + FR_SCRATCH + lane);
    const uint32_t old = __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
}
__device__ __forceinline__ uint32_t lane_rank(unsigned long long ballot, int lane) { return (uint32_t)__builtin_popcountll(ballot & ((1ull << lane) - 1ull)); }

// monotone map key -> bin of a linear histogram over [lo, lo + nb / scale)
__device__ __forceinline__ uint32_t fr_bin(double key, double lo, double scale) {
    const double t = (key - lo) * scale;
    if (!(t > 0.0)) return 0u;
    return t < (double)(FR_NBINS - 1) ? (uint32_t)t : (uint32_t)(FR_NBINS - 1);
}

struct Frontier {
    volatile lds_u32* sh;
    lds_u32* ready;  // [FR_READY_CAP] 1-based nodes of the running round (0 = not written yet)
    lds_u32* hist;   // [FR_NBINS]
    lds_u32* goal_list;  // [FR_GOAL_CAP] goal candidates of the running round (lives in the histogram's first half: free during a round)
    unsigned long long* glink;  // [max_nodes] parent | packed << 32
    double* near_key;
    uint32_t* near_id;
    double* far_key;
    uint32_t* far_id;
    double* gkey;  // key of node i at gkey[i]
    int n_waves;
};

// Smallest bin whose cumulative count reaches `target` (the last non-empty bin if the total is smaller), for two targets
// in one pass over the histogram.  One wave calls; bin and cumulative count go to sh[w1], sh[w1 + 1] and sh[w2], sh[w2 + 1].
__device__ void fr_select2(const Frontier& F, uint32_t target1, uint32_t target2, int w1, int w2, int lane) {
    const int per = FR_NBINS / PDMPC_WAVE;
    uint32_t hq[FR_NBINS / PDMPC_WAVE];
    uint32_t loc = 0;
#pragma unroll
    for (int q = 0; q < per; ++q) {
        hq[q] = F.hist[lane * per + q];
        loc += hq[q];
    }
    uint32_t inc = loc;  // inclusive prefix over the lanes
#pragma unroll
    for (int o = 1; o < PDMPC_WAVE; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, o);
        if (lane >= o) inc += v;
    }
    const uint32_t total = lane_u(inc, PDMPC_WAVE - 1);
    const uint32_t want1 = target1 < total ? target1 : total, want2 = target2 < total ? target2 : total;
    // every lane looks for the crossings in its own bins (no lane-dependent branch around values that are broadcast later)
    uint32_t c = inc - loc, bin1 = 0, cum1 = 0, bin2 = 0, cum2 = 0;
    bool f1 = false, f2 = false;
#pragma unroll
    for (int q = 0; q < per; ++q) {
        c += hq[q];
        const bool h1 = !f1 && hq[q] != 0u && c >= want1, h2 = !f2 && hq[q] != 0u && c >= want2;
        bin1 = h1 ? (uint32_t)(lane * per + q) : bin1;
        cum1 = h1 ? c : cum1;
        bin2 = h2 ? (uint32_t)(lane * per + q) : bin2;
        cum2 = h2 ? c : cum2;
        f1 = f1 || h1;
        f2 = f2 || h2;
    }
    const unsigned long long r1 = __ballot(f1), r2 = __ballot(f2);
    const int l1 = r1 ? __builtin_ctzll(r1) : 0, l2 = r2 ? __builtin_ctzll(r2) : 0;
    bin1 = lane_u(bin1, l1);
    cum1 = lane_u(cum1, l1);
    bin2 = lane_u(bin2, l2);
    cum2 = lane_u(cum2, l2);
    if (lane == 0) {
        F.sh[w1] = r1 ? bin1 : 0u;
        F.sh[w1 + 1] = r1 ? cum1 : 0u;
        F.sh[w2] = r2 ? bin2 : 0u;
        F.sh[w2 + 1] = r2 ? cum2 : 0u;
    }
}

// Workgroup-wide partition of the list (key[], id[]) of n entries, FR_PER entries per thread and chunk (their loads are
// issued together: the lists live in HBM and a pass is bound by the latency of its loads).  cls(key, id) == 0 keeps an entry
// (compacted in place), any other class hands it to emit(cls, key, id) — which every lane of a wave calls together (cls < 0:
// this lane has nothing), so it can aggregate its atomics per wave.  Returns the number of kept entries.
#define FR_PER 4
template <class Cls, class Emit>
__device__ uint32_t fr_partition(double* key, uint32_t* id, uint32_t n, volatile lds_u32* wsum, int n_waves, Cls cls, Emit emit) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // (opaque, as in fr_phase_b)
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t bd = blockDim.x;
    uint32_t w = 0;
    int buf = 0;
    for (uint32_t base = 0; base < n; base += FR_PER * bd, buf ^= 16) {
        double k[FR_PER];
        uint32_t i[FR_PER];
        int c[FR_PER];
#pragma unroll
        for (int j = 0; j < FR_PER; ++j) {
            const uint32_t e = base + (uint32_t)j * bd + (uint32_t)tid;
            k[j] = e < n ? key[e] : 0.0;
            i[j] = e < n ? id[e] : 0u;
        }
        uint32_t mine = 0;
        unsigned long long keep[FR_PER];
#pragma unroll
        for (int j = 0; j < FR_PER; ++j) {
            c[j] = cls(k[j], i[j]);  // (called by every lane; returns -1 for i == 0)
            keep[j] = __ballot(c[j] == 0);
            mine += (uint32_t)__builtin_popcountll(keep[j]);
        }
        if (lane == 0) wsum[buf + wave] = mine;
        __syncthreads();  // every entry of this chunk has been read
        uint32_t off = 0, tot = 0;
        for (int q = 0; q < n_waves; ++q) {
            const uint32_t v = wsum[buf + q];
            off += q < wave ? v : 0u;
            tot += v;
        }
#pragma unroll
        for (int j = 0; j < FR_PER; ++j) {
            if (c[j] == 0) {
                const uint32_t pos = w + off + lane_rank(keep[j], lane);
                key[pos] = k[j];
                id[pos] = i[j];
            }
            off += (uint32_t)__builtin_popcountll(keep[j]);
            emit(c[j] > 0 ? c[j] : -1, k[j], i[j]);
        }
        w += tot;
    }
    __syncthreads();
    return w;
}

// linear histogram (FR_NBINS bins over [lo, lo + FR_NBINS / scale)) of a list's keys; the caller has zeroed the bins
__device__ void fr_histogram(const Frontier& F, const double* key, uint32_t n, double lo, double scale) {
    const uint32_t bd = blockDim.x;
    for (uint32_t base = 0; base < n; base += FR_PER * bd) {
        double k[FR_PER];
#pragma unroll
        for (int j = 0; j < FR_PER; ++j) {
            const uint32_t e = base + (uint32_t)j * bd + threadIdx.x;
            k[j] = e < n ? key[e] : -1.0;
        }
#pragma unroll
        for (int j = 0; j < FR_PER; ++j)
            if (k[j] >= 0.0) __hip_atomic_fetch_add(&F.hist[fr_bin(k[j], lo, scale)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// What to do with the open nodes a round has selected, one node per lane (a: 1-based, 0 = this lane has none): 1 process it,
// 3 drop it because the reference pops the goal candidate G first (gp_path[d] = node of G's path at depth d, gp_mp[d] =
// largest key of that path below depth d; a leaves the path at some depth d and comes first iff the largest key on its own
// branch is smaller than gp_mp[d]), 4 drop it because one of its ancestors collides with areas that arrived after it was
// created (the reference never creates a).  Sets FRF_TIE on an equality that decides.
// The whole wave walks together (a wave-uniform loop over per-lane states): a per-lane loop in a divergent branch followed
// by a ballot is exactly the shape hipcc 7.2 mis-threads (see sh_add_uniform).
__device__ int fr_check_wave(const unsigned long long* glink, const VState& VS, const double* gkey, const lds_u32* gp_path, const lds_f64* gp_mp, bool have_goal, bool check_alive,
                             uint32_t a, volatile lds_u32* sh) {
    int res = a ? 0 : 1;  // 0: still walking
    uint32_t x = a ? a : 1u;
    double mx = -1.0;
    while (__ballot(res == 0)) {
        const uint32_t i = x - 1u;
        const uint64_t u = glink[i];
        const int d = NODE_K((uint32_t)(u >> 32));
        const bool on_path = have_goal && gp_path[d] == x;  // (the candidate's own ancestors are collision-free: it was validated after the last arrival)
        const double m = gp_mp[d];
        const bool dead = !on_path && x != a && check_alive && vs_load(VS, i) != VS_VALID;
        const double k = gkey[i];
        const uint32_t par = (uint32_t)(u & 0xffffffffull);
        int now = 0;
        now = on_path ? ((x == a || mx < m) ? 1 : 3) : now;
        now = dead ? 4 : now;
        now = (!on_path && !dead && par == 0u) ? 1 : now;  // reached the root: no candidate, every ancestor collision-free
        if (res == 0 && on_path && x != a && mx == m) atomicOr((uint32_t*)&sh[FR_FLAGS], FRF_TIE);
        const bool walking = res == 0;
        res = walking ? now : res;
        mx = (walking && k > mx) ? k : mx;
        x = (walking && now == 0) ? par : x;
    }
    return res;
}

// X (1-based, at the horizon, edge known to be collision-free): largest key on its path, and whether every ancestor is
// still collision-free (a predecessor's late areas may have invalidated one).  Uniform over the wave.
__device__ bool fr_goal_path(const Search& S, const VState& VS, const double* gkey, uint32_t x, double& b1) {
    double m = 0.0;
    bool alive = true;
    uint32_t nd = x;
    for (;;) {
        const double k = gkey[nd - 1];
        m = k > m ? k : m;
        if (nd != x && vs_load(VS, nd - 1) != VS_VALID) alive = false;
        const uint32_t par = node_parent(S, nd - 1);
        if (!par) break;
        nd = par;
    }
    b1 = m;
    return alive;
}

// Which of two nodes of equal depth does the reference pop first?  -1: x, +1: y, 0: undecidable (equal keys).
__device__ int fr_before(const Search& S, const double* gkey, uint32_t x, uint32_t y) {
    double mx = -1.0, my = -1.0;
    while (x != y) {
        const double kx = gkey[x - 1], ky = gkey[y - 1];
        mx = kx > mx ? kx : mx;
        my = ky > my ? ky : my;
        x = node_parent(S, x - 1);
        y = node_parent(S, y - 1);
        if (!x || !y) break;
    }
    return mx < my ? -1 : (my < mx ? 1 : 0);
}

// A valid node at the horizon has been found.  If its ancestors are all collision-free it becomes a goal candidate: the
// largest key of its path goes into its record (the cos / sin slot, which a node at the horizon never needs) and its id into
// the round's candidate list; the best candidate is chosen at the round boundary (fr_resolve_goals) — no lock, nothing a
// wavefront could wait for while it processes a node.  Whole wave calls, uniform.
__device__ void fr_offer_goal(const Frontier& F, const Search& S, const VState& VS, uint32_t x, int lane) {
    double b1;
    const bool alive = fr_goal_path(S, VS, F.gkey, x, b1);
    if (lane == 0 && alive) node_store_cs(S, x - 1u, b1, 0.0);
    const uint32_t pos = sh_add_uniform(F.sh, FR_GOAL_N, alive ? 1u : 0u, lane);
    if (lane == 0 && alive && pos < (uint32_t)FR_GOAL_CAP) F.goal_list[pos] = x;
}

// Round boundary: the best of the round's goal candidates against the best one so far.  The reference pops the candidate
// with the smallest path maximum first; equal maxima mean the same bottleneck node, and the order is decided below it
// (fr_before).  Every thread calls (barriers inside).
__device__ void fr_resolve_goals(const Frontier& F, const Search& S, int tid, int lane, int wave) {
    volatile lds_u32* sh = F.sh;
    uint32_t n = sh[FR_GOAL_N];
    n = n < (uint32_t)FR_GOAL_CAP ? n : (uint32_t)FR_GOAL_CAP;
    if (n == 0u) return;  // (uniform)
    const double inf = __longlong_as_double(0x7FF0000000000000LL);
    if (tid == 0) sh_st_d(sh, FR_ROUND_B1, inf);
    __syncthreads();
    double b1 = inf;
    uint32_t id = 0;
    if ((uint32_t)tid < n) {
        id = F.goal_list[tid];
        const d2 v = (id - 1u) < S.NL ? (d2)S.ln[4 * (size_t)(id - 1u) + 2] : ((const d2*)(S.gn + (id - 1u)))[2];
        b1 = v.x;
    }
    for (uint32_t e = (uint32_t)tid + blockDim.x; e < n; e += blockDim.x) {  // (more candidates than threads: keep this thread's best)
        const uint32_t id2 = F.goal_list[e];
        const d2 v = (id2 - 1u) < S.NL ? (d2)S.ln[4 * (size_t)(id2 - 1u) + 2] : ((const d2*)(S.gn + (id2 - 1u)))[2];
        if (v.x < b1) {
            b1 = v.x;
            id = id2;
        }
    }
    if (id) sh_min_d(sh, FR_ROUND_B1, b1);
    __syncthreads();
    const double rb = sh_ld_d(sh, FR_ROUND_B1);
    // the candidates that share the smallest maximum go to the front of the list (usually one)
    __syncthreads();
    if (tid == 0) sh[FR_GOAL_N] = 0;
    __syncthreads();
    uint32_t mine[2] = {0, 0};
    int nm = 0;
    for (uint32_t e = (uint32_t)tid; e < n; e += blockDim.x) {
        const uint32_t id2 = F.goal_list[e];
        const d2 v = (id2 - 1u) < S.NL ? (d2)S.ln[4 * (size_t)(id2 - 1u) + 2] : ((const d2*)(S.gn + (id2 - 1u)))[2];
        if (v.x == rb) {
            if (nm < 2)
                mine[nm++] = id2;
            else
                atomicOr((uint32_t*)&sh[FR_FLAGS], FRF_TIE);  // (more finalists than this thread can carry: let the exact search decide)
        }
    }
    __syncthreads();  // (everybody has read the list)
    for (int q = 0; q < 2; ++q) {
        const bool have = q < nm;
        const unsigned long long bm = __ballot(have);
        if (bm) {
            const uint32_t base = sh_add_uniform(sh, FR_GOAL_N, (uint32_t)__builtin_popcountll(bm), lane);
            if (have) F.goal_list[base + lane_rank(bm, lane)] = mine[q];
        }
    }
    __syncthreads();
    if (wave == 0) {  // uniform scalar code over the (few) finalists
        const uint32_t nf = sh[FR_GOAL_N];
        uint32_t best = sh[FR_BEST_ID];
        double bb = best ? sh_ld_d(sh, FR_BEST_B1) : inf;
        bool tie = false;
        for (uint32_t q = 0; q < nf; ++q) {
            const uint32_t x = uni_u(F.goal_list[q]);
            bool take = false;
            if (!best || rb < bb) {
                take = true;
            } else if (rb == bb) {
                const int r = fr_before(S, F.gkey, x, best);
                take = r < 0;
                tie = tie || r == 0;
            }
            if (take) {
                best = x;
                bb = rb;
            }
        }
        if (lane == 0) {
            sh[FR_BEST_ID] = best;
            sh_st_d(sh, FR_BEST_B1, bb);
            sh[FR_GOAL_N] = 0;
            if (tie) sh[FR_FLAGS] = sh[FR_FLAGS] | FRF_TIE;
        }
    }
    __syncthreads();
}

// Children of one expansion join the open set: near or far by key.  (They never join the running round: the reference pops
// the smallest open key next, and a round that also swallowed everything its own nodes generate would walk whole subtrees
// the reference leaves as soon as it reaches the horizon.)  Whole wave calls.
// BULK: most lanes carry a child (the pass over the children helper workgroups created): the lists' key ranges are folded per
// wave, not per lane.
template <bool BULK = false>
__device__ __forceinline__ void fr_push_children(const Frontier& F, bool active, uint32_t i0, double f, int lane) {
    const double l_far = sh_ld_d(F.sh, FR_L_FAR);
    const int cls = active ? (f > l_far ? 2 : 1) : -1;
    const unsigned long long b1 = __ballot(cls == 1);
    if (b1) {
        const uint32_t base = sh_add_uniform(F.sh, FR_NEAR_N, (uint32_t)__builtin_popcountll(b1), lane);
        if (cls == 1) {
            const uint32_t pos = base + lane_rank(b1, lane);
            F.near_key[pos] = f;
            F.near_id[pos] = i0 + 1u;
            if (!BULK) {
                sh_min_d(F.sh, FR_NEAR_MIN, f);
                sh_max_d(F.sh, FR_NEAR_MAX, f);
            }
        }
        if (BULK) sh_minmax_wave(F.sh, FR_NEAR_MIN, FR_NEAR_MAX, cls == 1 ? f : __longlong_as_double(0x7FF0000000000000LL), cls == 1 ? f : 0.0, lane);
    }
    const unsigned long long b2 = __ballot(cls == 2);
    if (b2) {
        const uint32_t base = sh_add_uniform(F.sh, FR_FAR_N, (uint32_t)__builtin_popcountll(b2), lane);
        if (cls == 2) {
            const uint32_t pos = base + lane_rank(b2, lane);
            F.far_key[pos] = f;
            F.far_id[pos] = i0 + 1u;
            if (!BULK) {
                sh_min_d(F.sh, FR_FAR_MIN, f);
                sh_max_d(F.sh, FR_FAR_MAX, f);
            }
        }
        if (BULK) sh_minmax_wave(F.sh, FR_FAR_MIN, FR_FAR_MAX, cls == 2 ? f : __longlong_as_double(0x7FF0000000000000LL), cls == 2 ? f : 0.0, lane);
    }
}

// Does the edge into node i0 (0-based, one node per lane) cross the areas of the predecessors in `arr`?  The arithmetic of
// interx_check restricted to those polygons (InterX.m:63-76): the edge's area is transformed once, every polygon segment goes
// through interx_segment.
__device__ bool fr_node_hits_areas(const Search& S, const CheckCtx& C, const SpecCtx& P, uint32_t i0, unsigned long long arr) {
    const NodeRec cn = node_load(S, i0);
    if (!cn.parent) return false;
    const NodeRec pn = node_load(S, cn.parent - 1);
    const int m = NODE_MAN(cn.packed), ncols = NODE_COLS(cn.packed), k = NODE_K(cn.packed);
    const double c = pn.cs, s = pn.sn, pX = pn.x, pY = pn.y;
    const size_t abase = (size_t)m * 3 * PDMPC_VMAX;
    const lds_d2* polys = P.l_soup + P.l_soff[k - 1] + P.l_lit[k - 1];
    bool hit = false;
    // the area's edges in two parts (0 .. H-1 and H .. VMAX-2): one node per lane means the points live in registers, and all
    // VMAX of them at once would push the whole kernel into spilling
    constexpr int H = PDMPC_VMAX / 2;
    auto part = [&](auto np_tag, int first, int ne) {
        constexpr int NP = decltype(np_tag)::value;
        d2 pt[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {  // (columns beyond ncols are padding: transformed, never used)
            const d2 a = C.areas_in_lds ? (d2)C.l_area[abase + first + i] : C.g_area[abase + first + i];
            pt[i].x = c * a.x - s * a.y + pX;  // GraphSearch.m:158
            pt[i].y = s * a.x + c * a.y + pY;  // :159
        }
        unsigned long long rem = arr;
        while (rem) {
            const int p = (int)__builtin_ctzll(rem);
            rem &= rem - 1;
            const lds_d2* poly = polys + p * PDMPC_VMAX;
            d2 q0 = poly[0];
#pragma unroll 1
            for (int j = 0; j + 1 < PDMPC_VMAX; ++j) {
                const d2 q1 = poly[j + 1];
                hit = hit || interx_segment_n<NP>(pt, ne, q0, q1);
                q0 = q1;
            }
        }
    };
    part(std::integral_constant<int, H + 1>{}, 0, ncols - 1 < H ? ncols - 1 : H);
    if (ncols - 1 > H) part(std::integral_constant<int, PDMPC_VMAX - H>{}, H, ncols - 1 - H);
    return hit;
}

// Phase B: position of every node relative to the goal path P_0..P_Hp (goal == 0: exhausted search, every generated node
// was popped).  Nodes are visited in index order, a chunk of blockDim nodes at a time (a parent's index is smaller than
// its children's); a node's (d, b) follows from its parent's.  Results: ref_ids[j] = id of P_j in the reference's tree,
// n_popped, n_expanded.  Sets FRF_TIE if a comparison that matters is an equality.
struct PhaseB {
    uint32_t n_popped, n_expanded;
};
#define PB_ALIVE 0x100u
#define PB_ONPATH 0x200u
template <int NW>
__device__ PhaseB fr_phase_b(const KernelArgs& A, Ctx& X, const Frontier& F, const ExpandEnv& EE, uint32_t goal, lds_u32* ref_ids, LDS_AS unsigned char* scratch,
                             double* st_b, uint32_t* st_d, const lds_u32* gp_path) {
    int tid = X.tid;
    asm volatile("" : "+v"(tid));  // (opaque: what the compiler derives from the thread index here — a few LDS addresses — would otherwise be computed in front of the caller's round loop and kept, or spilled, across all of it)
    const int Hp = X.Hp;
    const Search& S = X.S;
    const VState& VS = X.VS;
    lds_u32* l_path = X.l_path;
    lds_f64* Mp = (lds_f64*)F.hist;                                        // [HP_MAX + 1][HP_MAX + 2]
    lds_f64* pk = Mp + (PDMPC_HP_MAX + 1) * (PDMPC_HP_MAX + 2);            // [HP_MAX + 1] keys of the path nodes
    lds_u32* cnt_pop = (lds_u32*)(pk + PDMPC_HP_MAX + 1);                  // [HP_MAX + 2]
    lds_u32* cnt_ch = cnt_pop + PDMPC_HP_MAX + 2;                          // [HP_MAX + 2]
    lds_u32* ch_d = (lds_u32*)scratch;                                     // [blockDim] state of the chunk's nodes
    lds_f64* ch_b = (lds_f64*)(scratch + 4 * (size_t)blockDim.x);          // [blockDim]
    const int n = EE.n, nw = NW > 0 ? NW : EE.nw;
    lds_u32* pkw = cnt_ch + PDMPC_HP_MAX + 2;                              // [HP_MAX + 1] packed words of the path nodes
    ch_d[tid] = 0;
#ifdef PDMPC_PB_INVALIDATE_L1
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (experiment: phase B reads nothing from this CU's L1 that was cached before)
#endif
    // G's path: the selection's relevance tables hold it when G was the best candidate at the last round boundary (the usual
    // end of a search) — then its keys and packed words are one parallel load; else walked from G
    const bool have_path = goal && F.sh[FR_PATH_FOR] == goal;
    if (goal && !have_path && tid == 0) {
        uint32_t nd = goal;
        for (int i = Hp; i >= 0; --i) {
            l_path[i] = nd;
            nd = node_parent(S, nd - 1);
        }
    }
    if (have_path && tid <= Hp) l_path[tid] = gp_path[tid];
    if (tid < PDMPC_HP_MAX + 2) {
        cnt_pop[tid] = 0;
        cnt_ch[tid] = 0;
    }
    __syncthreads();
    if (goal && tid <= Hp) {
        const uint32_t nd = l_path[tid];
        pk[tid] = F.gkey[nd - 1];
        pkw[tid] = ((const uint32_t*)(S.gn + (nd - 1u)))[15];
    }
    __syncthreads();
    if (goal && tid <= Hp) {
        double m = -1.0;
        for (int j = tid + 1; j <= Hp; ++j) {
            m = pk[j] > m ? pk[j] : m;
            Mp[tid * (PDMPC_HP_MAX + 2) + j] = m;
        }
    }
    __syncthreads();
    uint32_t N = F.sh[FR_NNODES];
    N = N < S.max_nodes ? N : S.max_nodes;
    for (uint32_t base = 0; base < N; base += blockDim.x) {
        const uint32_t i = base + (uint32_t)tid;
        const bool in = i < N;
        uint32_t par = 0, pk_ = 0, vst = 0;
        double key = 0.0;
        if (in) {
            // (keys and links are compact arrays: a pass over 1024 nodes touches 256 lines, not the 1024 lines of their records —
            // a single CU keeps only so many misses in flight)
            const uint64_t u = F.glink[i];
            par = (uint32_t)(u & 0xffffffffull);
            pk_ = (uint32_t)(u >> 32);
            key = F.gkey[i];
            vst = vs_load(VS, i);
        }
        const int depth = NODE_K(pk_);
        bool resolved = !in;
        int guard = 0;
        uint32_t my_d = 0;
        double my_b = -1.0;
        for (;;) {
            if (!resolved) {
                if (!par) {  // the root
                    my_d = PB_ALIVE | (goal ? PB_ONPATH : 0u);
                    resolved = true;
                } else {
                    const uint32_t pi = par - 1u;
                    bool have = false;
                    uint32_t pd = 0;
                    double pb = -1.0;
                    if (pi < base) {
                        pd = st_d[pi];
                        pb = st_b[pi];
                        have = true;
                    } else if (ch_d[pi - base] & 0x80000000u) {
                        pd = ch_d[pi - base] & 0x7fffffffu;
                        pb = ch_b[pi - base];
                        have = true;
                    }
                    if (have) {
                        // generated <=> the parent was generated, its edge is collision-free, and it was expanded
                        const bool alive = (pd & PB_ALIVE) && vs_load(VS, pi) == VS_VALID;
                        if (!alive) {
                            my_d = 0;
                        } else if (!goal) {
                            my_d = PB_ALIVE;
                        } else if ((pd & PB_ONPATH) && l_path[depth] == i + 1u) {
                            my_d = PB_ALIVE | PB_ONPATH | (uint32_t)depth;
                        } else if (pd & PB_ONPATH) {
                            my_d = PB_ALIVE | (uint32_t)(depth - 1);
                            my_b = key;
                        } else {
                            my_d = PB_ALIVE | (pd & 0xffu);
                            my_b = pb > key ? pb : key;
                        }
                        resolved = true;
                    }
                }
                if (resolved) {
                    ch_b[tid] = my_b;
                    ch_d[tid] = my_d | 0x80000000u;  // (after the value it announces: LDS keeps a wave's accesses in order)
                }
            }
            {  // all resolved?  (a shared word and two barriers; __syncthreads_and wants the linear thread index — y, z, the block's
               // dimensions from the dispatch packet —, which then stay alive across the caller's round loop)
               // TWO words, taken in turn: thread 0 starts the next pass's word while slower wavefronts may not have read this pass's
               // yet (with one word a wavefront that was held up behind the second barrier — two workgroups on a CU — read the 1 of
               // the NEXT pass, left the loop alone and took the workgroup's barriers apart)
                const int sw = FR_SLOWEST + (guard & 1);
                if (tid == 0) F.sh[sw] = 1u;
                __syncthreads();
                if (!resolved) F.sh[sw] = 0u;
                __syncthreads();
                if (F.sh[sw] != 0u) break;  // (uniform)
            }
            if (++guard > PDMPC_HP_MAX + 4) {  // (cannot happen: a chain inside a chunk is at most Hp long)
                if (tid == 0) atomicOr((uint32_t*)&F.sh[FR_FLAGS], FRF_BUG);
                break;
            }
        }
        ch_d[tid] = 0;  // (nobody reads this chunk's states any more)
        if (in) {  // for the children in later chunks (stored here, not where the state is found: a barrier waits for the stores in flight)
            st_d[i] = my_d;
            st_b[i] = my_b;
        }
        if (in && (my_d & PB_ALIVE)) {
            int t;
            if (!goal) {
                t = 0;
            } else if (my_d & PB_ONPATH) {
                t = depth + 1;
            } else {
                const int d = (int)(my_d & 0xffu);
                t = Hp + 1;
                for (int j = d + 1; j <= Hp; ++j) {
                    const double m = Mp[d * (PDMPC_HP_MAX + 2) + j];
                    if (my_b == m) atomicOr((uint32_t*)&F.sh[FR_FLAGS], FRF_TIE);
                    if (my_b < m) {
                        t = j;
                        break;
                    }
                }
            }
            if (t <= Hp) {
                if (vst == VS_UNKNOWN) atomicOr((uint32_t*)&F.sh[FR_FLAGS], FRF_BUG);  // a node the reference pops was never processed
                atomicAdd((uint32_t*)&cnt_pop[t], 1u);
                if (vst == VS_VALID && depth < Hp) {
                    const lds_mask64* mrow = EE.l_mask + ((size_t)depth * n + (NODE_TRIM(pk_) - 1)) * nw;
                    uint32_t c = 0;
                    for (int w = 0; w < nw; ++w) c += (uint32_t)__builtin_popcountll(mrow[w]);
                    atomicAdd((uint32_t*)&cnt_ch[t], c);
                }
            }
        }
        __syncthreads();  // the chunk's LDS state is rewritten by the next chunk
    }
    __syncthreads();
    PhaseB R;
    R.n_popped = 0;
    R.n_expanded = 1;
    if (!goal) {
        R.n_popped = cnt_pop[0];
        R.n_expanded = 1u + cnt_ch[0];
        return R;
    }
    if (tid == 0) {
        uint32_t s = 1;  // S_j = tree size when P_j is popped
        ref_ids[0] = 1;
        for (int j = 0; j < Hp; ++j) {
            s += cnt_ch[j];  // nodes expanded before P_j: t <= j
            const uint32_t ppk = pkw[j], cpk2 = pkw[j + 1];
            const lds_mask64* mrow = EE.l_mask + ((size_t)NODE_K(ppk) * n + (NODE_TRIM(ppk) - 1)) * nw;
            const int t2 = NODE_TRIM(cpk2) - 1;  // 0-based successor trim
            uint32_t rank = 0;
            for (int w = 0; w < nw; ++w) {
                const uint64_t m = mrow[w];
                if (w < t2 / 64) rank += (uint32_t)__builtin_popcountll(m);
                if (w == t2 / 64) rank += (uint32_t)__builtin_popcountll(m & ((1ull << (t2 % 64)) - 1ull));
            }
            ref_ids[j + 1] = s + 1u + rank;
        }
    }
    uint32_t np = 1, ne = 1;
    for (int t = 0; t <= Hp; ++t) {
        np += cnt_pop[t];
        ne += cnt_ch[t];
    }
    R.n_popped = np;
    R.n_expanded = ne;
    __syncthreads();
    return R;
}

}  // namespace
