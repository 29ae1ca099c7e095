// frontier_kernel.hip — the optimal graph search without a pop-ordered open list: every wavefront of the workgroup
// evaluates and expands nodes side by side; the reference's pop order is reconstructed, not executed.
//
// What the reference computes (GraphSearch.m:53-107): pop the open-list entry with the smallest key f = g + h, discard
// it if its edge collides (:75-77), stop if it is at the horizon (:81-90), otherwise expand it (expand_node.m) and push
// the children.  Node ids are positions in the tree, i.e. they count the children created before.
//
// With pairwise distinct keys that process has a closed form.  For nodes X, Y with lowest common ancestor A:
//     X is popped before Y  <=>  X is an ancestor of Y,  or  max key on the path (A, X]  <  max key on the path (A, Y].
// (After A is popped the open list holds both branches; the branch whose largest key is smaller is walked to its end
// before the other branch's largest key can become the minimum.  tools/sigma_order.py checks this against the oracle's
// pop sequences.)  Hence, with G the first valid node at the horizon in that order and B(X) the largest key on the
// path root..X:
//   * the reference pops exactly the nodes that come before G; all of them have B <= B(G);
//   * the search may therefore process nodes in ANY order and in parallel, as long as in the end every generated node
//     with key <= B(G) has been processed.  Phase A below does that in rounds ("process everything with key <= L",
//     L growing by a few dozen entries per round), one node per wavefront at a time: eval_edge_exact (GraphSearch.m:
//     111-196) with the wave-wide InterX / SAT code, then expand_node.m.  No wave waits for another one's pop;
//   * n_popped, n_expanded (= tree size) and the ids along tree_path are counts of nodes that come before the nodes
//     P_0..P_Hp of G's path; with (d, b) = (depth at which X leaves that path, largest key of X's path below it),
//     X comes before P_j (j > d) iff b < max key(P_{d+1..j}).  Phase B evaluates that for every node in one pass.
// Equal keys where the order matters (in a comparison against G's path or between goal candidates) make the result
// depend on the layout of the reference's binary heap: the search is then redone with the libstdc++-faithful heap
// (serial_search.hpp), exactly as the round-1 kernel does after a tied pop.
//
// Predecessors that finish while the search runs (PrioritizedController.m:476-491): their areas enter the soup at the
// next round boundary, every edge found collision-free so far is re-checked against the new areas only, and nodes that
// now collide simply become invalid — their subtrees are ignored by phase B.  Nothing restarts.
//
// Open set: `ready` (LDS, this round), `near` (HBM, the few thousand smallest keys), `far` (HBM, the rest); entries are
// (key, node).  near and far are unordered; a histogram pass picks the key below which a round (or a refill of near)
// takes its entries, a partition pass moves them.  Arithmetic, operation order and -ffp-contract=off are those of the
// serial kernel: every record is bit-identical to the oracle's.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "serial_search.hpp"

#include "frontier_common.hpp"

namespace {

// One node of the round: eval_edge_exact (GraphSearch.m:111-196), the goal test (:81-90), expand_node.m.  Whole wave.
// cu / pu: the node's record and its parent's (the same in every lane).  Returns 0, or the child this wave goes on with right
// away (its record then in cu, this node's in pu): the child with the smallest key if that key is not above l_join, the
// largest key the round has selected.  The reference pops such a child before anything the round left behind, so taking
// it now is no more of a guess than the round's own entries are — and a search that runs straight to the horizon gets there
// in one round instead of one round per level.
template <int CHECKER, int NW>
__device__ __forceinline__ uint32_t fr_process(const KernelArgs& A, Ctx& X, const Frontier& F, const ExpandEnv& EE, uint32_t cur, NodeBits& cu, NodeBits& pu, double l_join, bool known_valid = false, bool have_cs = false) {
    const int lane = X.lane, Hp = X.Hp;
    Search& S = X.S;
    const VState& VS = X.VS;
    const uint32_t c0 = cur - 1u;
    const NodeRec& cn = cu.r;  // same record in every lane
    const bool valid = known_valid || edge_valid_recs<CHECKER>(X.C, cn, pu.r, lane);  // (known_valid: a helper workgroup has checked the edge)
    if (!valid) {
        if (lane == 0) vs_store(VS, c0, VS_INVALID);
        return 0u;
    }
    const uint32_t cpk = uni_u(cn.packed);
    if (NODE_K(cpk) == Hp) {
        if (lane == 0) vs_store(VS, c0, VS_VALID);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        fr_offer_goal(F, S, VS, cur, lane);
        return 0u;
    }
    double sn, cs;
    if (have_cs) {  // (a helper has evaluated them, with the same function: uniform)
        cs = cn.cs;
        sn = cn.sn;
    } else {
        pdmpc_sincos(cn.yaw, &sn, &cs);  // expand_node.m:50-51
    }
    if (lane == 0) node_store_cs(S, c0, cs, sn);
    // reserve the children's node indices
    const int n = EE.n, nw = NW > 0 ? NW : EE.nw;
    const lds_mask64* mrow = EE.l_mask + ((size_t)NODE_K(cpk) * n + (NODE_TRIM(cpk) - 1)) * nw;
    uint32_t total = 0;
    for (int w = 0; w < nw; ++w) total += (uint32_t)__builtin_popcountll(mrow[w]);
    total = uni_u(total);
    const uint32_t base = sh_add_uniform(F.sh, FR_NNODES, total, lane);
    if (base + total > S.max_nodes) {
        if (lane == 0) {
            atomicOr((uint32_t*)&F.sh[FR_FLAGS], FRF_OVERFLOW);
            vs_store(VS, c0, VS_VALID);
        }
        return 0u;
    }
    if (lane == 0) vs_store(VS, c0, VS_VALID);  // before any child can be picked up: a child's path check looks at it
    uint32_t nn = base;
    uint32_t next = 0;
    lds_d2* hand = X.C.sh;  // the chosen child's record travels through the wave's shape scratch (rewritten by the next edge check only)
    (void)expand_children<false, NW>(EE, S, VS, cur, cn, cs, sn, nn, [&](uint64_t mask, bool active, uint32_t i0, double f, int ccnt, const NodeRec& ch) {
        (void)mask;
        (void)ccnt;
        if (active) {
            F.gkey[i0] = f;
            F.glink[i0] = (unsigned long long)ch.parent | ((unsigned long long)ch.packed << 32);
        }
        // (no fence here: nothing reads the children's records, keys or list entries before the barrier that ends the round)
        bool mine = false;
        const unsigned long long bj = __ballot(active && f <= l_join);
        if (bj && next == 0u) {  // (uniform) the smallest of those keys; equal keys: the lower lane
            double m = (active && f <= l_join) ? f : __longlong_as_double(0x7FF0000000000000LL);
#pragma unroll
            for (int o = PDMPC_WAVE / 2; o > 0; o >>= 1) {
                const double v = __shfl_xor(m, o);
                m = v < m ? v : m;
            }
            const unsigned long long bw = __ballot(active && f == m);
            const int wl = (int)__builtin_ctzll(bw);
            mine = lane == wl;
            if (mine) {
                NodeBits u;
                u.r = ch;
#pragma unroll
                for (int q = 0; q < 4; ++q) hand[q] = u.q[q];
            }
            next = lane_u(i0, wl) + 1u;
        }
        fr_push_children(F, active && !mine, i0, f, lane);
    });
    if (next) {
        wave_sync();
        pu = cu;
        pu.r.cs = cs;
        pu.r.sn = sn;
#pragma unroll
        for (int q = 0; q < 4; ++q) cu.q[q] = hand[q];
    }
    return next;
}


// The search.  Returns true (to every wave) if a tie was met and the search has to be redone on the binary heap.
template <int CHECKER, int NW>
__device__ __forceinline__ bool frontier_search(const KernelArgs& A, Ctx& X, lds_u32* ref_ids) {
    const int tid_k = X.tid, lane_k = X.lane, wave = X.wave, slot = X.slot, Hp = X.Hp;
    int tid = tid_k, lane = lane_k;
    volatile lds_u32* sh = X.l_shared;
    Search& S = X.S;
    const VState& VS = X.VS;
    const SpecCtx& P = X.P;
    const DevVehicle* __restrict__ V = X.V;
    const int n_waves = (int)(blockDim.x >> 6);
    const double inf = __longlong_as_double(0x7FF0000000000000LL);
    const size_t voff = (size_t)slot * A.max_nodes;

    Frontier F;
    F.sh = sh;
    F.ready = (lds_u32*)(X.lsm + A.lds.heap_key);
    F.hist = F.ready + 2048;
    F.goal_list = F.hist;
    lds_u32* gp_path = F.ready + FR_READY_CAP;            // [HP_MAX + 1] path of the best goal candidate (relevance test)
    lds_f64* gp_mp = (lds_f64*)(F.ready + FR_READY_CAP + 32);  // [HP_MAX + 1] largest key of that path below depth d
    F.near_key = A.arena.near_key + voff;
    F.near_id = A.arena.near_id + voff;
    F.far_key = A.arena.pop_log + voff;
    F.far_id = A.arena.heap_id + voff;
    F.gkey = S.gkey;
    F.glink = A.arena.link + voff;
    F.n_waves = n_waves;
    volatile lds_u32* wsum = (volatile lds_u32*)(F.hist + FR_NBINS);  // [32] per-wave counts of fr_partition

    // per-wave expansion scratch lives in the wave's candidate list (an edge check is over before its node is expanded)
    ExpandEnv EE;
    EE.l_mask = X.l_mask;
    EE.l_mi = X.l_mi;
    EE.l_pose = X.l_pose;
    EE.l_rx = X.l_rx;
    EE.l_ry = X.l_ry;
    EE.l_dcum = X.l_dcum;
    EE.l_term = (lds_f64*)X.C.cand;
    EE.l_chxy = (lds_d2*)(EE.l_term + 16 * PDMPC_HP_MAX);
    EE.Hp = Hp;
    EE.n = X.n;
    EE.nw = X.nw;
    EE.lane = lane;

    // ---- root node (GraphSearch.m:34-46): the first round
    if (tid == 0) {
        NodeRec r;
        r.x = V->x0;
        r.y = V->y0;
        r.yaw = V->yaw0;
        r.g = 0.0;
        r.cs = 0.0;
        r.sn = 0.0;
        r.h = 0.0;
        r.parent = 0;
        r.packed = (uint32_t)V->trim0;
        node_store(S, 0, r);
        F.gkey[0] = 0.0;
        F.glink[0] = (unsigned long long)r.parent | ((unsigned long long)r.packed << 32);
        vs_store(VS, 0, VS_UNKNOWN);
        for (int w = FR_NNODES; w < SH_WORDS; ++w) sh[w] = 0;
        sh[FR_NNODES] = 1;
        sh[FR_RD_TAIL] = 1;
        sh[FR_VLIST_N] = 0;
        sh_st_d(sh, FR_JOIN_MAX, A.fr_root_dive ? inf : 0.0);  // the root's round: its own key, or (fr_root_dive) no limit — the wave that expands the root follows the best children to the horizon: on a free road that is the goal, in the first round
        sh_st_d(sh, FR_NEAR_MIN, inf);
        sh_st_d(sh, FR_FAR_MIN, inf);
        sh_st_d(sh, FR_L_FAR, inf);
        sh[SH_NNODES] = 1;
        if (A.debug_tail == 2) sh[FR_EVER_INVAL] = 1;  // (debugging: exercise the ancestor check without arrivals)
    }
    for (int i = tid; i < FR_READY_CAP; i += (int)blockDim.x) F.ready[i] = i == 0 ? 1u : 0u;
    __syncthreads();

    FR_PROGRESS(10)
    int status = PDMPC_OK;
    bool dep_timeout = X.dep_timeout;
    uint32_t goal = 0;
    uint32_t idle_polls = 0;
    // where the time goes (100 MHz ticks, reported in the record's spare rows with PDMPC_DEBUG_TAIL=1)
    unsigned long long tk_work = 0, tk_arrival = 0, tk_select = 0, tk_wait = 0, tk_mark = __builtin_amdgcn_s_memrealtime();
    const unsigned long long tk_start = tk_mark;
#define FR_TICK(acc)                                                       \
    {                                                                      \
        const unsigned long long now__ = __builtin_amdgcn_s_memrealtime(); \
        acc += now__ - tk_mark;                                            \
        tk_mark = now__;                                                   \
    }
    // appends (k, i) of the lanes with `take` to far (whole wave calls); the keys' range is folded into FR_FAR_MIN/MAX by flush_far
    double far_mn = inf, far_mx = 0.0, near_mn = inf, near_mx = 0.0, sel_mx = 0.0;
    auto to_far = [&](bool take, double k, uint32_t i) {
        const unsigned long long b = __ballot(take);
        if (b) {
            const uint32_t base = sh_add_uniform(sh, FR_FAR_N, (uint32_t)__builtin_popcountll(b), lane);
            if (take) {
                const uint32_t pos = base + lane_rank(b, lane);
                F.far_key[pos] = k;
                F.far_id[pos] = i;
                far_mn = k < far_mn ? k : far_mn;
                far_mx = k > far_mx ? k : far_mx;
            }
        }
    };
    // (after the pass that called to_far; whole wave)
    auto flush_far = [&]() {
        sh_minmax_wave(sh, FR_FAR_MIN, FR_FAR_MAX, far_mn, far_mx, lane);
        far_mn = inf;
        far_mx = 0.0;
    };
    const lds_d2* stage = (const lds_d2*)(X.lsm + A.lds.stage);  // [2 * fr_stage_cap] records: node, parent
    uint32_t n_staged = 0;  // ready entries 0 .. n_staged - 1 have their records staged
    PhaseB R;               // phase B's result (valid while pb_valid)
    R.n_popped = 0;
    R.n_expanded = 0;
    bool pb_valid = false, far_clobbered = false;
    // shared rounds (helper workgroups)
    unsigned long long* board = A.help_board + (size_t)slot * PDMPC_HB_WORDS;
    uint32_t* hlist = A.help_list + (size_t)slot * PDMPC_HELP_CAP;
    const uint32_t* hverdict = A.help_verdict + (size_t)slot * PDMPC_HELP_CAP;
    const d2* hcs = (const d2*)A.help_cs + (size_t)slot * PDMPC_HELP_CAP;
    uint32_t help_seq = 0;  // rounds shared so far (same value in every thread)
    for (;;) {
        // (the thread's index is made opaque once per round: the per-thread addresses the compiler derives from it would otherwise stay
        // in registers across the whole loop; see bulk_kernel.hip / DESIGN.md section 3.4)
        tid = tid_k;
        lane = lane_k;
        asm volatile("" : "+v"(tid), "+v"(lane));
        // ================= a round: every wave takes nodes off the ready list until the list is empty =================
        // children up to the largest key the round took may be taken along (fr_process), but none that comes after the best goal candidate
        double l_join = sh_ld_d(sh, FR_JOIN_MAX);
        if (sh[FR_BEST_ID] != 0u) {
            const double bb1 = sh_ld_d(sh, FR_BEST_B1);
            l_join = bb1 > 0.0 ? (l_join < bb1 ? l_join : __longlong_as_double(__double_as_longlong(bb1) - 1)) : 0.0;  // (strictly below it)
        }
        if (sh[FR_RD_TAIL] > (uint32_t)A.fr_dive) l_join = -1.0;  // (a large round keeps every wave busy as it is; chains only delay its end)
        uint32_t chained = 0;  // nodes this wave processed beyond the round's entries
        // entries RD_HEAD .. RD_TAIL of the ready list, one LDS ticket each; entries st0 .. st0 + stn - 1 have their records in the
        // staging area; checked: helper workgroups have found their edges collision-free
        auto run_ready = [&](uint32_t st0, uint32_t stn, bool checked) {
            for (;;) {
                const uint32_t t = sh_add_uniform(sh, FR_RD_HEAD, 1u, lane);
                if (t >= uni_u(sh[FR_RD_TAIL])) break;
                uint32_t cur = uni_u(F.ready[t]);
                bool known_valid = checked;
                bool have_cs = false;
                // the node's record and its parent's: staged in LDS when the round was selected (one HBM latency per round instead of
                // two dependent ones per node), else through L2
                NodeBits cu, pu;
                if (t - st0 < stn) {  // (unsigned: t < st0 is far outside)
                    have_cs = checked;  // (staged with the helper's cos / sin in the record's own slot)
                    const lds_d2* staged = stage + 8 * (size_t)(t - st0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        cu.q[q] = staged[q];
                        pu.q[q] = staged[4 + q];
                    }
                } else {
                    cu.r = node_load(S, cur - 1u);
                    const uint32_t par = uni_u(cu.r.parent);
                    pu.r = node_load(S, par ? par - 1u : 0u);
                }
                const unsigned long long tp0 = __builtin_amdgcn_s_memrealtime();
                for (;;) {  // the node, then the chain of best children the round's key range covers (fr_process)
                    const uint32_t next = fr_process<CHECKER, NW>(A, X, F, EE, cur, cu, pu, l_join, known_valid, have_cs);
                    if (!next) break;
                    cur = next;
                    known_valid = false;
                    have_cs = false;
                    ++chained;
                }
                if (A.debug_tail && lane == 0) {  // the slowest single entry of this search (ticks << 32 | node)
                    const unsigned long long dtp = __builtin_amdgcn_s_memrealtime() - tp0;
                    __hip_atomic_fetch_max((lds_u64s*)(sh + FR_SLOWEST), (dtp << 32) | cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        };
        // A large round is shared with the helper workgroups (CUs this launch leaves idle): the owner keeps the front of the
        // ready list, the rest is posted; helpers claim entries from its start, check the edges (the expensive half of a node,
        // and a pure function of the tree and the soups) and leave verdicts; the owner takes what nobody has claimed when it is
        // through with its own part, waits for the claimed entries and expands the collision-free ones.
        const uint32_t tail = sh[FR_RD_TAIL];
        if (tail) pb_valid = false;  // (the tree grows: phase B's result is stale)
        uint32_t n_own = tail, n_sh = 0;
        const bool hx = A.help_expand != 0 && sh[FR_NNODES] >= S.NL;  // helpers also expand (the round's new nodes then lie beyond the LDS copies)
        if (A.n_helpers > 0 && tail >= (uint32_t)A.fr_share_min && P.n_pred <= 64) {
            n_own = tail / (uint32_t)A.fr_own_div > 2u * (uint32_t)n_waves ? tail / (uint32_t)A.fr_own_div : 2u * (uint32_t)n_waves;
            if (hx) n_own = 0;  // (node indices come from the board while helpers expand: the owner creates no nodes meanwhile)
            n_sh = tail - n_own;
        }
        if (n_sh) {  // (uniform)
            if (A.fr_dive < 2048) l_join = -1.0;  // no chains in a shared round: their edge checks would be the owner's, the next round's are the helpers'
            ++help_seq;
            for (uint32_t e = (uint32_t)tid; e < n_sh; e += blockDim.x) hlist[e] = F.ready[n_own + e];
            // every wave's stores so far — the list, the records of the tree — must have reached L2 before thread 0 writes L2 back: a
            // workgroup barrier alone does not wait for them (one CU, one L1: it has no need to)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const unsigned long long all = P.n_pred >= 64 ? ~0ull : ((1ull << P.n_pred) - 1ull);
                __hip_atomic_store(board + PDMPC_HB_N, (unsigned long long)n_sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(board + PDMPC_HB_MASK, all & ~sh_load64(sh, SH_PEND_LO), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(board + PDMPC_HB_DONE, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(board + PDMPC_HB_NNODES, (unsigned long long)sh[FR_NNODES], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(board + PDMPC_HB_FLAGS, hx ? 2ull : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // bit 1: expand what you find collision-free
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_store(board + PDMPC_HB_TICKET, ((unsigned long long)help_seq << 32) | ((unsigned long long)n_sh << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sh[FR_RD_TAIL] = n_own;
            }
            __syncthreads();
        }
        // Phase 0: the owner's part (or the whole round).  Then: close the shared part (what no helper has claimed is the owner's:
        // the remainder) and collect what the helpers did.  Helpers that only check: remainder in phase 1 (while the helpers finish),
        // collect + expand the collision-free entries in phase 2.  Helpers that expand: close + collect in phase 1 (the node counter
        // comes back from the board; left to do are the goal tests of the collision-free entries at the horizon), remainder in
        // phase 2.  (One copy of the processing code for all phases: not unrolled.)
        const int n_phases = n_sh ? 3 : 1;
        const int collect_phase = hx ? 1 : 2, remainder_phase = hx ? 2 : 1;
        uint32_t n_chk = 0;      // collected entries whose records are staged
        uint32_t closed_at = 0;  // shared entries the helpers claimed
        const uint32_t nn_post = sh[FR_NNODES];  // (no node is created between here and the posting)
#pragma unroll 1
        for (int phase = 0; phase < n_phases; ++phase) {
            if (phase == 1) {
                __syncthreads();
                if (tid == 0) {  // close the shared part
                    unsigned long long cur = __hip_atomic_load(board + PDMPC_HB_TICKET, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    uint32_t closed = n_sh;
                    if (hx) {  // (nothing of its own to do meanwhile: the owner gives the helpers time to claim, and closes when the claims stall)
                        uint32_t last = (uint32_t)(cur & 0xffffull), stall = 0;
                        while (last < n_sh && stall < (uint32_t)A.help_patience) {
                            __builtin_amdgcn_s_sleep(8);
                            cur = __hip_atomic_load(board + PDMPC_HB_TICKET, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const uint32_t idx = (uint32_t)(cur & 0xffffull);
                            stall = idx == last ? stall + 1u : 0u;
                            last = idx;
                        }
                    }
                    for (;;) {
                        const uint32_t idx = (uint32_t)(cur & 0xffffull);
                        if (idx >= n_sh) break;
                        if (__hip_atomic_compare_exchange_strong(board + PDMPC_HB_TICKET, &cur, ((unsigned long long)help_seq << 32) | ((unsigned long long)n_sh << 16) | n_sh, __ATOMIC_RELAXED,
                                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            closed = idx;
                            break;
                        }
                    }
                    sh[FR_HELP_CLOSED] = closed;
                }
                __syncthreads();
                closed_at = sh[FR_HELP_CLOSED];
            }
            if (phase >= 1 && phase == remainder_phase) {
                __syncthreads();
                if (tid == 0) {
                    sh[FR_RD_HEAD] = n_own + closed_at;
                    sh[FR_RD_TAIL] = tail;
                }
                __syncthreads();
            }
            if (phase >= 1 && phase == collect_phase) {
                __syncthreads();
                const uint32_t closed = closed_at;
                if (tid == 0) {
                    if (closed) {  // wait for the helpers' part
                        uint32_t spins = 0;
                        while (__hip_atomic_load(board + PDMPC_HB_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)closed) {
                            __builtin_amdgcn_s_sleep(4);
                            if (++spins > A.spin_limit) {
                                atomicOr((uint32_t*)&sh[FR_FLAGS], FRF_BUG);  // reported as an error status: must never happen
                                break;
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                    sh[FR_VLIST_N] = 0;
                    if (hx) {  // the node counter comes back; the children the helpers created wait in the child buffer
                        const unsigned long long nn = __hip_atomic_load(board + PDMPC_HB_NNODES, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long fl = __hip_atomic_load(board + PDMPC_HB_FLAGS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sh[FR_NNODES] = (uint32_t)nn;
                        if (fl & 1ull) atomicOr((uint32_t*)&sh[FR_FLAGS], FRF_OVERFLOW);
                    }
                }
                __syncthreads();
                // The verdicts in one pass (a load per entry in the processing loop would be a round trip to memory each): 2 collides,
                // 1 collision-free (expanded, if helpers expand), 3 collision-free at the horizon, 4 collision-free but the arena is full.
                // What is left to do for an entry is packed to the front of the shared part and its records are staged.
                const bool ok = !(sh[FR_FLAGS] & FRF_BUG);
                if (tid == 0) {
                    atomicAdd(A.work_count + 4, 1ull);
                    atomicAdd(A.work_count + 5, (unsigned long long)closed);
                    sh[FR_HELP_CLOSED] = 0;  // now: entries packed so far
                }
                __syncthreads();
                for (uint32_t base = 0; base < closed && ok; base += blockDim.x) {  // (uniform trip count)
                    const uint32_t e = base + (uint32_t)tid;
                    const bool in = e < closed;
                    const uint32_t v = in ? hverdict[e] : 0u;
                    const uint32_t id = in ? F.ready[n_own + e] : 0u;
                    if (in && v == 2u) vs_store(VS, id - 1u, VS_INVALID);
                    if (in && (v < 1u || v > 4u)) atomicOr((uint32_t*)&sh[FR_FLAGS], FRF_BUG);  // (claimed, reported finished, and no verdict)
                    if (in && v == 4u) atomicOr((uint32_t*)&sh[FR_FLAGS], FRF_OVERFLOW);
                    if (in && hx && (v == 1u || v == 4u)) {  // expanded by a helper (4: found collision-free, then the arena was full)
                        vs_store(VS, id - 1u, VS_VALID);
                        if (v == 1u && id - 1u < S.NL) S.ln[4 * (size_t)(id - 1u) + 2] = hcs[e];  // (the LDS copy of its record: cos / sin of its yaw)
                    }
                    const bool todo = in && (hx ? v == 3u : (v == 1u || v == 3u));
                    const unsigned long long bv = __ballot(todo);
                    uint32_t pos = 0;
                    if (bv) pos = sh_add_uniform(sh, FR_HELP_CLOSED, (uint32_t)__builtin_popcountll(bv), lane) + lane_rank(bv, lane);
                    __syncthreads();  // every entry of this pass has been read: the packed ones may overwrite them
                    if (todo) {
                        F.ready[n_own + pos] = id;
                        if (pos < (uint32_t)A.fr_stage_cap) ((lds_d2*)(X.lsm + A.lds.stage))[8 * (size_t)pos + 2] = hcs[e];  // the helper's cos / sin of the node's yaw (unused at the horizon)
                    }
                }
                __syncthreads();
                // the children the helpers created (node indices from the counter at posting time up to where it stands now): validity
                // unknown, into the open lists by key (fr_push_children)
                {
                    const uint32_t i_lo = nn_post, i_hi = sh[FR_NNODES] < S.max_nodes ? sh[FR_NNODES] : S.max_nodes;
                    for (uint32_t base = i_lo; base < i_hi && ok && hx; base += blockDim.x) {  // (uniform trip count)
                        const uint32_t i = base + (uint32_t)tid;
                        const bool in = i < i_hi;
                        const double k = in ? F.gkey[i] : 0.0;
                        if (in) vs_store(VS, i, 0);
                        fr_push_children<true>(F, in, in ? i : 0u, k, lane);
                    }
                }
                __syncthreads();
                const uint32_t n_ok = sh[FR_HELP_CLOSED];
                n_chk = n_ok < (uint32_t)A.fr_stage_cap ? n_ok : (uint32_t)A.fr_stage_cap;
                if (n_chk) {  // (uniform)
                    lds_d2* st = (lds_d2*)(X.lsm + A.lds.stage);
                    for (uint32_t w0 = (uint32_t)tid; w0 < n_chk * 4u; w0 += blockDim.x) {
                        const uint32_t e = w0 >> 2, q = w0 & 3u;
                        if (q != 2u) st[8 * (size_t)e + q] = ((const d2*)(S.gn + (F.ready[n_own + e] - 1u)))[q];  // (piece 2 holds the helper's cos / sin)
                    }
                    __syncthreads();
                    for (uint32_t w0 = (uint32_t)tid; w0 < n_chk * 4u; w0 += blockDim.x) {
                        const uint32_t e = w0 >> 2, q = w0 & 3u;
                        const uint32_t par = (uint32_t)((uint64_t)__double_as_longlong(st[8 * (size_t)e + 3].y) & 0xffffffffull);
                        st[8 * (size_t)e + 4 + q] = ((const d2*)(S.gn + (par ? par - 1u : 0u)))[q];
                    }
                }
                if (tid == 0) {
                    sh[FR_VLIST_N] = 0;
                    sh[FR_RD_HEAD] = n_own;
                    sh[FR_RD_TAIL] = n_own + n_ok;
                }
                __syncthreads();
            }
            // (one call site: the processing code is instantiated once.  Remainder: the staging area holds the round's first entries;
            // what is left of them there, if anything, goes through L2)
            const bool collected = phase >= 1 && phase == collect_phase;
            const uint32_t st0 = collected ? n_own : 0u, stn = phase == 0 ? n_staged : (collected ? n_chk : 0u);
            run_ready(st0, stn, collected);
        }
        if (n_sh) {
            __syncthreads();
            if (tid == 0) sh[FR_RD_TAIL] = tail;
            __syncthreads();
        }
        (void)sh_add_uniform(sh, FR_PROCESSED, chained + (tid == 0 ? tail : 0u), lane);  // (the round's size is fixed while it runs)
        __syncthreads();
        FR_TICK(tk_work)
        FR_PROGRESS(1)
        fr_resolve_goals(F, S, tid, lane, wave);

        // ================= round boundary (every thread; decisions are uniform) =====================================
        uint32_t flags = sh[FR_FLAGS];
        if (flags & FRF_OVERFLOW) {
            status = PDMPC_ARENA_OVERFLOW;
            break;
        }
        if ((flags & FRF_BUG) || sh[FR_ROUNDS] > A.spin_limit) {  // watchdog: reported as an error status
            dep_timeout = true;
            status = PDMPC_EXHAUSTED;
            break;
        }
        // predecessors that finished meanwhile: their areas enter the soup, collision-free edges are re-checked
        if (wave == 0) (void)poll_predecessors(A, P, sh, lane);
        __syncthreads();
        if (sh[SH_STATE] == ST_ARRIVED) {
            const unsigned long long arr = sh_load64(sh, SH_ARR_LO);
            uint32_t nn = sh[FR_NNODES];
            nn = nn < S.max_nodes ? nn : S.max_nodes;
            incorporate_areas(P, arr, tid);
            __syncthreads();
            // Only collision-free nodes can lose their edge: they are gathered first (a third of the tree, scattered), so that
            // the check runs on full wavefronts.  The list lives in the histogram's bins, unused at a round boundary.
            for (uint32_t base0 = 0; base0 < nn; base0 += FR_NBINS) {  // (uniform trip counts: barriers inside)
                const uint32_t end = base0 + FR_NBINS < nn ? base0 + FR_NBINS : nn;
                if (tid == 0) sh[FR_VLIST_N] = 0;
                __syncthreads();
                for (uint32_t b = base0; b < end; b += blockDim.x) {
                    const uint32_t i0 = b + (uint32_t)tid;
                    const bool v = i0 < end && vs_load(VS, i0 < end ? i0 : 0u) == VS_VALID;
                    const unsigned long long bal = __ballot(v);
                    if (bal) {
                        const uint32_t pos0 = sh_add_uniform(sh, FR_VLIST_N, (uint32_t)__builtin_popcountll(bal), lane);
                        if (v) F.hist[pos0 + lane_rank(bal, lane)] = i0;
                    }
                }
                __syncthreads();
                const uint32_t nv = sh[FR_VLIST_N];
                for (uint32_t e = (uint32_t)tid; e < nv; e += blockDim.x) {
                    const uint32_t i0 = F.hist[e];
                    if (fr_node_hits_areas(S, X.C, P, i0, arr)) {
                        vs_store(VS, i0, VS_INVALID);
                        atomicOr((uint32_t*)&sh[FR_FLAGS], FRF_INVALIDATED);
                    }
                }
                __syncthreads();
            }
            flags = sh[FR_FLAGS];
            const bool reopen = (flags & FRF_INVALIDATED) && (sh[FR_DROPPED] != 0u || far_clobbered);  // (far_clobbered: phase B has run over far's arrays)
            if (flags & FRF_INVALIDATED) pb_valid = false;
            if (reopen) far_clobbered = false;
            __syncthreads();
            if (tid == 0) {
                atomicAdd(P.counters + 2, 1);
                const unsigned long long pend = sh_load64(sh, SH_PEND_LO) & ~arr;
                sh[SH_PEND_LO] = (uint32_t)pend;
                sh[SH_PEND_HI] = (uint32_t)(pend >> 32);
                sh[SH_ARR_LO] = 0;
                sh[SH_ARR_HI] = 0;
                sh[SH_STATE] = ST_RUN;
                if (flags & FRF_INVALIDATED) {
                    sh[FR_EVER_INVAL] = 1;
                    sh[FR_BEST_ID] = 0;  // the best candidate may have lost an ancestor: look at all of them again
                    sh[FR_PATH_FOR] = 0;
                    sh[FR_FLAGS] = flags & ~FRF_INVALIDATED;
                }
                if (reopen) {
                    // open entries were dropped because they come after a candidate that may be gone now: rebuild the
                    // open set from the tree (every generated node that was never evaluated is open)
                    sh[FR_NEAR_N] = 0;
                    sh[FR_FAR_N] = 0;
                    sh[FR_DROPPED] = 0;
                    sh_st_d(sh, FR_NEAR_MIN, inf);
                    sh_st_d(sh, FR_NEAR_MAX, 0.0);
                    sh_st_d(sh, FR_FAR_MIN, inf);
                    sh_st_d(sh, FR_FAR_MAX, 0.0);
                    sh_st_d(sh, FR_L_FAR, -1.0);  // (everything goes to far until the next refill)
                }
            }
            __syncthreads();
            if (flags & FRF_INVALIDATED) {
                for (uint32_t base = 0; base < nn; base += blockDim.x) {  // (uniform trip count: barriers inside)
                    const uint32_t b = base + (uint32_t)wave * PDMPC_WAVE;
                    const uint32_t i0 = b + (uint32_t)lane;
                    const bool in = i0 < nn;
                    const uint32_t j0 = in ? i0 : 0u;  // (straight-line code: every lane loads something valid)
                    const uint32_t vst = vs_load(VS, j0);
                    const uint32_t par = node_parent(S, j0);
                    const bool cand = in && vst == VS_VALID && NODE_K(((const uint32_t*)(S.gn + j0))[15]) == Hp;
                    const bool open = reopen && in && vst == VS_UNKNOWN && par != 0u && vs_load(VS, par ? par - 1u : 0u) == VS_VALID;
                    if (reopen) to_far(open, F.gkey[j0], j0 + 1u);
                    unsigned long long bc = __ballot(cand);
                    while (bc) {
                        const int l = __builtin_ctzll(bc);
                        bc &= bc - 1;
                        fr_offer_goal(F, S, VS, b + (uint32_t)l + 1u, lane);
                    }
                    __syncthreads();  // at most blockDim candidates per pass: the list cannot overflow
                    fr_resolve_goals(F, S, tid, lane, wave);
                }
                flush_far();
                __syncthreads();
            }
            flags = sh[FR_FLAGS];
        }
        if (flags & FRF_TIE) return true;
        FR_TICK(tk_arrival)

        // the relevance tables follow the best goal candidate
        const uint32_t best = sh[FR_BEST_ID];
        if (best && sh[FR_PATH_FOR] != best) {
            __syncthreads();
            if (tid == 0) {
                uint32_t nd = best;
                double m = -1.0;
                for (int d = Hp; d >= 0; --d) {
                    gp_path[d] = nd;
                    gp_mp[d] = m;  // largest key of the path below depth d
                    const double k = F.gkey[nd - 1];
                    m = k > m ? k : m;
                    nd = node_parent(S, nd - 1);
                }
                sh[FR_PATH_FOR] = best;
            }
            __syncthreads();
        }

        // are we done?  Open entries above the candidate's path maximum come after it; the others are looked at one by one
        // when a round selects them (fr_relevant).  An empty open set without a candidate is exhaustion (GraphSearch.m:57-61).
        const uint32_t near_n = sh[FR_NEAR_N], far_n = sh[FR_FAR_N];
        const double near_min = near_n ? sh_ld_d(sh, FR_NEAR_MIN) : inf, far_min = far_n ? sh_ld_d(sh, FR_FAR_MIN) : inf;
        const double open_min = near_min < far_min ? near_min : far_min;
        const double bb = best ? sh_ld_d(sh, FR_BEST_B1) : inf;
        bool done = false;
        if (best) {
            if (bb == open_min) return true;  // a tie between an open node and a node of the best path
            done = bb < open_min;
        } else {
            done = near_n == 0u && far_n == 0u;
        }
        if (done) {
            // Phase B right away, also when predecessors are still planning: an arrival that invalidates nothing leaves the tree,
            // hence the counts and ids, as they are, and the result goes out as soon as the last predecessor has been looked at
            // (≈ 10 µs earlier on every level of a time step's chain).  Phase B uses far's arrays as scratch: if the search has to
            // go on after all (an arrival invalidated a node), the open set is rebuilt from the tree.
            if (!pb_valid && !dep_timeout) {
                __syncthreads();
                R = fr_phase_b<NW>(A, X, F, EE, best, ref_ids, (LDS_AS unsigned char*)(X.lsm + A.lds.cand), F.far_key, F.far_id, gp_path);
                pb_valid = true;
                far_clobbered = true;
                const uint32_t pflags = sh[FR_FLAGS];
                __syncthreads();
                if (pflags & FRF_TIE) return true;
                if (pflags & FRF_BUG) dep_timeout = true;  // reported as an error status: must never happen
            }
            if (sh_load64(sh, SH_PEND_LO) == 0ull || dep_timeout) {
                goal = best;
                status = best ? PDMPC_OK : PDMPC_EXHAUSTED;
                break;
            }
            // finished, but predecessors that are still planning may yet invalidate what we found
            __builtin_amdgcn_s_sleep(16);
            if (++idle_polls > A.spin_limit) dep_timeout = true;  // a predecessor never finished: give up on it (reported as an error status)
            if (tid == 0) {
                sh[FR_RD_HEAD] = 0;
                sh[FR_RD_TAIL] = 0;
            }
            __syncthreads();
            FR_TICK(tk_wait)
            continue;
        }

        // ---- near is empty (or holds nothing below far's smallest key): refill it from far
        FR_PROGRESS(2)
        uint32_t nn_near = near_n;
        if (nn_near == 0u || far_min < near_min) {
            if (nn_near != 0u) {  // (rare: merge near into far first so that the refill sees every open entry)
                const uint32_t kept = fr_partition(
                    F.near_key, F.near_id, nn_near, wsum, n_waves, [&](double, uint32_t i) -> int { return i ? 1 : -1; }, [&](int c, double k, uint32_t i) { to_far(c == 1, k, i); });
                (void)kept;
                flush_far();
                if (tid == 0) sh[FR_NEAR_N] = 0;
                __syncthreads();
            }
            const uint32_t fn = sh[FR_FAR_N];
            double lo = sh_ld_d(sh, FR_FAR_MIN), hi = sh_ld_d(sh, FR_FAR_MAX);
            uint32_t bsel = FR_NBINS - 1;
            double scale = 0.0;
            for (int zoom = 0; zoom < 6; ++zoom) {
                scale = hi > lo ? (double)FR_NBINS / (hi - lo) : 0.0;
                for (int i = tid; i < FR_NBINS; i += (int)blockDim.x) F.hist[i] = 0;
                __syncthreads();
                fr_histogram(F, F.far_key, fn, lo, scale);
                __syncthreads();
                if (wave == 0) fr_select2(F, (uint32_t)A.fr_near_fill, (uint32_t)A.fr_near_fill, FR_SEL_BIN, FR_SEL2_BIN, lane);
                __syncthreads();
                bsel = sh[FR_SEL_BIN];
                const uint32_t cum = sh[FR_SEL_CUM];
                __syncthreads();
                if (bsel != 0u || cum <= 2u * (uint32_t)A.fr_near_fill || scale == 0.0) break;
                hi = lo + (hi - lo) / (double)FR_NBINS;  // nearly everything sits in the first bin: look closer
            }
            const double l_far_new = (bsel >= FR_NBINS - 1 || scale == 0.0) ? inf : lo + (double)(bsel + 1u) / scale;
            if (tid == 0) {
                sh_st_d(sh, FR_FAR_MIN, inf);
                sh_st_d(sh, FR_FAR_MAX, 0.0);
                sh_st_d(sh, FR_NEAR_MIN, inf);
                sh_st_d(sh, FR_NEAR_MAX, 0.0);
                sh_st_d(sh, FR_L_FAR, l_far_new);
            }
            __syncthreads();
            const double lo_c = lo, scale_c = scale;
            const uint32_t kept = fr_partition(
                F.far_key, F.far_id, fn, wsum, n_waves, [&](double k, uint32_t i) -> int { return i == 0u ? -1 : (fr_bin(k, lo_c, scale_c) <= bsel ? 1 : 0); },
                [&](int c, double k, uint32_t i) {
                    const unsigned long long b = __ballot(c == 1);
                    if (b) {
                        const uint32_t base = sh_add_uniform(sh, FR_NEAR_N, (uint32_t)__builtin_popcountll(b), lane);
                        if (c == 1) {
                            const uint32_t pos = base + lane_rank(b, lane);
                            F.near_key[pos] = k;
                            F.near_id[pos] = i;
                            near_mn = k < near_mn ? k : near_mn;
                            near_mx = k > near_mx ? k : near_mx;
                        }
                    }
                    if (c < 0 && i != 0u) {  // kept entries (i is their node, never 0)
                        far_mn = k < far_mn ? k : far_mn;
                        far_mx = k > far_mx ? k : far_mx;
                    }
                });
            flush_far();
            sh_minmax_wave(sh, FR_NEAR_MIN, FR_NEAR_MAX, near_mn, near_mx, lane);
            near_mn = inf;
            near_mx = 0.0;
            if (tid == 0) sh[FR_FAR_N] = kept;
            __syncthreads();
            nn_near = sh[FR_NEAR_N];
        }

        // ---- this round's entries: the smallest keys of near (histogram -> bin -> partition)
        FR_PROGRESS(3)
        {
            const double lo = sh_ld_d(sh, FR_NEAR_MIN);
            double hi = sh_ld_d(sh, FR_NEAR_MAX);
            // A round takes the smallest open keys: one per wavefront while the search is young (an easy search is over after
            // Hp + 1 pops; what a round takes beyond what the reference pops is wasted, but a wavefront that would otherwise idle
            // costs nothing), growing with the work done up to fr_round: the overshoot stays a fraction of the search.
            const uint32_t done_so_far = sh[FR_PROCESSED];
            const uint32_t ramp = (uint32_t)n_waves + done_so_far / (uint32_t)A.fr_ramp;
            const uint32_t round_target = ramp < (uint32_t)A.fr_round ? ramp : (uint32_t)A.fr_round;
            uint32_t bsel = FR_NBINS - 1, bspill = FR_NBINS - 1;
            double scale = 0.0;
            // (a round that takes all of near needs no histogram: scale 0 puts every key into bin 0 — the rounds of a young search)
            for (int zoom = 0; zoom < 8 && nn_near > round_target; ++zoom) {
                scale = hi > lo ? (double)FR_NBINS / (hi - lo) : 0.0;
                for (int i = tid; i < FR_NBINS; i += (int)blockDim.x) F.hist[i] = 0;
                __syncthreads();
                fr_histogram(F, F.near_key, nn_near, lo, scale);
                __syncthreads();
                if (wave == 0) fr_select2(F, round_target, (uint32_t)A.fr_near_fill, FR_SEL_BIN, FR_SEL2_BIN, lane);
                __syncthreads();
                bsel = sh[FR_SEL_BIN];
                bspill = sh[FR_SEL2_BIN];
                const uint32_t cum = sh[FR_SEL_CUM];
                __syncthreads();
                if (cum <= 4u * round_target + 16u || scale == 0.0) break;
                hi = lo + (hi - lo) / (double)FR_NBINS;  // too many entries share the first bins: look closer
            }
            // near has grown too large to scan every round: everything beyond its smallest entries moves to far
            const bool spill = nn_near > (uint32_t)A.fr_near_max && bspill < FR_NBINS - 1 && scale != 0.0;
            if (tid == 0) {
                sh[FR_RD_HEAD] = 0;
                sh[FR_RD_TAIL] = 0;
                sh_st_d(sh, FR_NEAR_MIN, inf);
                sh_st_d(sh, FR_NEAR_MAX, 0.0);
                sh_st_d(sh, FR_JOIN_MAX, 0.0);
                if (spill) sh_st_d(sh, FR_L_FAR, lo + (double)(bspill + 1u) / scale);
                sh_add(sh, FR_ROUNDS, 1u);
            }
            __syncthreads();
            const double lo_c = lo, scale_c = scale;
            const bool have_goal = best != 0u;
#ifdef FR_NO_ALIVE
            const bool check_alive = false;
#else
            const bool check_alive = sh[FR_EVER_INVAL] != 0u;  // some node lost its edge to late areas: its descendants are dead
#endif
            const uint32_t kept = fr_partition(
                F.near_key, F.near_id, nn_near, wsum, n_waves,
                [&](double k, uint32_t i) -> int {  // (every lane of the wave calls; straight-line code around the wave-wide walk)
                    const uint32_t b = fr_bin(k, lo_c, scale_c);
                    const bool sel = i != 0u && b <= bsel;
                    const bool above = have_goal && k > bb;  // above the candidate's path maximum: comes after it
                    const bool walk = sel && !above && (have_goal || check_alive);
                    if (sel && have_goal && k == bb) atomicOr((uint32_t*)&sh[FR_FLAGS], FRF_TIE);
                    const int r = fr_check_wave(F.glink, VS, F.gkey, gp_path, gp_mp, have_goal, check_alive, walk ? i : 0u, sh);
                    const int rest = (spill && b > bspill) ? 2 : 0;
                    return i == 0u ? -1 : (sel ? (above ? 3 : r) : rest);
                },
                [&](int c, double k, uint32_t i) {
                    const unsigned long long b1 = __ballot(c == 1);
                    bool spill_over = false;
                    if (b1) {
                        const uint32_t base = sh_add_uniform(sh, FR_RD_TAIL, (uint32_t)__builtin_popcountll(b1), lane);
                        const uint32_t pos = base + lane_rank(b1, lane);
                        const bool fits = c == 1 && pos < (uint32_t)FR_READY_CAP;
                        if (fits) F.ready[pos] = i;
                        sel_mx = (fits && k > sel_mx) ? k : sel_mx;
                        spill_over = c == 1 && !fits;  // (only if a thousand keys are equal to the last bit: they wait in far)
                    }
                    to_far(c == 2 || spill_over, k, i);
                    const unsigned long long b4 = __ballot(c == 4);
                    if (b4) (void)sh_add_uniform(sh, FR_DEAD, (uint32_t)__builtin_popcountll(b4), lane);
                    const unsigned long long b3 = __ballot(c == 3);
                    if (b3) (void)sh_add_uniform(sh, FR_DROPPED, (uint32_t)__builtin_popcountll(b3), lane);  // comes after the candidate: never popped
                    if (c < 0 && i != 0u) {  // kept entries
                        near_mn = k < near_mn ? k : near_mn;
                        near_mx = k > near_mx ? k : near_mx;
                    }
                });
            flush_far();
            sh_minmax_wave(sh, FR_NEAR_MIN, FR_NEAR_MAX, near_mn, near_mx, lane);
            sh_minmax_wave(sh, FR_NEAR_MIN, FR_JOIN_MAX, inf, sel_mx, lane);  // (largest key the round takes; min with +inf changes nothing)
            near_mn = inf;
            near_mx = 0.0;
            sel_mx = 0.0;
            FR_PROGRESS(4)
            if (tid == 0) {
                sh[FR_NEAR_N] = kept;
                if (A.fr_join_scale != 1.0) {  // the chains' key range: the round's own, stretched (any range gives the same results)
                    const double mxk = sh_ld_d(sh, FR_JOIN_MAX);
                    if (mxk > 0.0) sh_st_d(sh, FR_JOIN_MAX, lo + A.fr_join_scale * (mxk - lo));
                }
                const uint32_t tl = sh[FR_RD_TAIL];
                if (tl > (uint32_t)FR_READY_CAP) sh[FR_RD_TAIL] = FR_READY_CAP;
            }
            __syncthreads();
            {
                // stage the records of the round's first entries and of their parents: four threads per entry, one 16-byte
                // piece each; the parent's index comes out of the node's own last piece
                uint32_t tl = sh[FR_RD_TAIL];
                n_staged = tl < (uint32_t)A.fr_stage_cap ? tl : (uint32_t)A.fr_stage_cap;
                lds_d2* st = (lds_d2*)(X.lsm + A.lds.stage);
                for (uint32_t w0 = (uint32_t)tid; w0 < n_staged * 4u; w0 += blockDim.x) {
                    const uint32_t e = w0 >> 2, q = w0 & 3u;
                    st[8 * (size_t)e + q] = ((const d2*)(S.gn + (F.ready[e] - 1u)))[q];
                }
                __syncthreads();
                for (uint32_t w0 = (uint32_t)tid; w0 < n_staged * 4u; w0 += blockDim.x) {
                    const uint32_t e = w0 >> 2, q = w0 & 3u;
                    const uint32_t par = (uint32_t)((uint64_t)__double_as_longlong(st[8 * (size_t)e + 3].y) & 0xffffffffull);
                    st[8 * (size_t)e + 4 + q] = ((const d2*)(S.gn + (par ? par - 1u : 0u)))[q];
                }
                __syncthreads();
            }
            FR_TICK(tk_select)
        }
    }

    FR_PROGRESS(5)
    // ================= phase B: the reference's counts and ids =================
    uint32_t nnodes_raw = sh[FR_NNODES];
    nnodes_raw = nnodes_raw < S.max_nodes ? nnodes_raw : S.max_nodes;
    __syncthreads();
    const bool pb_ran = pb_valid;
    if (!pb_valid) {
        R.n_popped = 0;
        R.n_expanded = nnodes_raw;
    }
    FR_PROGRESS(6)
    // validity bytes of the LDS-resident nodes go to HBM with the rest (debug read-back of the tree, pdmpc_debug_tree)
    {
        const uint32_t nv = VS.NV < nnodes_raw ? VS.NV : nnodes_raw;
        for (uint32_t i = (uint32_t)tid; i < nv; i += blockDim.x) VS.g[i] = VS.l[i];
    }
    if (tid == 0) {
        atomicAdd(A.work_count + 2, (unsigned long long)sh[FR_PROCESSED]);
        atomicAdd(A.work_count + 3, (unsigned long long)sh[FR_ROUNDS]);
        A.tree_size[slot] = (int32_t)(nnodes_raw | 0x40000000u);  // marks the arena as a frontier tree (api.cpp reconstructs the reference's)
    }
    if (tid == 0 && A.debug_tail) {  // diagnostics in the unused tail of the record (row HP_MAX of path_nodes); PDMPC_DEBUG_TAIL=1
        double* dbg = X.O->path_nodes[PDMPC_HP_MAX];
        dbg[0] = (double)sh[FR_ROUNDS];
        dbg[1] = (double)sh[FR_PROCESSED];
        dbg[2] = (double)nnodes_raw;
        dbg[3] = (double)sh[FR_NEAR_N];
        dbg[4] = (double)sh[FR_FAR_N];
        dbg[5] = (double)sh[FR_FLAGS];
        dbg[6] = (double)(tk_work) + 1e-9 * (double)tk_select;      // work ticks . select ticks (GHz-style packing avoided: see dbg[7])
        dbg[7] = (double)(__builtin_amdgcn_s_memrealtime() - tk_start);
        X.O->path_nodes[PDMPC_HP_MAX - 1][0] = (double)tk_work;
        X.O->path_nodes[PDMPC_HP_MAX - 1][1] = (double)tk_arrival;
        X.O->path_nodes[PDMPC_HP_MAX - 1][2] = (double)tk_select;
        X.O->path_nodes[PDMPC_HP_MAX - 1][3] = (double)tk_wait;
        X.O->path_nodes[PDMPC_HP_MAX - 1][4] = (double)(tk_mark - tk_start);
        const unsigned long long slow = *(volatile lds_u64s*)(sh + FR_SLOWEST);
        X.O->path_nodes[PDMPC_HP_MAX - 1][5] = (double)(slow >> 32);
        X.O->path_nodes[PDMPC_HP_MAX - 1][6] = (double)(slow & 0xffffffffull);
    }
    X.status = status;
    X.n_popped = (int)R.n_popped;
    X.path_ready = pb_ran && goal != 0u;  // (l_path holds G's path: the epilogue need not walk it again; not set when a tie sends the search to the heap)
    X.goal = goal;
    X.nnodes = R.n_expanded;
    X.dep_timeout = dep_timeout;
    return false;
}

// ---------------------------------------------------------------------------------------------------
// Helper workgroups (pdmpc_helper_kernel, launched next to the searches on a stream of its own — a kernel of its own so that
// its registers are its own): they look for a search that has posted the shared part of a large round,
// claim a run of its entries (compare-and-swap on the board's ticket word, which carries the round's sequence number: a
// claim made on a stale view fails), mirror that search's obstacle soup in their own LDS (literal obstacles, lanelet
// boundary, the areas of the predecessors the owner had incorporated when it posted), check the entries' edges with the
// search's own device function and leave one verdict byte per entry.  A helper never waits for anything but memory, so an
// owner that waits for claimed entries always gets them; helpers leave when every search of the launch has published.
template <int CHECKER>
__device__ __forceinline__ void helper_body(const KernelArgs& A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LDS_AS unsigned char* lsm = (LDS_AS unsigned char*)smem;
    const int tid_k = threadIdx.x, lane_k = tid_k & (PDMPC_WAVE - 1), wave = uni_i(tid_k >> 6);
    int tid = tid_k, lane = lane_k;
    const int Hp = A.Hp, n_s = A.n_searches;
    // the owners' carve (search_prologue): only the regions an edge check reads are filled
    lds_u32* l_path = (lds_u32*)(lsm + A.lds.path);
    lds_i32* l_soff = (lds_i32*)(l_path + PDMPC_HP_MAX + 2);
    lds_i32* l_hoff = l_soff + PDMPC_HP_MAX + 1;
    volatile lds_u32* hs = (volatile lds_u32*)(l_hoff + PDMPC_HP_MAX + 1);
    lds_i32* l_lit = (lds_i32*)(hs + SH_WORDS);
    lds_d2* l_soup = (lds_d2*)(lsm + A.lds.soup);
    CheckCtx C;
    C.l_area = (const lds_d2*)(lsm + A.lds.area);
    C.g_area = (const d2*)A.man_area;
    C.l_soup = l_soup;
    C.l_soff = l_soff;
    C.l_hoff = l_hoff;
    C.areas_in_lds = A.areas_in_lds;
    C.Hp = Hp;
    C.checker = A.checker;
    C.sh = (lds_d2*)(lsm + A.lds.shape) + wave * (2 * PDMPC_VMAX + 1);
    C.tally = (LDS_AS unsigned long long*)(C.sh + 2 * PDMPC_VMAX);
    C.cand = nullptr;
    C.ll_base = 0;
    C.ll_len = 0;
    if (lane == 0) {
        C.tally[0] = 0;
        C.tally[1] = 0;
    }
    if (A.areas_in_lds) stage16(lsm + A.lds.area, A.man_area, A.n_man * 3 * PDMPC_VMAX, tid);
    // what an expansion reads (search_prologue, parts 1 and 2): the automaton's tables once, the reference trajectory per search
    const int n = A.n_trims, nw = A.n_words;
    lds_mask64* l_mask = (lds_mask64*)(lsm + A.lds.mask);
    lds_i16* l_mi = (lds_i16*)(lsm + A.lds.man_index);
    lds_pose* l_pose = (lds_pose*)(lsm + A.lds.pose);
    lds_f64* l_rx = (lds_f64*)(lsm + A.lds.ref);
    lds_f64* l_ry = l_rx + PDMPC_HP_MAX;
    lds_f64* l_dcum = (lds_f64*)(lsm + A.lds.expand);
    if (A.help_expand) {
        stage16(l_mask, A.succ_mask, (Hp * n * nw * 8 + 15) / 16, tid);
        stage16(l_mi, A.man_index, (n * n * 2 + 15) / 16, tid);
        stage16(l_pose, A.man_pose, A.n_man * 2, tid);
    }
    ExpandEnv EE;
    EE.l_mask = l_mask;
    EE.l_mi = l_mi;
    EE.l_pose = l_pose;
    EE.l_rx = l_rx;
    EE.l_ry = l_ry;
    EE.l_dcum = l_dcum;
    EE.l_term = (lds_f64*)((lds_u32*)(lsm + A.lds.cand) + (size_t)wave * A.cand_cap);  // this wave's scratch (as in the searches)
    EE.l_chxy = (lds_d2*)(EE.l_term + 16 * PDMPC_HP_MAX);
    EE.Hp = Hp;
    EE.n = n;
    EE.nw = nw;
    EE.lane = lane;
    if (tid < SH_WORDS) hs[tid] = 0;
    __syncthreads();
    int cur_slot = -1;
    unsigned long long cur_mask = 0;
    const int pref = (int)blockIdx.x % n_s;  // where this helper starts to look
    SpecCtx P;
    P.sh = hs;
    P.l_soup = l_soup;
    P.l_soff = l_soff;
    P.l_lit = l_lit;
    P.out = A.out;
    P.pred = A.pred;
    P.counters = A.tie_count;
    P.n_pred = 0;
    P.Hp = Hp;
    uint32_t idle = 0;
    for (;;) {
        tid = tid_k;  // (opaque per pass of the loop: see frontier_search)
        lane = lane_k;
        asm volatile("" : "+v"(tid), "+v"(lane));
        // ---- look for work: one lane per search, the first one (from pref on) with unclaimed entries is tried
        if (wave == 0) {
            uint32_t cmd = 0;
            for (int base = 0; base < n_s; base += PDMPC_WAVE) {  // (uniform trip count)
                const int k = base + lane;
                const int s_rel = k < n_s ? (pref + k) % n_s : 0;
                const unsigned long long* b = A.help_board + (size_t)(A.first + s_rel) * PDMPC_HB_WORDS;
                // one word holds the round's sequence number, its shared entries and the next unclaimed one: two loads could be
                // served out of order and pair one round's count with another round's ticket
                unsigned long long word = __hip_atomic_load(b + PDMPC_HB_TICKET, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t nsh = (uint32_t)((word >> 16) & 0xffffull);
                const uint32_t idx = (uint32_t)(word & 0xffffull);
                const bool has = k < n_s && (word >> 32) != 0ull && idx < nsh;
                const unsigned long long m = __ballot(has);
                const int l = m ? (int)__builtin_ctzll(m) : -1;
                if (lane == l) {  // (one lane; what it finds goes through LDS)
                    const uint32_t cnt = nsh - idx < (uint32_t)A.help_chunk ? nsh - idx : (uint32_t)A.help_chunk;
                    unsigned long long* bw = A.help_board + (size_t)(A.first + s_rel) * PDMPC_HB_WORDS;
                    if (__hip_atomic_compare_exchange_strong(bw + PDMPC_HB_TICKET, &word, word + cnt, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        const unsigned long long mask = __hip_atomic_load(b + PDMPC_HB_MASK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        hs[HS_SLOT] = (uint32_t)(A.first + s_rel);
                        hs[HS_FIRST] = idx;
                        hs[HS_COUNT] = cnt;
                        hs[HS_MASK_LO] = (uint32_t)mask;
                        hs[HS_MASK_HI] = (uint32_t)(mask >> 32);
                        hs[HS_EXPAND] = (uint32_t)((__hip_atomic_load(b + PDMPC_HB_FLAGS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 1) & 1ull);
                        hs[HS_TICKET] = 0;
                        hs[HS_CMD] = 1;
                    }
                }
                wave_sync();
                cmd = uni_u(hs[HS_CMD]);
                if (cmd) break;  // (uniform)
            }
            if (!cmd) {
                const uint32_t fin = __hip_atomic_load(A.help_finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 0 && fin >= (uint32_t)n_s) hs[HS_CMD] = 2;
            } else {
                // what the owner wrote before it posted (and the predecessors it had seen) is visible from here on; not on an idle
                // poll: the fence empties this XCD's L2, which searches on neighbouring CUs share
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        __syncthreads();
        const uint32_t cmd = hs[HS_CMD];
        if (cmd == 2u) break;
        if (cmd == 0u) {
            if (idle < 64u)
                __builtin_amdgcn_s_sleep(2);
            else
                __builtin_amdgcn_s_sleep(32);
            if (++idle > (A.spin_limit >> 4)) break;  // (uniform) the searches never came: leave; they do without helpers
            __syncthreads();
            continue;
        }
        idle = 0;
        const int slot = (int)hs[HS_SLOT];
        const uint32_t first = hs[HS_FIRST], cnt = hs[HS_COUNT];
        const unsigned long long mask = ((unsigned long long)hs[HS_MASK_HI] << 32) | hs[HS_MASK_LO];
        const bool expand_round = A.help_expand != 0 && hs[HS_EXPAND] != 0u;
        const DevVehicle* __restrict__ V = A.veh + slot;
        // ---- the search's obstacle soup (search_prologue, parts 3 and 4)
        if (slot != cur_slot) {
            const int pred_cols = V->n_pred * PDMPC_VMAX;
            int off = 0;
            for (int k = 0; k < Hp; ++k) {
                const int a = V->lit_off[k], b = V->lit_off[k + 1];
                if (tid == 0) {
                    l_soff[k] = off;
                    l_lit[k] = b - a;
                }
                stage16(l_soup + off, (const d2*)A.points + a, b - a, tid);
                off += (b - a) + pred_cols;
            }
            if (tid == 0) l_soff[Hp] = off;
            for (int k = 0; k < Hp; ++k) {
                const int a = V->hdv_off[k], b = V->hdv_off[k + 1];
                if (tid == 0) l_hoff[k] = off;
                stage16(l_soup + off, (const d2*)A.points + a, b - a, tid);
                off += (b - a);
            }
            if (tid == 0) l_hoff[Hp] = off;
            stage16(l_soup + off, (const d2*)A.points + V->ll_off, V->ll_len, tid);
            C.ll_base = off;
            C.ll_len = V->ll_len;
            __syncthreads();
            const d2 nanpt = d2{__longlong_as_double(0x7ff8000000000000LL), __longlong_as_double(0x7ff8000000000000LL)};
            for (int idx = tid; idx < Hp * pred_cols; idx += (int)blockDim.x) {
                const int k = idx / pred_cols;
                l_soup[l_soff[k] + l_lit[k] + (idx - k * pred_cols)] = nanpt;
            }
            __syncthreads();
            P.pred = A.pred + V->pred_off;
            P.n_pred = V->n_pred;
            incorporate_areas(P, mask, tid);
            if (A.help_expand) {  // (search_prologue, part 2)
                if (tid < Hp) {
                    l_rx[tid] = V->ref_x[tid];
                    l_ry[tid] = V->ref_y[tid];
                }
                if (tid >= PDMPC_WAVE && tid < PDMPC_WAVE + Hp) {
                    const int k_exp = tid - PDMPC_WAVE + 1;
                    double d = 0.0;
                    for (int it = 1; it <= Hp - k_exp; ++it) {
                        d = d + A.dt * V->v_ref[k_exp + it - 1];
                        l_dcum[(k_exp - 1) * PDMPC_HP_MAX + (it - 1)] = d;
                    }
                }
            }
            cur_slot = slot;
            cur_mask = mask;
        } else if (mask != cur_mask) {
            incorporate_areas(P, mask & ~cur_mask, tid);  // (within a launch a search's set of incorporated predecessors only grows)
            cur_mask = mask;
        }
        __syncthreads();
        lds_u32* r_cnt = (lds_u32*)(lsm + A.lds.stage);      // [128] children of the run's entries that will be expanded (0: none)
        lds_u32* r_off = r_cnt + 128;                          // [128] their offsets in the run's block of node indices
        lds_d2* r_rec = (lds_d2*)(r_off + 128);                // [128][4] their records (piece 2: cos / sin of the yaw)
        if (tid < 128) r_cnt[tid] = 0;
        __syncthreads();
        // ---- the claimed entries, one per wave at a time
        {
            const NodeRec* gn = A.arena.nodes + (size_t)slot * A.max_nodes;
            const uint32_t* list = A.help_list + (size_t)slot * PDMPC_HELP_CAP;
            uint32_t* verdict = A.help_verdict + (size_t)slot * PDMPC_HELP_CAP;
            d2* hcs_out = (d2*)A.help_cs + (size_t)slot * PDMPC_HELP_CAP;
            // Four entries per wave at a time, their loads side by side: an entry is three dependent round trips to memory (list ->
            // node -> parent) and a helper has nothing else to hide them behind.  Of the records only what an edge check reads.
            const uint32_t nwv = blockDim.x >> 6;
            for (uint32_t base = (uint32_t)wave; base < cnt; base += 4u * nwv) {
                uint32_t ee[4], id[4];
                bool in[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t t = base + (uint32_t)j * nwv;
                    in[j] = t < cnt;
                    ee[j] = first + (in[j] ? t : 0u);
                    id[j] = list[ee[j]];
                }
                d2 c3[4], c1[4], c0[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    c3[j] = ((const d2*)(gn + (uni_u(id[j]) - 1u)))[3];  // h, parent | packed << 32
                    c1[j] = ((const d2*)(gn + (uni_u(id[j]) - 1u)))[1];  // yaw, g
                    c0[j] = ((const d2*)(gn + (uni_u(id[j]) - 1u)))[0];  // x, y
                }
                d2 p0[4], p2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t par = (uint32_t)((uint64_t)__double_as_longlong(c3[j].y) & 0xffffffffull);
                    const NodeRec* pr = gn + (par ? uni_u(par) - 1u : 0u);
                    p0[j] = ((const d2*)pr)[0];  // x, y
                    p2[j] = ((const d2*)pr)[2];  // cos, sin of its yaw
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (in[j]) {  // (uniform)
                        NodeBits cu, pu;
                        cu.q[3] = c3[j];
                        pu.q[0] = p0[j];
                        pu.q[2] = p2[j];
                        cu.q[0] = c0[j];
                        cu.q[1] = c1[j];
                        const bool valid = edge_valid_recs<CHECKER>(C, cu.r, pu.r, lane);
                        uint32_t v = valid ? 1u : 2u;  // 1: collision-free (and expanded, if helpers expand), 2: collides
                        const uint32_t cpk = uni_u(cu.r.packed);
                        if (valid && NODE_K(cpk) == Hp) v = 3u;  // at the horizon: the owner's goal test
                        if (valid && NODE_K(cpk) < Hp) {  // (uniform) it will be expanded: cos / sin of its yaw (expand_node.m:50-51) on this CU's time
                            double hsn, hcos;
                            pdmpc_sincos(c1[j].x, &hsn, &hcos);
                            d2 t;
                            t.x = hcos;
                            t.y = hsn;
                            if (lane == 0) hcs_out[ee[j]] = t;
                            if (expand_round) {  // remembered for the run's expansion pass (below)
                                const lds_mask64* mrow = l_mask + ((size_t)NODE_K(cpk) * n + (NODE_TRIM(cpk) - 1)) * nw;
                                const uint32_t t_run = ee[j] - first;
                                if (lane == 0) {
                                    r_cnt[t_run] = (uint32_t)__builtin_popcountll(mrow[0]);
                                    r_rec[4 * (size_t)t_run + 0] = c0[j];
                                    r_rec[4 * (size_t)t_run + 1] = c1[j];
                                    r_rec[4 * (size_t)t_run + 2] = t;
                                    r_rec[4 * (size_t)t_run + 3] = c3[j];
                                }
                            }
                        }
                        if (lane == 0) verdict[ee[j]] = v;
                    }
                }
            }
        }
        if (expand_round) {
            // One allocation for the whole run: the children counts of its collision-free entries are summed (wave 0, a lane per
            // entry), the board's node counter is advanced once, every entry gets its offset; then the wavefronts take the entries
            // to expand off an LDS ticket.  (A global atomic per expansion cost more than the expansion.)
            __syncthreads();
            const NodeRec* gn = A.arena.nodes + (size_t)slot * A.max_nodes;
            (void)gn;
            unsigned long long* board = A.help_board + (size_t)slot * PDMPC_HB_WORDS;
            uint32_t* verdict = A.help_verdict + (size_t)slot * PDMPC_HELP_CAP;
            if (wave == 0) {
                uint32_t run_total = 0;
                for (uint32_t b0 = 0; b0 < cnt; b0 += PDMPC_WAVE) {  // (uniform; a run is at most 128 entries)
                    const uint32_t t = b0 + (uint32_t)lane;
                    const uint32_t c = t < cnt ? r_cnt[t] : 0u;
                    uint32_t inc = c;
#pragma unroll
                    for (int o = 1; o < PDMPC_WAVE; o <<= 1) {
                        const uint32_t vv = (uint32_t)__shfl_up((int)inc, o);
                        if (lane >= o) inc += vv;
                    }
                    if (t < cnt) r_off[t] = run_total + inc - c;
                    run_total += lane_u(inc, PDMPC_WAVE - 1);
                }
                if (lane == 0) {
                    hs[HS_RUN_BASE] = run_total ? (uint32_t)__hip_atomic_fetch_add(board + PDMPC_HB_NNODES, (unsigned long long)run_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    hs[HS_RUN_TOTAL] = run_total;
                    hs[HS_TICKET] = 0;
                }
            }
            __syncthreads();
            const uint32_t run_base = hs[HS_RUN_BASE], run_total = hs[HS_RUN_TOTAL];
            const bool full = run_total != 0u && run_base + run_total > A.max_nodes;
            if (full && tid == 0) __hip_atomic_fetch_or(board + PDMPC_HB_FLAGS, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            double* gkey = A.arena.heap_key + (size_t)slot * A.max_nodes;
            unsigned long long* glink = A.arena.link + (size_t)slot * A.max_nodes;
            Search SH;  // the owner's arena, no LDS copies
            SH.ln = nullptr;
            SH.gn = A.arena.nodes + (size_t)slot * A.max_nodes;
            SH.NL = 0;
            SH.max_nodes = A.max_nodes;
            SH.lkey = nullptr;
            SH.lid = nullptr;
            SH.gkey = gkey;
            SH.gid = nullptr;
            SH.HL = 0;
            SH.heap_len = 0;
            SH.lane = lane;
            SH.pl = make_pop_lane(lane);
            VState VH;  // (the owner sets the children's validity bytes itself: part of them lives in its LDS)
            VH.l = nullptr;
            VH.g = A.arena.vstate + (size_t)slot * A.max_nodes;
            VH.NV = 0;
            const uint32_t* list = A.help_list + (size_t)slot * PDMPC_HELP_CAP;
            for (;;) {
                const uint32_t t = sh_add_uniform(hs, HS_TICKET, 1u, lane);
                if (t >= cnt) break;
                const uint32_t c = uni_u(r_cnt[t]);
                if (c == 0u) continue;  // (collides, or at the horizon)
                if (full) {
                    if (lane == 0) verdict[first + t] = 4u;  // collision-free, but the arena is full
                    continue;
                }
                NodeBits cu;
#pragma unroll
                for (int q = 0; q < 4; ++q) cu.q[q] = r_rec[4 * (size_t)t + q];
                const double hcos = cu.r.cs, hsn = cu.r.sn;
                const uint32_t cur = uni_u(list[first + t]);
                if (lane == 0) node_store_cs(SH, cur - 1u, hcos, hsn);
                uint32_t nn = run_base + uni_u(r_off[t]);
                (void)expand_children<false, 1>(EE, SH, VH, cur, cu.r, hcos, hsn, nn, [&](uint64_t mask, bool active, uint32_t i0, double f, int ccnt, const NodeRec& ch) {
                    (void)mask;
                    (void)ccnt;
                    if (active) {
                        gkey[i0] = f;
                        glink[i0] = (unsigned long long)ch.parent | ((unsigned long long)ch.packed << 32);
                    }
                });
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's verdicts have reached L2 (the barrier alone does not wait for them) ...
        __syncthreads();                                     // ... every wave's have: thread 0 can write L2 back and report
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(A.help_board + (size_t)slot * PDMPC_HB_WORDS + PDMPC_HB_DONE, (unsigned long long)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hs[HS_CMD] = 0;
        }
        __syncthreads();
    }
    if (lane == 0) {
        atomicAdd(A.work_count + 0, C.tally[0]);
        atomicAdd(A.work_count + 1, C.tally[1]);
    }
}

template <int CHECKER, int NW>
__device__ __forceinline__ void frontier_body(const KernelArgs& A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Ctx X;
    search_prologue(A, X, (LDS_AS unsigned char*)smem, CHECKER == PDMPC_CHECK_INTERX);
    const int tid = X.tid, lane = X.lane, wave = X.wave;
    volatile lds_u32* l_shared = X.l_shared;
    lds_u32* ref_ids = (lds_u32*)(X.lsm + A.lds.heap_key) + FR_READY_CAP + 128;  // behind the relevance tables (nothing else uses those words: the ids survive an arrival that changes nothing)
    const bool tie = frontier_search<CHECKER, NW>(A, X, ref_ids);
    bool serial = false;
    if (tie) {  // (uniform over the workgroup) start over with the exact open list; areas that arrived so far stay in the soup
        __syncthreads();
        if (tid == 0) {
            l_shared[SH_STATE] = ST_RUN;
            l_shared[SH_ARR_LO] = 0;
            l_shared[SH_ARR_HI] = 0;
            l_shared[SH_RESTART] = 0;
            atomicAdd(A.tie_count, 1);
        }
        __syncthreads();
        (void)search_loops<CHECKER, false, NW>(A, X);
        serial = true;
    }
    if (lane == 0) {
        atomicAdd(A.work_count + 0, X.C.tally[0]);
        atomicAdd(A.work_count + 1, X.C.tally[1]);
    }
    __syncthreads();
    if (wave != 0) return;
    search_epilogue(A, X, serial ? nullptr : ref_ids);
    if (lane == 0 && A.n_helpers > 0) atomicAdd(A.help_finished, 1u);  // (the helpers leave when every search is through)
}

}  // namespace

extern "C" __global__ __launch_bounds__(PDMPC_MAX_THREADS) void pdmpc_frontier_kernel(const KernelArgs A) { frontier_body<PDMPC_CHECK_INTERX, 1>(A); }
extern "C" __global__ __launch_bounds__(PDMPC_MAX_THREADS) void pdmpc_frontier_kernel_sat(const KernelArgs A) { frontier_body<PDMPC_CHECK_SAT, 1>(A); }
extern "C" __global__ __launch_bounds__(PDMPC_MAX_THREADS) void pdmpc_frontier_kernel_wide(const KernelArgs A) { frontier_body<PDMPC_CHECK_INTERX, 0>(A); }
extern "C" __global__ __launch_bounds__(PDMPC_MAX_THREADS) void pdmpc_frontier_kernel_sat_wide(const KernelArgs A) { frontier_body<PDMPC_CHECK_SAT, 0>(A); }

extern "C" int pdmpc_launch_frontier(const KernelArgs* args, int count, void* stream, uint32_t* lds_high_water) {
    if (count <= 0) return 0;
    typedef void (*kernel_t)(const KernelArgs);
    const bool interx = args->checker == PDMPC_CHECK_INTERX, one_word = args->n_words == 1;
    kernel_t fn = interx ? (one_word ? pdmpc_frontier_kernel : pdmpc_frontier_kernel_wide) : (one_word ? pdmpc_frontier_kernel_sat : pdmpc_frontier_kernel_sat_wide);
    // (no register-capped variants for two workgroups per CU: measured on C5, 1280 searches on 256 CUs: 2 x 12 wavefronts at 80
    // VGPRs 283 steps/s, 2 x 8 at 128 VGPRs 324, one workgroup of 16 per CU 345)
    // (the attribute is a maximum: raised when a launch needs more than any before it, not on every launch; the high-water marks
    // live in the handle -- one per kernel variant --, so handles on different devices or host threads share nothing)
    uint32_t& have = lds_high_water[(interx ? 0 : 1) + (one_word ? 0 : 2)];
    if (args->lds.total > have) {
        hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)args->lds.total);
        if (e != hipSuccess) return (int)e;
        have = args->lds.total;
    }
    hipLaunchKernelGGL(fn, dim3(count), dim3(PDMPC_WAVE * args->n_waves), args->lds.total, (hipStream_t)stream, *args);
    return (int)hipGetLastError();
}

extern "C" __global__ __launch_bounds__(PDMPC_MAX_THREADS) void pdmpc_helper_kernel(const KernelArgs A) { helper_body<PDMPC_CHECK_INTERX>(A); }

extern "C" int pdmpc_launch_helpers(const KernelArgs* args, void* stream, uint32_t* lds_high_water) {
    if (args->n_helpers <= 0) return 0;
    uint32_t& have = lds_high_water[4];
    if (args->lds.total > have) {
        hipError_t e = hipFuncSetAttribute((const void*)pdmpc_helper_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)args->lds.total);
        if (e != hipSuccess) return (int)e;
        have = args->lds.total;
    }
    hipLaunchKernelGGL(pdmpc_helper_kernel, dim3(args->n_helpers), dim3(PDMPC_WAVE * args->n_waves), args->lds.total, (hipStream_t)stream, *args);
    return (int)hipGetLastError();
}
