// bulk_search.hpp — the optimal graph search (GraphSearch.m:23-196) as bulk-synchronous passes over a round of open nodes; the helper
// workgroups that share a search's large rounds; the kernel body.  Instantiated in bulk_kernel.hip (InterX, one successor-mask word:
// every BASELINE road-network configuration), bulk_kernel_wide.hip (InterX, automata of more than 64 trims), bulk_kernel_sat.hip (the
// separating-axis checker) and bulk_kernel_compact.hip (InterX, one word, fixed LDS regions sized for 8 wavefronts: two workgroups
// per CU for launches of more searches than CUs).
//
// The closed form the rounds rest on (tools/sigma_order.py checks it against the oracle's pop sequences): with
// pairwise distinct keys the reference pops X before Y iff X is an ancestor of Y or the largest key on the path (LCA, X] is smaller
// than the largest key on (LCA, Y].  With G the first collision-free node at the horizon in that order, the reference pops exactly the
// nodes in front of G, so ANY processing order gives its result as long as every generated node in front of G has been evaluated in
// the end; ids, n_popped and the tree size are counts of nodes in front of G's path nodes (phase B, frontier_common.hpp).
//
// A round is a handful of passes in which every LANE has an item of its own and talks to nobody until the barrier that ends the pass:
//
//   P1  check      item = (ready node, chunk of S obstacle segments): eval_edge_exact (GraphSearch.m:111-196) with InterX
//                  (InterX.m:63-76, are_constraints_satisfied_interx.m:17-37) restricted to the chunk — or, for the separating-axis
//                  checker, a chunk of soup columns / boundary segments (are_constraints_satisfied_sat.m:15-53) —; a hit sets the
//                  node's flag; chunk-major order, so later chunks of a node that already collides are skipped (the reference's early
//                  out).  S is chosen per round so that the items fill the workgroup once.  A large round is shared with the helper
//                  workgroups that have taken a seat at this search (bulk_helper_body): posted records, one assignment word per seat.
//       + sincos   item = ready node: cos / sin of its yaw (expand_node.m:50-51) into its record, next to the checks
//   P2  verdicts   item = ready node: validity byte, goal candidates (GraphSearch.m:81-90), children counts; one workgroup scan
//                  hands out the children's node indices (Tree.m:61: the children of a node are consecutive, ascending trim)
//   P3  expand     item = (collision-free node, successor): expand_node.m:18-90 — pose, cost-to-come, cost-to-go summed in the
//                  reference's order by the lane itself —, record, key, link, open-list entry
//   P4  boundary   goal candidates resolved; predecessors that finished meanwhile copied in and verified
//                  (PrioritizedController.m:476-491; expected areas until then); termination test; phase B or
//                  the replay when done; else the next round: the smallest keys of `near` — picked by the first wavefront alone
//                  while near is small (bisection on the key, ballots), by a 256-bin histogram over all wavefronts otherwise
//
// The open set: `near` lives in LDS (keys + nodes, unordered, BK_PER entries per thread so that a selection pass holds it in
// registers), `far` in HBM takes what near cannot hold, a heavy search's far list feeds near through the `mid` list (section 3.3).
// Arithmetic, operation order and -ffp-contract=off are the oracle's: every record is bit-identical to it.
// Equal keys where the pop order decides (priority_queue_interface_mex.cpp:19-31: the layout of std::priority_queue's binary heap
// decides, not the nodes): the search goes into tie mode, processes a superset of what any heap order can pop, and ends — in the same
// launch — on bk_replay, which pops its tree once more through the libstdc++-faithful heap (heap_queue.hpp; section 3.6).
#include <hip/hip_runtime.h>

#include "search_common.hpp"

#include "frontier_common.hpp"

// shared words of the bulk kernel (aliases of words the frontier kernel uses for things this kernel does not have)
#ifndef PDMPC_BK_CULL
#define PDMPC_BK_CULL 0
#endif
#define BK_P2 3                   // ready entries a thread handles in the verdict pass (the ready list holds at most BK_P2 * blockDim entries)
#define BK_PER PDMPC_BK_PER       // near entries per thread a selection pass holds in registers (near capacity = BK_PER * blockDim)
#define BK_FAST_PER 4             // ... and per lane of the first wavefront when it selects alone (a small open set: at most 256 entries)

namespace {

typedef volatile LDS_AS unsigned long long lds_vu64;

// exclusive prefix of v over the threads of the workgroup (thread order), total to every thread.  Every thread calls; ONE barrier:
// consecutive calls alternate between two partials arrays (the caller passes them), so the partials of a call are not rewritten
// before the call after the next — which lies behind at least one more barrier.
__device__ __forceinline__ unsigned long long wg_scan_excl(unsigned long long v, lds_vu64* wsum, int lane, int wave, int n_waves, unsigned long long& total) {
    unsigned long long inc = v;
#pragma unroll
    for (int o = 1; o < PDMPC_WAVE; o <<= 1) {
        const unsigned long long t = __shfl_up(inc, o);
        inc += lane >= o ? t : 0ull;
    }
    if (lane == PDMPC_WAVE - 1) wsum[wave] = inc;
    __syncthreads();
    unsigned long long off = 0, tot = 0;
    for (int q = 0; q < n_waves; ++q) {
        const unsigned long long w = wsum[q];
        off += q < wave ? w : 0ull;
        tot += w;
    }
    total = tot;
    return off + inc - v;
}

// piece q (16 bytes) of node i0's record: LDS copy if it has one
__device__ __forceinline__ d2 node_piece(const Search& S, uint32_t i0, int q) {
    if (i0 < S.NL) return S.ln[4 * (size_t)i0 + q];
    return ((const d2*)(S.gn + i0))[q];
}
__device__ __forceinline__ void piece_link(d2 p3, uint32_t& parent, uint32_t& packed) {
    const uint64_t u = (uint64_t)__double_as_longlong(p3.y);
    parent = (uint32_t)(u & 0xffffffffull);
    packed = (uint32_t)(u >> 32);
}

// What a check item reads.  The same for the owner of a search and for a workgroup that helps it.
struct BkCheck {
    const lds_d2* l_area;
    const d2* g_area;
    const lds_d2* l_soup;
    const lds_i32* l_soff;
    const lds_i32* l_hoff;
    const lds_i32* l_lit;  // literal soup length per step (the predecessors' slots of VMAX columns each follow)
    int areas_in_lds, ll_base, ll_len, Hp;
};

// Tentative areas.  A predecessor that is still planning has its EXPECTED areas in its soup slots: what it publishes should its search
// be exhausted, i.e. its previous plan shifted by one step (PrioritizedController.m:568-616, 678-718) — where it most likely ends up
// driving.  An edge that crosses only such areas is neither collision-free nor colliding: its node is parked (VS_TENT) and comes back
// into the open set when a predecessor arrives.  The result is a function of the final areas alone (every node that comes before
// the goal ends up evaluated against them); what changes is that the plan found ahead of the arrivals usually survives them.
#define VS_TENT 6u
#define BK_TENT_MIN FR_JOIN_MAX  // (64 bit) smallest key among the parked nodes
#define BK_NTENT FR_SEL2_BIN     // parked nodes
#define BK_ARRIVALS FR_SEL2_CUM  // arrival events handled by this search
#define BK_DEPTH FR_RD_HEAD       // deepest collision-free node so far (its step k)
#define BK_IDLE FR_RD_TAIL        // polls a waiting search has made (the watchdog's count)
// the mid list's words (words 10-15 of the serial block: the pop-ordered kernel's candidate list, unused by this search)
#define BK_MID_N 10    // entries of mid
#define BK_MID_MIN 12  // (64 bit) exact minimum key of mid
#define BK_L_MID 14    // (64 bit) open entries that leave near, and children beyond near's limit, go to mid up to this key and to far above it (-1: no mid list)
// the fast arrival path (bk_wait_done; words 16-19 of the serial block: the pop-ordered kernel's mail boxes, unused by this search)
#define BK_FD_LO 16    // (64 bit) predecessors whose areas are in the soup and have passed the path of the finished plan, but whose re-check of the
#define BK_FD_HI 17    //          other collision-free nodes is still to come (they stay in SH_PEND until the arrival block has seen them)
#define BK_WAITRES 18  // result of bk_wait_done: 0 nothing yet, 1 an arrival crosses the path, 2 the last predecessor has passed: published
#define BK_TIEMODE 20   // the search has met equal keys where the pop order decides (or PDMPC_BK_FORCE_TIE): it ends on bk_replay
#define BK_RP_NEED 21   // bk_replay: a node (1-based) the reference's heap pops that no round has evaluated (0: none)
#define BK_RP_GOAL 22   // ... the goal it ended on (1-based arena index, 0: exhausted)
#define BK_RP_NPOP 23   // ... nodes popped
#define BK_RP_NREF 24   // ... nodes of the reference's tree
#define BK_PUBLISHED 19 // the done flag is out (bk_wait_done): the areas of the record in HBM are final and may be read; only counts and ids may still be written

// copies the expected areas of the predecessors in `who` into their soup slots
__device__ __forceinline__ void bk_tentative_areas(const KernelArgs& A, const SpecCtx& P, unsigned long long who, int tid, int nthreads) {
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    const int per = P.Hp * PDMPC_VMAX, n_pred = P.n_pred < 64 ? P.n_pred : 64;
    for (int idx = tid; idx < n_pred * per; idx += nthreads) {  // (every pending predecessor's loads side by side)
        const int p = idx / per, rest = idx - p * per;
        if (!((who >> p) & 1ull)) continue;
        const DevVehicle* PV = A.veh + P.pred[p];
        if (PV->fb_off[0] < 0) continue;  // no expectation: its slots stay empty
        const int k = rest / PDMPC_VMAX, v = rest - k * PDMPC_VMAX;
        const int a = PV->fb_off[k], b = PV->fb_off[k + 1];
        const int cols = (b - a < PDMPC_VMAX) ? (b - a) : PDMPC_VMAX;
        d2 pt;
        pt.x = v < cols ? A.points[2 * (size_t)(a + (v < cols ? v : 0))] : qnan;
        pt.y = v < cols ? A.points[2 * (size_t)(a + (v < cols ? v : 0)) + 1] : qnan;
        P.l_soup[P.l_soff[k] + P.l_lit[k] + p * PDMPC_VMAX + v] = pt;
    }
}

// Where a check item finds its node: the owner of a search reads the tree (LDS copies where they exist), a workgroup that helps it
// reads the 48-byte records the owner has posted for the round (staged in the helper's LDS).
struct BkTreeSrc {
    const Search* S;
    const lds_u32* ready;
    __device__ __forceinline__ void link(uint32_t r, uint32_t& parent, uint32_t& packed) const { piece_link(node_piece(*S, ready[r] - 1u, 3), parent, packed); }
    __device__ __forceinline__ void pose(uint32_t, uint32_t parent, double& px, double& py, double& cs, double& sn) const {
        const d2 pxy = node_piece(*S, parent - 1u, 0), pcs = node_piece(*S, parent - 1u, 2);
        px = pxy.x;
        py = pxy.y;
        cs = pcs.x;
        sn = pcs.y;
    }
};
struct BkPostSrc {
    const lds_d2* rec;  // [tile][3]: (x, y) and (cos, sin) of the parent, (parent | packed << 32 as bits, -)
    __device__ __forceinline__ void link(uint32_t r, uint32_t& parent, uint32_t& packed) const { piece_link(d2{0.0, rec[3 * r + 2].x}, parent, packed); }
    __device__ __forceinline__ void pose(uint32_t r, uint32_t, double& px, double& py, double& cs, double& sn) const {
        const d2 pxy = rec[3 * r], pcs = rec[3 * r + 1];
        px = pxy.x;
        py = pxy.y;
        cs = pcs.x;
        sn = pcs.y;
    }
};

// P1, the check items of the entries r0 .. r0 + R - 1: item = c * R + r (chunk-major), chunk c = S = 1 << ls consecutive segments of
// one of the node's three soups (vehicle obstacles of its step and HDV sets against the area, lanelet boundary against the
// boundary-check area: are_constraints_satisfied_interx.m:17-37).  chmax = chunks of the step with the most segments.  Lanes work alone.
// What a node's check items cover, per soup: segments (InterX: the vehicle obstacles of its step, the HDV sets, the lanelet boundary) or,
// for the separating-axis checker, COLUMNS of the vehicle obstacles' soup — the lane whose chunk holds a polygon's first column tests
// that polygon whole (are_constraints_satisfied_sat.m:15-35; its HDV loop is dead code, :55-66) — and segments of the boundary (:46-53).
template <int CHECKER>
__device__ __forceinline__ void bk_item_counts(int M_k, int Hk, int ll_len, int& n0, int& n1, int& n2) {
    n0 = CHECKER == PDMPC_CHECK_INTERX ? (M_k > 1 ? M_k - 1 : 0) : M_k;
    n1 = CHECKER == PDMPC_CHECK_INTERX ? (Hk > 1 ? Hk - 1 : 0) : 0;
    n2 = ll_len > 1 ? ll_len - 1 : 0;
}
template <int CHECKER, class Src>
__device__ __forceinline__ void bk_check_items(const BkCheck& C, const Src& src, volatile lds_u32* r_flag, uint32_t r0, uint32_t R, int ls, uint32_t chmax, unsigned long long pend, int tid, int nthreads) {
    const uint32_t items = R * chmax;
    const int Sg = 1 << ls;
    for (uint32_t item = (uint32_t)tid; item < items; item += (uint32_t)nthreads) {
        const uint32_t c = item / R, r = r0 + (item - c * R);
        if (r_flag[r] & 1u) continue;  // collides already: the reference's early out
        uint32_t parent, packed;
        src.link(r, parent, packed);
        if (!parent) continue;  // the root has no edge (GraphSearch.m:137-139)
        const int k = NODE_K(packed), m = NODE_MAN(packed), ncols = NODE_COLS(packed);
        const int so = C.l_soff[k - 1], ho = C.l_hoff[k - 1];
        const int M_k = C.l_soff[k] - so, Hk = C.l_hoff[k] - ho;
        int n0, n1, n2;
        bk_item_counts<CHECKER>(M_k, Hk, C.ll_len, n0, n1, n2);
        const uint32_t c0 = (uint32_t)((n0 + Sg - 1) >> ls), c1 = (uint32_t)((n1 + Sg - 1) >> ls), c2 = (uint32_t)((n2 + Sg - 1) >> ls);
        int base, t0, left, which = 0;
        if (c < c0) {
            base = so;
            t0 = (int)(c << ls);
            left = n0 - t0;
        } else if (c < c0 + c1) {
            base = ho;
            t0 = (int)((c - c0) << ls);
            left = n1 - t0;
        } else if (c < c0 + c1 + c2) {
            base = C.ll_base;
            t0 = (int)((c - c0 - c1) << ls);
            left = n2 - t0;
            which = k == C.Hp ? 2 : 1;  // large offset at k == Hp, else without offset (GraphSearch.m:161-174)
        } else {
            continue;
        }
        const int tn = left < Sg ? left : Sg;
        double cc, ss, pX, pY;
        src.pose(r, parent, pX, pY, cc, ss);
        const size_t abase = ((size_t)m * 3 + (size_t)which) * PDMPC_VMAX;
        d2 pt[PDMPC_VMAX];
#pragma unroll
        for (int i = 0; i < PDMPC_VMAX; ++i) {  // (columns beyond ncols are padding: transformed, never used)
            const d2 a = C.areas_in_lds ? (d2)C.l_area[abase + i] : C.g_area[abase + i];
            pt[i].x = cc * a.x - ss * a.y + pX;  // GraphSearch.m:158 / :162 / :168
            pt[i].y = ss * a.x + cc * a.y + pY;  // :159 / :163 / :169
        }
        const lds_d2* q = C.l_soup + base + t0;
        if (CHECKER == PDMPC_CHECK_SAT) {
            uint32_t fnd = 0;  // 1: overlaps a real area, 2: an expected one (the slot of a predecessor that is still planning, `pend`)
            if (which == 0) {  // the polygons that begin in this chunk of the step's soup ([polygon, NaN] ..., then the predecessors' slots)
                const int lit = C.l_lit[k - 1];
                for (int t = 0; t < tn && !(fnd & 1u); ++t) {
                    const int j = t0 + t;
                    // (a predecessor's slot begins a polygon of its own: its first column follows the last column of a full slot in front of it)
                    const bool start = !is_nan(q[t].x) && (j == 0 || is_nan(C.l_soup[base + j - 1].x) || (j >= lit && ((j - lit) & (PDMPC_VMAX - 1)) == 0));
                    if (start) {
                        int e = j + 1;
                        const int lim = j >= lit ? j + PDMPC_VMAX - ((j - lit) & (PDMPC_VMAX - 1)) : M_k;  // (a slot ends with its VMAX columns)
                        while (e < lim && !is_nan(C.l_soup[base + e].x)) ++e;
                        if (sat_pair_lane(pt, ncols, C.l_soup + base + j, e - j)) {  // intersect_sat.m:1-42
                            const bool tent = j >= lit && ((pend >> ((j - lit) >> 3)) & 1ull) != 0ull;
                            fnd |= tent ? 2u : 1u;
                        }
                    }
                }
            } else {  // intersect_lanelet_boundary.m:16-54 on [left, NaN, right, NaN]
                double min_x, max_x, min_y, max_y;
                sat_area_bbox(pt, ncols, min_x, max_x, min_y, max_y);
                for (int t = 0; t < tn && !fnd; ++t) fnd = sat_boundary_segment_lane(pt, ncols, min_x, max_x, min_y, max_y, q[t], q[t + 1]) ? 1u : 0u;
            }
            if (fnd) __hip_atomic_fetch_or((lds_u32*)&r_flag[r], fnd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            continue;
        }
        d2 q0 = q[0];
        // (segments of the slots of predecessors that are still planning — `pend` — hold expected areas: a crossing there is tentative)
        const int rel0 = (which == 0 && base == so) ? t0 - C.l_lit[k - 1] : -(1 << 20);
        uint32_t found = 0;  // 1: crosses a real area, 2: crosses an expected one
#if PDMPC_BK_CULL
        // The exact bounding-box cull, built to be measured (-DPDMPC_BK_CULL=1; off in the product): a segment whose supporting line has
        // the area's bounding box strictly on one side cannot pass C2 — the expression (y dx2 - x dy2) - S2 is monotone in x and in y
        // under IEEE rounding, so its extremes over the box are taken at two corners and bound every point's value.  Survivors are
        // collected in a bit mask first and tested afterwards (a lane-level skip inside the segment loop saves nothing: some lane of
        // the 64 always needs the full test).  Results identical; C2 / C3 / C4 / C5: 1 040 / 1 003 / 69.6 / 458 steps/s against
        // 1 067 / 1 031 / 71.3 / 478 without — the corner tests cost what the skipped C2 halves save, the survivors' loop runs as
        // long as the lane with the most survivors, and eleven more live registers spill.
        double bx0 = pt[0].x, bx1 = pt[0].x, by0 = pt[0].y, by1 = pt[0].y;
#pragma unroll
        for (int i = 1; i < PDMPC_VMAX; ++i) {
            const bool in = i < ncols;
            bx0 = (in && pt[i].x < bx0) ? pt[i].x : bx0;
            bx1 = (in && pt[i].x > bx1) ? pt[i].x : bx1;
            by0 = (in && pt[i].y < by0) ? pt[i].y : by0;
            by1 = (in && pt[i].y > by1) ? pt[i].y : by1;
        }
        uint32_t surv = 0;
        {
            d2 c0 = q0;
            for (int t = 0; t < tn; ++t) {
                const d2 c1 = q[t + 1];
                const double dx2 = c1.x - c0.x, dy2 = c1.y - c0.y;
                const double S2 = dx2 * c0.y - dy2 * c0.x;
                const double yhi = dx2 >= 0 ? by1 : by0, ylo = dx2 >= 0 ? by0 : by1;
                const double xlo = dy2 >= 0 ? bx0 : bx1, xhi = dy2 >= 0 ? bx1 : bx0;
                const double emax = (yhi * dx2 - xlo * dy2) - S2, emin = (ylo * dx2 - xhi * dy2) - S2;
                const bool culled = emin > 0 || emax < 0;
                surv |= culled ? 0u : (1u << t);
                c0 = c1;
            }
        }
        while (surv && !(found & 1u)) {
            const int t = __builtin_ctz(surv);
            surv &= surv - 1u;
            asm volatile("" : "+v"(pt[0].x), "+v"(pt[0].y), "+v"(pt[1].x), "+v"(pt[1].y), "+v"(pt[2].x), "+v"(pt[2].y), "+v"(pt[3].x), "+v"(pt[3].y), "+v"(pt[4].x), "+v"(pt[4].y),
                         "+v"(pt[5].x), "+v"(pt[5].y), "+v"(pt[6].x), "+v"(pt[6].y), "+v"(pt[7].x), "+v"(pt[7].y));
            if (interx_segment_n<PDMPC_VMAX>(pt, ncols - 1, q[t], q[t + 1])) {
                const int rel = rel0 + t;
                const bool tent = rel >= 0 && ((pend >> (rel >> 3)) & 1ull) != 0ull;
                found |= tent ? 2u : 1u;
            }
        }
#else
        for (int t = 0; t < tn && !(found & 1u); ++t) {
            // (the area's points are made opaque per segment: the compiler would otherwise hoist the seven edges' dx1, dy1, S1 of the
            // C1 test out of this loop — 42 registers for a test that one segment in ten reaches)
            asm volatile("" : "+v"(pt[0].x), "+v"(pt[0].y), "+v"(pt[1].x), "+v"(pt[1].y), "+v"(pt[2].x), "+v"(pt[2].y), "+v"(pt[3].x), "+v"(pt[3].y), "+v"(pt[4].x), "+v"(pt[4].y),
                         "+v"(pt[5].x), "+v"(pt[5].y), "+v"(pt[6].x), "+v"(pt[6].y), "+v"(pt[7].x), "+v"(pt[7].y));
            const d2 q1 = q[t + 1];
            if (interx_segment_n<PDMPC_VMAX>(pt, ncols - 1, q0, q1)) {
                const int rel = rel0 + t;
                const bool tent = rel >= 0 && ((pend >> (rel >> 3)) & 1ull) != 0ull;  // (rel >> 3 < 64: a search with more predecessors waits for them all)
                found |= tent ? 2u : 1u;
            }
            q0 = q1;
        }
#endif
        if (found) __hip_atomic_fetch_or((lds_u32*)&r_flag[r], found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// the chunk size (as a shift) with which the check items of R entries fit the workgroup once, at most 16 segments per item
__device__ __forceinline__ int bk_chunk_shift(const lds_u32* chm, uint32_t R, uint32_t nthreads) {
    int ls = 0;
    while (ls < 4 && R * chm[ls] > nthreads) ++ls;
    return ls;
}

// position of the rk-th set bit of mask (rk < popcount)
__device__ __forceinline__ int nth_bit(uint64_t mask, int rk) {
    for (int b = 0; b < rk; ++b) mask &= mask - 1ull;
    return (int)__builtin_ctzll(mask);
}

// Late predecessors (PrioritizedController.m:476-491): the collision-free nodes of `list` against the areas of the predecessors in
// `arr`, which have just entered the soup, as far as those differ from the areas the edges were checked against (chg, see
// bk_incorporate_body).  item = (node, j): the j-th such predecessor of the node's step, j < maxc = the most any step has; the edge's
// area is transformed once, the predecessor's polygon of the node's step goes through interx_segment_n segment by segment
// (InterX.m:63-76 restricted to those polygons).
template <int CHECKER>
__device__ __forceinline__ void bk_recheck_items(const Search& S, const VState& VS, const BkCheck& C, const SpecCtx& P, const lds_u32* list, uint32_t count, unsigned long long arr,
                                                 const lds_u64s* chg, uint32_t maxc, volatile lds_u32* sh, int tid, int nthreads) {
    const uint32_t items = count * maxc;
    for (uint32_t item = (uint32_t)tid; item < items; item += (uint32_t)nthreads) {
        const uint32_t v = item / maxc, a = item - v * maxc;
        const uint32_t i0 = list[v];
        uint32_t parent, packed;
        piece_link(node_piece(S, i0, 3), parent, packed);
        if (!parent) continue;
        const int k = NODE_K(packed), m = NODE_MAN(packed), ncols = NODE_COLS(packed);
        const unsigned long long due = chg[k - 1] & arr;
        if (a >= (uint32_t)__builtin_popcountll(due)) continue;
        const int p = nth_bit(due, (int)a);
        const d2 pxy = node_piece(S, parent - 1u, 0), pcs = node_piece(S, parent - 1u, 2);
        const double cc = pcs.x, ss = pcs.y, pX = pxy.x, pY = pxy.y;
        const size_t abase = (size_t)m * 3 * PDMPC_VMAX;
        d2 pt[PDMPC_VMAX];
#pragma unroll
        for (int i = 0; i < PDMPC_VMAX; ++i) {
            const d2 ar = C.areas_in_lds ? (d2)C.l_area[abase + i] : C.g_area[abase + i];
            pt[i].x = cc * ar.x - ss * ar.y + pX;  // GraphSearch.m:158
            pt[i].y = ss * ar.x + cc * ar.y + pY;  // :159
        }
        const lds_d2* poly = P.l_soup + P.l_soff[k - 1] + P.l_lit[k - 1] + p * PDMPC_VMAX;
        d2 q0 = poly[0];
        bool hit = false;
        if (CHECKER == PDMPC_CHECK_SAT) {  // are_constraints_satisfied_sat.m:24-35 for this one dynamic obstacle
            int cols = 0;
            while (cols < PDMPC_VMAX && !is_nan(poly[cols].x)) ++cols;
            hit = cols > 0 && sat_pair_lane(pt, ncols, poly, cols);
        }
#pragma unroll 1
        for (int j = 0; CHECKER == PDMPC_CHECK_INTERX && j + 1 < PDMPC_VMAX; ++j) {
            asm volatile("" : "+v"(pt[0].x), "+v"(pt[0].y), "+v"(pt[1].x), "+v"(pt[1].y), "+v"(pt[2].x), "+v"(pt[2].y), "+v"(pt[3].x), "+v"(pt[3].y), "+v"(pt[4].x), "+v"(pt[4].y),
                         "+v"(pt[5].x), "+v"(pt[5].y), "+v"(pt[6].x), "+v"(pt[6].y), "+v"(pt[7].x), "+v"(pt[7].y));  // (as in bk_check_items)
            const d2 q1 = poly[j + 1];
            hit = hit || interx_segment_n<PDMPC_VMAX>(pt, ncols - 1, q0, q1);
            q0 = q1;
        }
        if (hit) {
            vs_store(VS, i0, VS_INVALID);
            atomicOr((uint32_t*)&sh[FR_FLAGS], FRF_INVALIDATED);
        }
    }
}

// monotone map key -> bin of a linear histogram of BK_NB bins over [lo, lo + BK_NB / scale)
#define BK_NB 256
__device__ __forceinline__ uint32_t bk_bin(double key, double lo, double scale) {
    const double t = (key - lo) * scale;
    if (!(t > 0.0)) return 0u;
    return t < (double)(BK_NB - 1) ? (uint32_t)t : (uint32_t)(BK_NB - 1);
}
// The first bin at which the cumulative count of the BK_NB-bin histogram reaches `target` (the last non-empty bin if the total is
// smaller) and that count.  Every lane of a wave calls (four bins per lane); every wave of the workgroup does it for itself, so
// the result needs no broadcast through LDS and no barrier.
__device__ __forceinline__ void bk_select(const lds_u32* bins, uint32_t target, int lane, uint32_t& bin, uint32_t& cum) {
    uint32_t hq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) hq[q] = bins[4 * lane + q];
    const uint32_t loc = hq[0] + hq[1] + hq[2] + hq[3];
    uint32_t inc = loc;
#pragma unroll
    for (int o = 1; o < PDMPC_WAVE; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, o);
        inc += lane >= o ? v : 0u;
    }
    const uint32_t total = lane_u(inc, PDMPC_WAVE - 1);
    const uint32_t want = target < total ? target : total;
    uint32_t c = inc - loc, b = 0, cu = 0;
    bool f = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        c += hq[q];
        const bool hit = !f && hq[q] != 0u && c >= want;
        b = hit ? (uint32_t)(4 * lane + q) : b;
        cu = hit ? c : cu;
        f = f || hit;
    }
    const unsigned long long m = __ballot(f);
    const int l = m ? (int)__builtin_ctzll(m) : 0;
    const uint32_t bb = lane_u(b, l), cc = lane_u(cu, l);
    bin = m ? bb : 0u;
    cum = m ? cc : 0u;
}

// The areas of a result record and their column counts — all a successor reads of it (PrioritizedController.m:476-491) — go to memory
// with agent-scope stores and are read with agent-scope loads: coherent across the XCDs' L2s by themselves.  PDMPC_BK_AREA_FENCES=1
// (build switch) puts the release / acquire fences of rounds 2-4 back around them (a write-back / invalidation of the whole L2).
#ifndef PDMPC_BK_AREA_FENCES
#define PDMPC_BK_AREA_FENCES 0
#endif
// The records a shared round posts for its helpers and everything else owner and helpers exchange likewise go through agent-scope
// stores and loads; PDMPC_BK_POST_FENCES=1 (build switch) posts with plain stores behind a release fence and reads behind an acquire
// fence instead, as rounds 2-4 did (every fence writes back / invalidates the whole L2 of its XCD: with a helper on every idle CU
// that is every L2 of the chip, once per round).
#ifndef PDMPC_BK_POST_FENCES
#define PDMPC_BK_POST_FENCES 0
#endif
__device__ __forceinline__ void bk_post_store(d2* p, d2 v) {
#if PDMPC_BK_POST_FENCES
    *p = v;
#else
    __hip_atomic_store((double*)p, (double)v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((double*)p + 1, (double)v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ d2 bk_post_load(const d2* p) {
#if PDMPC_BK_POST_FENCES
    return *p;
#else
    d2 v;
    v.x = __hip_atomic_load((const double*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v.y = __hip_atomic_load((const double*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
#endif
}
__device__ __forceinline__ void bk_area_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void bk_area_store_u32(int32_t* p, int v) { __hip_atomic_store(p, (int32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double bk_area_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int bk_area_load_i32(const int32_t* p) { return (int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The result record (GraphSearch.m:58-59, 82-89; return_path_to.m; return_path_area.m), written by the first wave.  A search that
// has finished while predecessors are still planning writes it right away: an arrival that invalidates nothing leaves it as it
// is, and the publication that follows the last arrival is one flag.  `again`: a record of this search was written before
// (its fields are reset first).  l_path[i] = node (1-based index into this vehicle's arena) of step i along the selected path
// (walked here unless path_ready); ref_ids = the ids those nodes carry in the reference's tree (info.tree_path).
__device__ __forceinline__ void bk_write_record(const KernelArgs& A, Ctx& X, uint32_t goal, int status, bool dep_timeout, uint32_t n_popped, uint32_t nnodes, bool path_ready,
                                                const lds_u32* ref_ids, bool again, bool counts_only, int lane) {
    const int Hp = X.Hp;
    const DevVehicle* __restrict__ V = X.V;
    pdmpc_vehicle_out* __restrict__ O = X.O;
    lds_u32* l_path = X.l_path;
    const Search& S = X.S;
    if (counts_only) {
        // The areas of this record are out already (the done flag was set when the last predecessor's areas had passed the path,
        // bk_wait_done) and successors may be reading them: only what the re-check of the other nodes can still change is written
        // — the reference's ids along the path and the two counts (the path itself is the one that was published).
        if (goal && ref_ids && lane <= Hp) O->tree_path[lane] = (int32_t)ref_ids[lane];
        if (lane == 0) {
            O->status = dep_timeout ? PDMPC_ERR_HIP : status;
            O->n_expanded = (int32_t)nnodes;
            O->n_popped = (int32_t)n_popped;
        }
        return;
    }
    lds_d2* pshape = (lds_d2*)(X.lsm + PDMPC_LK_PSHAPE);
    lds_u32* pcols = (lds_u32*)(pshape + Hp * PDMPC_VMAX);
    if (again) {  // as the prologue left it: zeros, y_predicted NaN (ControlResultsInfo.m:40)
        double* od = (double*)O;
        const int nd = (int)(offsetof(pdmpc_vehicle_out, path_nodes) / 8) + (Hp + 1) * 8;  // (rows beyond the path are never written here: the diagnostics of the tail stay)
        const int y0 = (int)(offsetof(pdmpc_vehicle_out, y_predicted) / 8);
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        for (int i = lane; i < nd; i += PDMPC_WAVE) od[i] = (i >= y0 && i < y0 + PDMPC_HP_MAX * 3) ? qnan : 0.0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the result stores below hit the same bytes from other lanes
    }
    if (goal) {
        if (lane == 0 && !path_ready) {  // path_to_root (Tree.m:44-52), reversed
            uint32_t nd = goal;
            for (int i = Hp; i >= 0; --i) {
                l_path[i] = nd;
                nd = node_parent(S, nd - 1);
            }
        }
        wave_sync();
        if (lane <= Hp) {
            const uint32_t nd = l_path[lane];
            const NodeRec r = node_load(S, nd - 1);
            O->tree_path[lane] = (int32_t)(ref_ids ? ref_ids[lane] : nd);
            double* row = O->path_nodes[lane];  // NodeInfo.m:5-13
            row[0] = r.x;
            row[1] = r.y;
            row[2] = r.yaw;
            row[3] = (double)NODE_TRIM(r.packed);
            row[4] = r.g;
            row[5] = r.h;
            row[6] = (double)NODE_K(r.packed);
            row[7] = 1.0;
            if (lane >= 1) {
                O->y_predicted[lane - 1][0] = r.x;
                O->y_predicted[lane - 1][1] = r.y;
                O->y_predicted[lane - 1][2] = r.yaw;
                O->predicted_trims[lane - 1] = (int32_t)NODE_TRIM(r.packed);
            }
        }
        // shapes along the path: same arithmetic as at pop time (GraphSearch.m:158-160), so the same bits
        for (int idx = lane; idx < Hp * PDMPC_VMAX; idx += PDMPC_WAVE) {
            const int i = idx / PDMPC_VMAX + 1;
            const int v = idx - (i - 1) * PDMPC_VMAX;
            const NodeRec pr = node_load(S, l_path[i - 1] - 1);
            const NodeRec cr = node_load(S, l_path[i] - 1);
            const int m = NODE_MAN(cr.packed);
            const int ncols = NODE_COLS(cr.packed);
            // (the areas and their column counts are what successors read, bk_area_load: stores that go through to memory, so that
            // the publication needs no write-back of this XCD's L2)
            if (v == 0) {
                bk_area_store_u32(&O->shape_cols[i - 1], ncols);
                pcols[i - 1] = (uint32_t)ncols;
            }
            d2 sp = d2{0.0, 0.0};
            if (v < ncols) {
                const d2 a = X.C.g_area[(size_t)m * 3 * PDMPC_VMAX + v];
                sp.x = pr.cs * a.x - pr.sn * a.y + pr.x;
                sp.y = pr.sn * a.x + pr.cs * a.y + pr.y;
                bk_area_store(&O->shapes[i - 1][0][v], sp.x);
                bk_area_store(&O->shapes[i - 1][1][v], sp.y);
            }
            pshape[idx] = sp;  // (columns beyond ncols are padding: read, never used)
        }
    } else if (V->fb_off[0] >= 0) {
        // exhausted: publish the caller-supplied fallback areas so successors of this launch avoid them (PrioritizedController.m:568-616, 678-718)
        for (int idx = lane; idx < Hp * PDMPC_VMAX; idx += PDMPC_WAVE) {
            const int k = idx / PDMPC_VMAX;
            const int v = idx - k * PDMPC_VMAX;
            const int a = V->fb_off[k], b = V->fb_off[k + 1];
            const int cols = (b - a < PDMPC_VMAX) ? (b - a) : PDMPC_VMAX;
            if (v == 0) bk_area_store_u32(&O->shape_cols[k], cols);
            if (v < cols) {
                bk_area_store(&O->shapes[k][0][v], A.points[2 * (size_t)(a + v)]);
                bk_area_store(&O->shapes[k][1][v], A.points[2 * (size_t)(a + v) + 1]);
            }
        }
    }
    if (lane == 0) {
        O->status = dep_timeout ? PDMPC_ERR_HIP : status;
        O->n_expanded = (int32_t)nnodes;
        O->n_popped = (int32_t)n_popped;
        O->n_hp = Hp;
    }
}

// Publication: plain stores -> this wave's vmcnt(0) -> lane-0 agent release -> flag.  First wave.
// (the areas were stored through to memory, bk_area_store: what the flag announces is there once this wave's stores have been
// acknowledged; the rest of the record is read by the host, after the kernel)
__device__ __forceinline__ void bk_publish_flag(const KernelArgs& A, int slot, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_sync();
    if (lane == 0) {
#if PDMPC_BK_AREA_FENCES
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __hip_atomic_store(A.done_flag + slot, A.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void bk_publish(const KernelArgs& A, const Ctx& X, int status, bool dep_timeout) {
    if (X.lane == 0 && (dep_timeout || (status != PDMPC_OK && status != PDMPC_EXHAUSTED))) atomicAdd(A.work_count + 6, 1ull);  // (device-side tally of plans that are not planning results)
    bk_publish_flag(A, X.slot, X.lane);
}

// One look at the done flags of the predecessors in `want` (poll_predecessors without its bookkeeping): the set that has finished.
// Whole wave calls.
__device__ __forceinline__ unsigned long long bk_poll_flags(const KernelArgs& A, const SpecCtx& P, unsigned long long want, int lane) {
    bool d = false;
    if ((want >> lane) & 1ull) d = __hip_atomic_load(A.done_flag + P.pred[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == A.epoch;
    const unsigned long long got = __ballot(d);
#if PDMPC_BK_AREA_FENCES
    if (got) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#endif
    return got;
}

// poll_predecessors on bk_poll_flags: the pending predecessors that have finished are posted in SH_ARR and the state goes
// ST_RUN -> ST_ARRIVED.  Whole wave calls.
__device__ __forceinline__ bool bk_poll_predecessors(const KernelArgs& A, const SpecCtx& P, volatile lds_u32* sh, unsigned long long skip, int lane) {
    const unsigned long long pend = sh_load64(sh, SH_PEND_LO) & ~skip;
    if (!pend) return false;
    const unsigned long long got = bk_poll_flags(A, P, pend, lane);
    if (!got) return false;
    if (lane == 0) {
        sh[SH_ARR_LO] = (uint32_t)got;
        sh[SH_ARR_HI] = (uint32_t)(got >> 32);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        atomicCAS((uint32_t*)&sh[SH_STATE], ST_RUN, ST_ARRIVED);
    }
    return true;
}

// incorporate_areas with loads that are coherent by themselves (bk_area_load): the solved areas of the predecessors in `arr` into
// their soup slots (PrioritizedController.m:476-491); nthreads threads call (a workgroup, or one wave).
// (chg, when given: [Hp] 64-bit masks — bit p of chg[k] is set when the area predecessor p publishes for step k + 1 differs from what
// its slot held, i.e. from the area it was expected to take.  A collision-free edge of that step has been checked against exactly
// those numbers: only the pairs (step, predecessor) marked here are due for the re-check, bk_recheck_items.)
__device__ __forceinline__ void bk_incorporate_body(const pdmpc_vehicle_out* out, const int32_t* pred, lds_d2* l_soup, const lds_i32* l_soff, const lds_i32* l_lit, int Hp, unsigned long long arr, int tid,
                                                    int nthreads, lds_u64s* chg) {
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    const int per = Hp * PDMPC_VMAX, n_arr = __builtin_popcountll(arr);
    for (int idx = tid; idx < n_arr * per; idx += nthreads) {  // (every arrived predecessor's loads side by side)
        const int a = idx / per, rest = idx - a * per;
        const int p = nth_bit(arr, a);
        const pdmpc_vehicle_out* PO = out + pred[p];
        const int k = rest / PDMPC_VMAX, v = rest - k * PDMPC_VMAX;
        const int cols = bk_area_load_i32(&PO->shape_cols[k]);
        const double sx = bk_area_load(&PO->shapes[k][0][v]), sy = bk_area_load(&PO->shapes[k][1][v]);
        d2 pt;
        pt.x = v < cols ? sx : qnan;
        pt.y = v < cols ? sy : qnan;
        lds_d2* slot = l_soup + l_soff[k] + l_lit[k] + p * PDMPC_VMAX + v;
        if (chg) {
            const d2 was = *slot;
            if (__double_as_longlong(was.x) != __double_as_longlong(pt.x) || __double_as_longlong(was.y) != __double_as_longlong(pt.y))
                __hip_atomic_fetch_or(chg + k, 1ull << p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        *slot = pt;
    }
}
__device__ __forceinline__ void bk_incorporate(const pdmpc_vehicle_out* out, const int32_t* pred, lds_d2* l_soup, const lds_i32* l_soff, const lds_i32* l_lit, int Hp, unsigned long long arr, int tid,
                                               int nthreads, lds_u64s* chg) {
    bk_incorporate_body(out, pred, l_soup, l_soff, l_lit, Hp, arr, tid, nthreads, chg);
}

// fr_check_wave with a memory: what to do with an open node a round has selected (1 process it, 3 it comes after the goal candidate,
// 4 one of its ancestors lost its edge to late areas), one node per lane, a = 0: none.
// The answer follows from the branch between the node and the candidate's path — where it joins the path (depth dJ), the largest key on
// it (M), whether a node on it is invalid — and a child's branch is its parent's plus itself.  Every processed node leaves these three
// in walk[] (16 bytes per node), stamped with the candidate and the number of arrival events they hold for; a node whose parent carries
// a current stamp is classified with two reads instead of a walk to the path (up to Hp dependent reads of three arrays each, which for a
// heavy search with a candidate cost as much as a third of its checks: C4).  Nodes whose parents were processed under another
// candidate, or before the last arrival, walk as before and leave their own stamp.
#define WC_DEAD 0x100u
#define WC_ONPATH 0x200u
#define WC_VALID 0x400u
__device__ int bk_classify_wave(const unsigned long long* glink, ulonglong2* walk, const VState& VS, const double* gkey, const lds_u32* gp_path, const lds_f64* gp_mp, bool have_goal,
                                bool check_alive, uint32_t best, uint32_t epoch, uint32_t a, double ka, volatile lds_u32* sh) {
    const uint32_t a0 = a ? a : 1u;
    const uint64_t ua = glink[a0 - 1u];
    const uint32_t p = (uint32_t)(ua & 0xffffffffull);
    const int k = NODE_K((uint32_t)(ua >> 32));
    const bool onp = have_goal && gp_path[k] == a0;
    const ulonglong2 c = walk[p ? p - 1u : 0u];
    const uint32_t ctag = (uint32_t)(c.y & 0xffffffffull), cinfo = (uint32_t)(c.y >> 32);
    const bool hit = p != 0u && ctag == best && (cinfo & (WC_VALID | 0xffff0000u)) == (WC_VALID | (epoch << 16));
    double M = -1.0;
    int dJ = k;
    bool dead = false;
    // ---- no current stamp at the parent: the walk (the whole wave together, see fr_check_wave)
    int st = (a == 0u || onp || hit) ? 1 : 0;  // 0: still walking
    uint32_t x = a0;
    while (__ballot(st == 0)) {
        const uint32_t i = x - 1u;
        const uint64_t u = glink[i];
        const int d = NODE_K((uint32_t)(u >> 32));
        const bool on_path = have_goal && gp_path[d] == x;
        const bool bad = !on_path && x != a0 && check_alive && vs_load(VS, i) != VS_VALID;
        const double kx = gkey[i];
        const uint32_t par = (uint32_t)(u & 0xffffffffull);
        const bool walking = st == 0;
        const bool stop = on_path || bad || par == 0u;
        dJ = (walking && on_path) ? d : dJ;
        dead = dead || (walking && bad);
        M = (walking && !on_path && !bad && kx > M) ? kx : M;  // (the nodes below the path: a itself, then its ancestors)
        st = (walking && stop) ? 1 : st;
        x = (walking && !stop) ? par : x;
    }
    if (hit && !onp) {
        const bool p_on = (cinfo & WC_ONPATH) != 0u;
        const double pM = __longlong_as_double((long long)c.x);
        const uint32_t vsp = check_alive && !p_on ? vs_load(VS, p - 1u) : (uint32_t)VS_VALID;
        dead = !p_on && ((cinfo & WC_DEAD) != 0u || vsp != VS_VALID);
        M = (!p_on && pM > ka) ? pM : ka;
        dJ = (int)(cinfo & 0xffu);
    }
    // (a node on the path: its children join the path at it)
    const double thr = gp_mp[dJ];
    int res = 1;
    if (a != 0u && !onp) {
        // (equal maxima: the order depends on the reference's binary heap — the node is processed, so that the replay of a tied search
        // finds everything the heap can pop evaluated, and the search is marked: FRF_TIE -> BK_TIEMODE)
        res = dead ? 4 : ((have_goal && M > thr) ? 3 : 1);
        if (!dead && have_goal && M == thr) atomicOr((uint32_t*)&sh[FR_FLAGS], FRF_TIE);
    }
    if (a != 0u && res == 1) {  // (it will be processed: its children look here)
        ulonglong2 w;
        w.x = (unsigned long long)__double_as_longlong(onp ? -1.0 : M);
        w.y = (unsigned long long)best | ((unsigned long long)((uint32_t)(onp ? k : dJ) | (onp ? WC_ONPATH : 0u) | WC_VALID | (epoch << 16)) << 32);
        walk[a0 - 1u] = w;
    }
    return res;
}

// A FINISHED search waiting for its predecessors (its record is written, bk_write_record; the areas along its path are in LDS).
// What its successors wait for are its areas, and those are final as soon as every predecessor's areas have passed the path: nothing
// open or parked comes before the goal (that is what finished means), and an arrival can only take edges away.  So the first wave
// alone — no workgroup barrier — polls, copies an arrived predecessor's areas into the soup, checks them against the Hp edges of
// the path (the arithmetic of bk_recheck_items on the same numbers) and sets the done flag when the last predecessor has passed.
// The verification of the other collision-free nodes, which can only change the counts and ids of the record, follows at the next
// round boundary for all of them at once (BK_FD: who is due).  A predecessor that crosses the path ends the wait: the verification
// takes the path's edge away and the search resumes.  `first`: predecessors whose areas are in the soup already but have not been
// checked against this path (copied while the search was running); max_spins 0: look at them and at the flags once, do not wait.
// Leaves BK_WAITRES (0 nothing decided, 1 an arrival crosses the path, 2 published), SH_PEND and BK_FD.
__device__ __forceinline__ void bk_wait_done(const uint32_t* done_flag, const int32_t* pred, const pdmpc_vehicle_out* out, uint32_t epoch, uint32_t slot, lds_d2* l_soup, const lds_i32* l_soff,
                                             const lds_i32* l_lit, lds_d2* pshape, volatile lds_u32* sh, int Hp, int n_pred, bool have_path, lds_vu64* tk_pub, unsigned long long first, uint32_t max_spins,
                                             bool sat, int lane) {
    const lds_u32* pcols = (const lds_u32*)(pshape + Hp * PDMPC_VMAX);
    lds_u64s* chg = (lds_u64s*)(pcols + PDMPC_HP_MAX);
    unsigned long long pend = sh_load64(sh, SH_PEND_LO), fd = sh_load64(sh, BK_FD_LO);
    unsigned long long got = first;  // (in the soup already: checked against the path before anybody is polled)
    uint32_t spins = 0, res = 0;
    for (;;) {
        if (!got && pend != 0ull) {  // (uniform; nobody outstanding and nothing to check: straight to the flag)
            bool d = false;
            if (lane < n_pred && ((pend >> lane) & 1ull)) d = __hip_atomic_load(done_flag + pred[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
            got = __ballot(d);
            if (!got) {
                if (++spins >= max_spins) break;
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
#if PDMPC_BK_AREA_FENCES
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            bk_incorporate_body(out, pred, l_soup, l_soff, l_lit, Hp, got, lane, PDMPC_WAVE, chg);
            wave_sync();
            pend &= ~got;
            fd |= got;
        }
        const int n_got = __builtin_popcountll(got);
        const int per = Hp * (PDMPC_VMAX - 1), items = have_path ? n_got * per : 0;
        bool hit = false;
        for (int base = 0; base < items; base += PDMPC_WAVE) {  // (uniform trip count; item = (arrived predecessor, step, segment of its area))
            const int item = base + lane;
            const bool in = item < items;
            const int it = in ? item : 0;
            const int a = it / per, rest = it - a * per;
            const int k0 = rest / (PDMPC_VMAX - 1), j = rest - k0 * (PDMPC_VMAX - 1);
            const int p = nth_bit(got, a);
            d2 pt[PDMPC_VMAX];
#pragma unroll
            for (int i = 0; i < PDMPC_VMAX; ++i) pt[i] = pshape[k0 * PDMPC_VMAX + i];
            const lds_d2* poly = l_soup + l_soff[k0] + l_lit[k0] + p * PDMPC_VMAX;
            bool h1;
            if (!((chg[k0] >> p) & 1ull)) {  // the area the path was checked against when its edge was evaluated: nothing new
                h1 = false;
            } else if (sat) {  // (uniform) the separating-axis checker: the lane with the area's first segment tests the pair of polygons
                int cols = 0;
                while (cols < PDMPC_VMAX && !is_nan(poly[cols].x)) ++cols;
                h1 = j == 0 && cols > 0 && sat_pair_lane(pt, (int)pcols[k0], poly, cols);
            } else {
                h1 = interx_segment_n<PDMPC_VMAX>(pt, (int)pcols[k0] - 1, poly[j], poly[j + 1]);
            }
            hit = hit || (in && h1);
        }
        got = 0ull;
        if (__ballot(hit)) {  // (uniform)
            res = 1;
            break;
        }
        if (pend == 0ull) {
            res = 2;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_sync();
            if (lane == 0) {
#if PDMPC_BK_AREA_FENCES
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                __hip_atomic_store((uint32_t*)done_flag + slot, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tk_pub) *tk_pub = __builtin_amdgcn_s_memrealtime();
            }
            break;
        }
        if (max_spins == 0u) break;
    }
    if (lane == 0) {
        sh[SH_PEND_LO] = (uint32_t)pend;
        sh[SH_PEND_HI] = (uint32_t)(pend >> 32);
        sh[BK_FD_LO] = (uint32_t)fd;
        sh[BK_FD_HI] = (uint32_t)(fd >> 32);
        sh[BK_WAITRES] = res;
        if (res == 2u) sh[BK_PUBLISHED] = 1u;
        sh[BK_IDLE] = sh[BK_IDLE] + spins + 1u;
    }
}

// Collision-free nodes at the horizon are offered as goal candidates again (after an invalidation that took the best one, or
// when a round had more candidates than its list holds): every lane with a candidate walks that candidate's path by itself — all
// of them Hp edges long, so the lanes of a wave finish together — instead of the wave walking them one after the other
// (fr_offer_goal; a walk is Hp dependent reads of the tree: C2's last vehicles spent 30 us per arrival there).  Whole wave calls.
__device__ __forceinline__ void bk_offer_goals_lanes(const Frontier& F, const Search& S, const VState& VS, bool cand, uint32_t id, int lane) {
    double m = 0.0;
    bool alive = cand;
    if (cand) {
        uint32_t nd = id;
        for (;;) {
            const double k = F.gkey[nd - 1];
            m = k > m ? k : m;
            if (nd != id && vs_load(VS, nd - 1) != VS_VALID) alive = false;
            const uint32_t par = node_parent(S, nd - 1);
            if (!par) break;
            nd = par;
        }
    }
    const unsigned long long bal = __ballot(alive);
    if (bal) {  // (uniform)
        const uint32_t base = sh_add_uniform(F.sh, FR_GOAL_N, (uint32_t)__builtin_popcountll(bal), lane);
        if (alive) {
            node_store_cs(S, id - 1u, m, 0.0);
            const uint32_t pos = base + lane_rank(bal, lane);
            if (pos < 1024u) F.goal_list[pos] = id;  // (at most blockDim candidates per pass: cannot overflow)
        }
    }
}

// The far-list selection stays an out-of-line call: it runs once per refill on one wavefront, and inlined into the search loop its
// registers push the whole kernel over the 168-VGPR budget of a twelve-wavefront workgroup (tests/test_build.py watches this).
__device__ __noinline__ void bk_far_select(const Frontier& F, uint32_t fill, int lane) {
    fr_select2(F, fill, fill, FR_SEL_BIN, FR_SEL_BIN, lane);
}

// The end of a search that met equal keys where the pop order decides (GraphSearch.m:53-107 on priority_queue_interface_mex.cpp:19-31:
// which of two equal keys std::priority_queue pops first follows from the layout of its binary heap, not from any property of the
// nodes).  The rounds have evaluated every node the heap can possibly pop before the goal; here ONE wavefront pops that tree once more
// through the libstdc++-faithful heap (heap_queue.hpp: entries = (arena index, key), LDS for the first `hl` entries, the far list's
// arrays beyond), in the reference's own sequence: pop, discard what collides (GraphSearch.m:75-77), stop at the first collision-free
// node at the horizon (:81-90), else push the children in ascending trim order (expand_node.m:18, mex.cpp:67-72) — validity and
// children are looked up, nothing is computed.  Leaves the goal, n_popped, the tree size, the reference's ids along the path (ref_ids,
// X.l_path), per node its reference id (node2ref) and the pop sequence (pops) — or, in BK_RP_NEED, a node the heap popped that no
// round has evaluated (dropped on account of a candidate that tied with the goal): the caller gives it a round and replays again.
template <int NW>
__device__ __forceinline__ void bk_replay(const KernelArgs& A, Ctx& X, const Frontier& F, const ExpandEnv& EE, lds_f64* heap_key, lds_u32* heap_id, uint32_t hl, uint32_t* pops, uint32_t* node2ref,
                                          const uint32_t* child0, lds_u32* ref_ids, int lane) {
    volatile lds_u32* sh = F.sh;
    const VState& VS = X.VS;
    const int Hp = X.Hp, n = EE.n, nw = NW > 0 ? NW : EE.nw;
    Search H = X.S;
    H.lkey = heap_key;
    H.lid = heap_id;
    H.HL = hl & ~1u;
    H.gkey = F.far_key;
    H.gid = F.far_id;
    H.heap_len = 0;
    H.lane = lane;
    H.pl = make_pop_lane(lane);
    heap_push(H, 0u, 0.0);  // pq.push(1, 0) (GraphSearch.m:45-46); the entry's id is the node's arena index
    if (lane == 0) node2ref[0] = 1u;
    uint32_t n_ref = 1, n_pop = 0, goal = 0, need = 0;
    bool bug = false;
    for (;;) {
        if (H.heap_len == 0u) break;  // exhausted (:57-61)
        double k0;
        uint32_t b;
        heap_load<true>(H, 0u, true, k0, b);
        b = uni_u(b);
        heap_pop(H);
        if (lane == 0) pops[n_pop] = b;
        n_pop += 1u;
        const uint32_t v = uni_u(vs_load(VS, b));
        if (v == VS_UNKNOWN || v == VS_TENT) {
            need = b + 1u;
            break;
        }
        if (v != VS_VALID) continue;  // :75-77
        const uint64_t u = F.glink[b];
        const uint32_t packed = uni_u((uint32_t)(u >> 32));
        const int k = NODE_K(packed);
        if (k == Hp) {  // :81-90
            goal = b + 1u;
            break;
        }
        const lds_mask64* mrow = EE.l_mask + ((size_t)k * n + (NODE_TRIM(packed) - 1)) * nw;
        uint32_t cnt = 0;
        for (int w = 0; w < nw; ++w) cnt += (uint32_t)__builtin_popcountll(mrow[w]);
        cnt = uni_u(cnt);
        if (cnt == 0u) continue;
        const uint32_t c0 = uni_u(__hip_atomic_load(child0 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (c0 == 0u) {  // (collision-free and never expanded: must never happen)
            bug = true;
            break;
        }
        for (uint32_t base = 0; base < cnt; base += PDMPC_WAVE) {  // (uniform) the children, ascending trim = ascending index (Tree.m:61)
            const uint32_t m = cnt - base < PDMPC_WAVE ? cnt - base : PDMPC_WAVE;
            const uint32_t c = c0 - 1u + base + (uint32_t)lane;
            double kc = 0.0;
            if ((uint32_t)lane < m) {
                kc = F.gkey[c];
                node2ref[c] = n_ref + 1u + base + (uint32_t)lane;
            }
            for (uint32_t i = 0; i < m; ++i) heap_push(H, lane_u(c, (int)i), lane_d(kc, (int)i));
        }
        n_ref += cnt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_sync();
    if (goal && lane == 0) {  // path_to_root (Tree.m:44-52) in arena indices and in the reference's ids
        uint32_t nd = goal;
        for (int i = Hp; i >= 0; --i) {
            X.l_path[i] = nd;
            ref_ids[i] = __hip_atomic_load(node2ref + (nd - 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            nd = (uint32_t)(F.glink[nd - 1u] & 0xffffffffull);
        }
    }
    if (lane == 0) {
        sh[BK_RP_NEED] = need;
        sh[BK_RP_GOAL] = goal;
        sh[BK_RP_NPOP] = n_pop;
        sh[BK_RP_NREF] = n_ref;
        if (bug) sh[FR_FLAGS] = sh[FR_FLAGS] | FRF_BUG;
        if (!need && !bug) atomicAdd(X.P.counters + 0, 1);  // (pdmpc_stats.queue_fallbacks: searches that ended on the binary heap)
    }
}

// The search (results in X).  The return value is unused (always false: equal keys are resolved inside the search, bk_replay).
template <int NW, int CHECKER>
__device__ __forceinline__ bool bulk_search(const KernelArgs& A, Ctx& X, lds_u32* ref_ids) {
    const int tid = X.tid, lane = X.lane, wave = X.wave, slot = X.slot, Hp = X.Hp;
    volatile lds_u32* sh = X.l_shared;
    Search& S = X.S;
    const VState& VS = X.VS;
    const SpecCtx& P = X.P;
    const DevVehicle* __restrict__ V = X.V;
    const int n_waves = (int)(blockDim.x >> 6), bd = (int)blockDim.x;
    const double inf = __longlong_as_double(0x7FF0000000000000LL);
    const size_t voff = (size_t)slot * A.max_nodes;
    const int n = X.n, nw = NW > 0 ? NW : X.nw;
    const uint32_t OC = (uint32_t)(BK_PER * bd), RC = (uint32_t)A.bk_ready_cap;
    const uint32_t VCAP = 1024u;  // expansion groups per tile (vlist / voffs)

    // ---- LDS carve of the bulk region
    lds_f64* near_key = (lds_f64*)(X.lsm + PDMPC_LK_NEAR_KEY);
    lds_u32* near_id = (lds_u32*)(X.lsm + PDMPC_LK_NEAR_ID);
    lds_u32* ready = (lds_u32*)(X.lsm + PDMPC_LK_READY);
    volatile lds_u32* r_flag = (volatile lds_u32*)(ready + RC);
    lds_u32* hist = (lds_u32*)(X.lsm + PDMPC_LK_HIST);      // [3072]: refill histogram [2048] | goal list [1024], expansion groups [1024], their children's offsets [1024]
    lds_u32* vlist = hist + 1024;
    lds_u32* voffs = hist + 2048;
    lds_u32* gp_path = (lds_u32*)(X.lsm + PDMPC_LK_MISC);    // [32] path of the best goal candidate
    lds_f64* gp_mp = (lds_f64*)(gp_path + 32);                // [HP_MAX + 1] largest key of that path below depth d
    lds_vu64* wsum64 = (lds_vu64*)(gp_mp + 32);                // [32] scan partials
    volatile lds_u32* wsum = (volatile lds_u32*)(wsum64 + 32); // [32] fr_partition's per-wave counts
    lds_u32* chm = (lds_u32*)(wsum + 32);                      // [8] chunks per node for S = 1, 2, 4, 8, 16, ...
    lds_u32* bins = gp_path + 256;                             // [BK_NB] the selection's histogram (second KB of the region)

    Frontier F;
    F.sh = sh;
    F.ready = ready;
    F.hist = hist;
    F.goal_list = hist;
    F.near_key = A.arena.pb_key + voff;  // (HBM arrays: phase B's per-node state; near itself lives in LDS)
    F.near_id = A.arena.pb_d + voff;
    F.far_key = A.arena.far_key + voff;
    F.far_id = A.arena.far_id + voff;
    F.gkey = S.gkey;
    F.glink = A.arena.link + voff;
    F.n_waves = n_waves;

    ExpandEnv EE;
    EE.l_mask = X.l_mask;
    EE.l_mi = X.l_mi;
    EE.l_pose = X.l_pose;
    EE.l_rx = X.l_rx;
    EE.l_ry = X.l_ry;
    EE.l_dcum = X.l_dcum;
    EE.l_term = nullptr;
    EE.l_chxy = nullptr;
    EE.Hp = Hp;
    EE.n = X.n;
    EE.nw = X.nw;
    EE.lane = lane;

    BkCheck CK;
    CK.l_area = X.C.l_area;
    CK.g_area = X.C.g_area;
    CK.l_soup = X.C.l_soup;
    CK.l_soff = X.C.l_soff;
    CK.l_hoff = X.C.l_hoff;
    CK.l_lit = P.l_lit;
    CK.areas_in_lds = X.C.areas_in_lds;
    CK.ll_base = X.C.ll_base;
    CK.ll_len = X.C.ll_len;
    CK.Hp = Hp;

    // ---- root node (GraphSearch.m:34-46) and the chunk tables
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
        ((ulonglong2*)A.arena.walk + voff)[0].y = 0ull;
        (A.arena.child0 + voff)[0] = 0u;
        vs_store(VS, 0, VS_UNKNOWN);
        for (int w = 26; w < SH_WORDS; ++w) sh[w] = 0;
        sh[FR_NNODES] = 1;
        sh_st_d(sh, FR_NEAR_MIN, inf);
        sh_st_d(sh, FR_FAR_MIN, inf);
        sh_st_d(sh, FR_L_FAR, inf);
        sh_st_d(sh, BK_TENT_MIN, inf);
        sh[BK_MID_N] = 0;
        sh[BK_MID_N + 1] = 0;
        sh_st_d(sh, BK_MID_MIN, inf);
        sh_st_d(sh, BK_L_MID, -1.0);
        sh[SH_NNODES] = 1;
        sh[BK_TIEMODE] = A.bk_force_tie ? 1u : 0u;
        ready[0] = 1u;
        r_flag[0] = 0u;
    }
    if (tid < BK_NB) bins[tid] = 0u;
    if (A.n_helpers > 0 && tid >= 64 && tid < 64 + PDMPC_HB_SEATS_MAX) {  // this search's board: no seats, no assignments, nothing reported (the words may hold an earlier launch's)
        unsigned long long* bd0 = A.help_board + (size_t)slot * PDMPC_HB_WORDS;
        const int k = tid - 64;
        __hip_atomic_store(bd0 + PDMPC_HB_ASSIGN + k, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(bd0 + PDMPC_HB_DONE + k, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == 0) {
            __hip_atomic_store(bd0 + PDMPC_HB_WANT, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(bd0 + PDMPC_HB_SEATS, (unsigned long long)A.launch_id << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (A.bk_tentative) bk_tentative_areas(A, P, sh_load64(sh, SH_PEND_LO), tid, (int)blockDim.x);  // (the pending set was fixed by the prologue)
    if (tid >= 64 && tid < 72) {
        const int ls = tid - 64;
        uint32_t mx = 0;
        for (int k = 1; k <= Hp; ++k) {
            const int M_k = CK.l_soff[k] - CK.l_soff[k - 1], Hk = CK.l_hoff[k] - CK.l_hoff[k - 1];
            int n0, n1, n2;
            bk_item_counts<CHECKER>(M_k, Hk, CK.ll_len, n0, n1, n2);
            const int Sg = 1 << ls;
            const uint32_t ch = (uint32_t)(((n0 + Sg - 1) >> ls) + ((n1 + Sg - 1) >> ls) + ((n2 + Sg - 1) >> ls));
            mx = ch > mx ? ch : mx;
        }
        chm[ls] = mx;
    }
    __syncthreads();

    int status = PDMPC_OK;
    bool dep_timeout = X.dep_timeout;
    uint32_t goal = 0;
    uint32_t Rn = 1;  // entries of the ready list (uniform: every thread carries it)
    uint32_t depth_seen = 0xffffffffu;  // deepest collision-free node at the last selection
    bool heavy = false;                 // the search has stalled once: its rounds grow
    uint32_t t_checks = 0, t_pairs = 0;  // this thread's share of the work counters
    // where the time goes (100 MHz ticks, PDMPC_DEBUG_TAIL=1): accumulated by thread 0 in LDS words, so that the bookkeeping costs
    // the round loop no registers
    lds_vu64* tk = (lds_vu64*)(gp_path + 200);  // [12]: mark, start, work, arrival, select (without the refills), wait, p1, p2, p3, phase B, refill, time of the early publication
    enum { TK_MARK, TK_START, tk_work, tk_arrival, tk_select, tk_wait, tk_p1, tk_p2, tk_p3, tk_pb, tk_refill, tk_pub };
    const bool ticking = A.debug_tail != 0 && tid == 0;
    if (ticking) {
        for (int i = 2; i < 12; ++i) tk[i] = 0ull;
        tk[TK_MARK] = tk[TK_START] = __builtin_amdgcn_s_memrealtime();
    }
    lds_u64s* chg = (lds_u64s*)((lds_u32*)((lds_d2*)(X.lsm + PDMPC_LK_PSHAPE) + Hp * PDMPC_VMAX) + PDMPC_HP_MAX);  // [HP_MAX] areas that differ from the expected ones (bk_incorporate_body)
    if (tid < PDMPC_HP_MAX) chg[tid] = 0ull;  // (read behind the barriers of the first round)
    lds_vu64* tk2 = (lds_vu64*)(gp_path + 242);  // [7] (diagnostics) the arrival handling in detail: poll + copy, re-check, parked nodes, bookkeeping + candidates, record + flag of a finished search
    if (ticking)
        for (int i = 0; i < 7; ++i) tk2[i] = 0ull;
#define BK_TICK2(i)                                                        \
    if (ticking && A.debug_tail != 3) {                                                         \
        const unsigned long long now__ = __builtin_amdgcn_s_memrealtime(); \
        tk2[i] += now__ - tk2[6];                                          \
        tk2[6] = now__;                                                    \
    }
#define BK_MARK2 \
    if (ticking && A.debug_tail != 3) tk2[6] = __builtin_amdgcn_s_memrealtime();
    // (debug_tail=3: the same six counters take a round's passes apart instead: the share decision, the check items, the sincos items, P1's
    // last barrier, the boundary up to the selection, the selection)
    const bool ticking3 = ticking && A.debug_tail == 3;
#define BK_TICK3(i)                                                        \
    if (ticking3) {                                                        \
        const unsigned long long now__ = __builtin_amdgcn_s_memrealtime(); \
        tk2[i] += now__ - tk2[6];                                          \
        tk2[6] = now__;                                                    \
    }
#define BK_MARK3 \
    if (ticking3) tk2[6] = __builtin_amdgcn_s_memrealtime();
#define BK_TICK(acc)                                                       \
    if (ticking) {                                                         \
        const unsigned long long now__ = __builtin_amdgcn_s_memrealtime(); \
        tk[acc] += now__ - tk[TK_MARK];                                    \
        tk[TK_MARK] = now__;                                               \
    }
    double far_mn = inf, far_mx = 0.0, near_mn = inf, near_mx = 0.0, mid_mn = inf;
    double* const mid_key = A.arena.mid_key + voff;
    uint32_t* const mid_id = A.arena.mid_id + voff;
    // appends (k, i) of the lanes with `take` to the open entries outside LDS (whole wave calls, straight-line): to mid up to the key
    // l_mid, to far above it.  A light search has no mid list (l_mid = -1): everything goes to far.
    double l_mid = -1.0;  // the value of the shared word BK_L_MID (uniform; read again wherever it changes: the refill, a reopened open set)
    auto load_l_mid = [&]() {
        const unsigned long long v = (unsigned long long)__double_as_longlong(sh_ld_d(sh, BK_L_MID));
        const uint32_t lo32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
        l_mid = __longlong_as_double((long long)(((unsigned long long)hi32 << 32) | lo32));
    };
    auto to_far = [&](bool take, double k, uint32_t i) {
        const unsigned long long b = __ballot(take);
        if (b && l_mid < 0.0) {  // (uniform) no mid list: the light searches' path
            const uint32_t base = sh_add_uniform(sh, FR_FAR_N, (uint32_t)__builtin_popcountll(b), lane);
            if (take) {
                const uint32_t pos = base + lane_rank(b, lane);
                F.far_key[pos] = k;
                F.far_id[pos] = i;
                far_mn = k < far_mn ? k : far_mn;
                far_mx = k > far_mx ? k : far_mx;
            }
        } else if (b) {
            const bool tm = k <= l_mid;
            const unsigned long long bm = __ballot(take && tm), bf = b & ~bm;
            // one LDS atomic reserves room in both lists: lane 0 adds far's count, lane 1 mid's (the other lanes add to scratch words, see sh_add_uniform)
            lds_u32* p = lane == 0 ? (lds_u32*)(sh + FR_FAR_N) : (lane == 1 ? (lds_u32*)(sh + BK_MID_N) : (lds_u32*)(sh + FR_SCRATCH + lane));
            const uint32_t old = __hip_atomic_fetch_add(p, lane == 0 ? (uint32_t)__builtin_popcountll(bf) : (lane == 1 ? (uint32_t)__builtin_popcountll(bm) : 0u), __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t base_f = (uint32_t)__builtin_amdgcn_readlane((int)old, 0), base_m = (uint32_t)__builtin_amdgcn_readlane((int)old, 1);
            if (take) {
                const uint32_t pos = tm ? base_m + lane_rank(bm, lane) : base_f + lane_rank(bf, lane);
                (tm ? mid_key : F.far_key)[pos] = k;
                (tm ? mid_id : F.far_id)[pos] = i;
                mid_mn = (tm && k < mid_mn) ? k : mid_mn;
                far_mn = (!tm && k < far_mn) ? k : far_mn;
                far_mx = (!tm && k > far_mx) ? k : far_mx;
            }
        }
    };
    auto flush_far = [&]() {
        sh_minmax_wave(sh, FR_FAR_MIN, FR_FAR_MAX, far_mn, far_mx, lane);
        far_mn = inf;
        far_mx = 0.0;
        if (__ballot(mid_mn < inf)) {  // (uniform over the wave)
            double m = mid_mn;
#pragma unroll
            for (int o = PDMPC_WAVE / 2; o > 0; o >>= 1) {
                const double a = __shfl_xor(m, o);
                m = a < m ? a : m;
            }
            if (lane == 0) sh_min_d(sh, BK_MID_MIN, m);
            mid_mn = inf;
        }
    };
    // appends (k, i) of the lanes with `take` to near (LDS; the caller has made sure there is room)
    auto to_near = [&](bool take, double k, uint32_t i) {
        const unsigned long long b = __ballot(take);
        if (b) {
            const uint32_t base = sh_add_uniform(sh, FR_NEAR_N, (uint32_t)__builtin_popcountll(b), lane);
            if (take) {
                const uint32_t pos = base + lane_rank(b, lane);
                near_key[pos] = k;
                near_id[pos] = i;
                near_mn = k < near_mn ? k : near_mn;
                near_mx = k > near_mx ? k : near_mx;
            }
        }
    };
    auto flush_near = [&]() {
        sh_minmax_wave(sh, FR_NEAR_MIN, FR_NEAR_MAX, near_mn, near_mx, lane);
        near_mn = inf;
        near_mx = 0.0;
    };
    // A search that can do nothing but wait for a predecessor (finished, or stalled on parked nodes): the first wave polls the
    // pending predecessors' done flags in a tight loop — an arrival is on every successor's critical path — while the others wait at
    // the barrier.  Returns (to every thread) whether the watchdog's limit of polls has been reached.
    auto bk_wait = [&]() -> bool {
        if (wave == 0) {
            uint32_t spins = 0;
            while (!bk_poll_predecessors(A, P, sh, 0ull, lane) && ++spins < 4096u) __builtin_amdgcn_s_sleep(1);
            if (lane == 0) sh[BK_IDLE] = sh[BK_IDLE] + spins + 1u;
        }
        __syncthreads();
        return sh[BK_IDLE] > A.spin_limit;
    };
    PhaseB R;
    R.n_popped = 0;
    R.n_expanded = 0;
    bool pb_valid = false;
    bool rec_valid = false, rec_written = false;  // the result record in HBM is the one this search would publish now / some record has been written
    bool vs_copied = false;                       // the LDS validity bytes have been copied to HBM since the tree last changed
    bool tie_replayed = false;                    // the search ended on bk_replay (its pop sequence is in the arena: pdmpc_debug_pop_trace)
    bool unpark_req = false;                      // (bk_flags bit 1) parked nodes come back at the next boundary; the re-check of the collision-free nodes waits
    bool verify_req = false;                      // the next round boundary verifies the tree against the areas that were copied since the last verification (BK_FD)
    // shared rounds (helper workgroups)
    unsigned long long* board = A.help_board + (size_t)slot * PDMPC_HB_WORDS;
    uint32_t help_seq = 0;  // rounds shared so far (same value in every thread)
    bool wanted = false;    // the board says that this search shares its rounds (helpers take seats)

    const int tid_k = tid, lane_k = lane;
    for (;;) {
        // (the thread's index is made opaque once per round: what the compiler derives from it — a few dozen per-thread addresses into
        // the LDS lists — would otherwise be kept in registers across the whole loop, and the kernel runs at the register cap)
        int tid_o = tid_k, lane_o = lane_k;
        asm volatile("" : "+v"(tid_o), "+v"(lane_o));
        int tid = tid_o, lane = lane_o;
#define BK_OPAQUE_TID                                      \
    {                                                      \
        int t__ = tid_k, l__ = lane_k;                     \
        asm volatile("" : "+v"(t__), "+v"(l__));           \
        tid = t__;                                         \
        lane = l__;                                        \
    }
        // ================= a round =================
        const bool ran_round = Rn != 0u;
        if (Rn) {
            pb_valid = false;  // (the tree grows: phase B's result is stale)
            rec_valid = false;
            vs_copied = false;
            // ---- P1: check items + sincos items.  A large round is shared with the helper workgroups that have taken a seat at this
            // search (CUs the launch leaves idle, bulk_helper_body): the owner posts one 48-byte record per entry (what a check reads of
            // the tree), keeps the first part of the list and hands every seated helper a contiguous range through that helper's own
            // assignment word; a helper mirrors the search's soup in its LDS, checks its range and leaves one verdict word per entry and
            // the round's number in its own done word.  Nobody claims anything: no compare-and-swap, and every word that is polled has
            // exactly one poller (rounds 3-4 let all helpers compete for tiles on one ticket word per search: a tile cost 15 us of which
            // the checks were 5, and more helpers made every search of the launch slower, profiles/r05_helper_sweep.txt).
            BK_MARK3
            const BkTreeSrc tsrc{&S, ready};
            const unsigned long long pend_now = A.bk_tentative ? sh_load64(sh, SH_PEND_LO) : 0ull;  // (their slots hold expected areas)
            bool share = A.n_helpers > 0 && Rn >= (uint32_t)A.bk_share_min && P.n_pred <= 64;
            uint32_t own_n = Rn, seats = 0, per_h = 0;
            if (share) {  // (uniform)
                if (tid == 0) {
                    __hip_atomic_store(board + PDMPC_HB_WEIGHT, (unsigned long long)(sh[FR_PROCESSED] + Rn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!wanted) __hip_atomic_store(board + PDMPC_HB_WANT, (unsigned long long)A.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // helpers may take seats from now on
                    const unsigned long long sw = __hip_atomic_load(board + PDMPC_HB_SEATS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    uint32_t k = (uint32_t)(sw >> 32) == A.launch_id ? (uint32_t)(sw & 0xffffffffull) : 0u;
                    sh[FR_HELP_CLOSED] = k < (uint32_t)PDMPC_HB_SEATS_MAX ? k : (uint32_t)PDMPC_HB_SEATS_MAX;
                }
                wanted = true;
                __syncthreads();
                seats = sh[FR_HELP_CLOSED];
                share = seats != 0u;
            }
            if (share) {  // (uniform)
                // the owner's part: a share as large as a helper's, less what posting and waiting cost it (1 / own_div of the round at most)
                per_h = (Rn + seats) / (seats + 1u);
                {
                    const uint32_t cap = (uint32_t)A.bk_tile;  // (a helper stages its range's records in LDS)
                    per_h = per_h < cap ? per_h : cap;
                }
                const uint32_t helped = per_h * seats < Rn ? per_h * seats : Rn;
                own_n = Rn - helped;
                ++help_seq;
                d2* post = (d2*)A.bk_post + (size_t)slot * RC * 3u;
                for (uint32_t r = own_n + (uint32_t)tid; r < Rn; r += (uint32_t)bd) {
                    uint32_t parent, packed;
                    const d2 p3 = node_piece(S, ready[r] - 1u, 3);
                    piece_link(p3, parent, packed);
                    bk_post_store(post + 3 * r, node_piece(S, parent - 1u, 0));
                    bk_post_store(post + 3 * r + 1, node_piece(S, parent - 1u, 2));
                    bk_post_store(post + 3 * r + 2, d2{p3.y, 0.0});
                }
                // every wave's stores must have reached L2 before thread 0 writes L2 back: a workgroup barrier alone does not wait for them
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (wave == 0) {
                    if (lane == 0) {
                        const unsigned long long all = P.n_pred >= 64 ? ~0ull : ((1ull << P.n_pred) - 1ull);
                        __hip_atomic_store(board + PDMPC_HB_N, (unsigned long long)Rn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(board + PDMPC_HB_MASK, all & ~sh_load64(sh, SH_PEND_LO), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if PDMPC_BK_POST_FENCES
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    wave_sync();
                    if ((uint32_t)lane < seats) {  // seat k checks entries own_n + k * per_h .. (its assignment word: round << 40 | first << 20 | count)
                        const uint32_t first = own_n + (uint32_t)lane * per_h;
                        const uint32_t cnt = first >= Rn ? 0u : (Rn - first < per_h ? Rn - first : per_h);
                        __hip_atomic_store(board + PDMPC_HB_ASSIGN + lane, ((unsigned long long)help_seq << 40) | ((unsigned long long)first << 20) | (unsigned long long)cnt, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            {
                BK_TICK3(0)
                const uint32_t Rr = own_n;
                const int ls = bk_chunk_shift(chm, Rr ? Rr : 1u, (uint32_t)bd);
                bk_check_items<CHECKER>(CK, tsrc, r_flag, 0u, Rr, ls, chm[ls], pend_now, tid, bd);
                BK_TICK3(1)
                for (uint32_t r = (uint32_t)(bd - 1 - tid); r < Rn; r += (uint32_t)bd) {  // (from the last thread down: the first waves carry the first chunks)
                    const uint32_t i0 = ready[r] - 1u;
                    uint32_t parent, packed;
                    piece_link(node_piece(S, i0, 3), parent, packed);
                    if (NODE_K(packed) < Hp) {
                        const d2 p1 = node_piece(S, i0, 1);
                        double sn, cs;
                        pdmpc_sincos(p1.x, &sn, &cs);  // expand_node.m:50-51
                        node_store_cs(S, i0, cs, sn);
                    }
                }
                BK_TICK3(2)
            }
            if (share) {
                if (wave == 0) {  // wait for the seated helpers: every lane on its helper's done word
                    uint32_t spins = 0;
                    bool bad = false;
                    for (;;) {
                        bool pending_h = false;
                        if ((uint32_t)lane < seats) pending_h = (uint32_t)__hip_atomic_load(board + PDMPC_HB_DONE + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != help_seq;
                        if (!__ballot(pending_h)) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > A.spin_limit) {
                            bad = true;
                            break;
                        }
                    }
                    if (lane == 0) {
                        if (bad) atomicOr((uint32_t*)&sh[FR_FLAGS], FRF_BUG);  // reported as an error status: must never happen
#if PDMPC_BK_POST_FENCES
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
                        atomicAdd(A.work_count + 4, 1ull);
                        atomicAdd(A.work_count + 5, (unsigned long long)(Rn - own_n));
                    }
                }
                __syncthreads();
                const uint32_t* hverdict = A.help_verdict + (size_t)slot * PDMPC_HELP_CAP;
                for (uint32_t r = own_n + (uint32_t)tid; r < Rn; r += (uint32_t)bd) {  // 1 collision-free, 2 collides, 3 crosses expected areas only
                    const uint32_t v = __hip_atomic_load(hverdict + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v < 1u || v > 3u) atomicOr((uint32_t*)&sh[FR_FLAGS], FRF_BUG);  // (assigned, reported finished, and no verdict)
                    r_flag[r] = v == 2u ? 1u : (v == 3u ? 2u : 0u);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (ticking && A.debug_tail == 2) {  // (diagnostics: rounds, P1 time, entries and the owner's part by the seats the round had: none, 1-7, 8-31, 32-64)
                const int bkt = !share ? 0 : (seats < 8u ? 1 : (seats < 32u ? 2 : 3));
                double* r0 = X.O->path_nodes[PDMPC_HP_MAX - 5];
                double* r1 = X.O->path_nodes[PDMPC_HP_MAX - 4];
                r0[bkt] += 1.0;
                r0[4 + bkt] += (double)(__builtin_amdgcn_s_memrealtime() - tk[TK_MARK]);
                r1[bkt] += (double)Rn;
                r1[4 + bkt] += (double)own_n;
            }
            BK_TICK3(3)
            BK_TICK(tk_p1)

            BK_OPAQUE_TID
            // ---- P2: verdicts, goal candidates, children counts; node indices by a scan (BK_P2 ready entries per thread).
            // The children of a node are expanded in groups of four lanes.
            bool ex[BK_P2];
            uint32_t cnt[BK_P2], rr[BK_P2];
            unsigned long long mine = 0;
#pragma unroll
            for (int s = 0; s < BK_P2; ++s) {
                const uint32_t r = (uint32_t)tid + (uint32_t)s * (uint32_t)bd;
                const bool in = r < Rn;
                rr[s] = r;
                ex[s] = false;
                cnt[s] = 0;
                if (in) {
                    const uint32_t id = ready[r], i0 = id - 1u;
                    const uint32_t fl = r_flag[r];
                    const bool valid = (fl & 3u) == 0u, parked = (fl & 3u) == 2u;  // (crosses nothing / only expected areas)
                    uint32_t parent, packed;
                    piece_link(node_piece(S, i0, 3), parent, packed);
                    const int k = NODE_K(packed);
                    vs_store(VS, i0, valid ? VS_VALID : (parked ? VS_TENT : VS_INVALID));
                    if (parked) {  // (rare: one LDS atomic each)
                        sh_add(sh, BK_NTENT, 1u);
                        sh_min_d(sh, BK_TENT_MIN, F.gkey[i0]);
                    }
                    if (parent) {  // the pairs the reference's InterX forms for this edge (InterX.m:63-76): (V - 1) x (M - 1) per soup
                        const int M_k = CK.l_soff[k] - CK.l_soff[k - 1], Hk = CK.l_hoff[k] - CK.l_hoff[k - 1];
                        t_checks += 1;
                        t_pairs += (uint32_t)(NODE_COLS(packed) - 1) * (uint32_t)((M_k > 1 ? M_k - 1 : 0) + (Hk > 1 ? Hk - 1 : 0) + (CK.ll_len > 1 ? CK.ll_len - 1 : 0));
                    }
                    if (valid && k == Hp) {
                        // a goal candidate if its ancestors are all collision-free still: the largest key of its path goes into its
                        // record (the cos / sin slot, which a node at the horizon never needs), the best one is chosen at the boundary
                        double b1;
                        const bool alive = fr_goal_path(S, VS, F.gkey, id, b1);
                        if (alive) {
                            node_store_cs(S, i0, b1, 0.0);
                            const uint32_t pos = sh_add(sh, FR_GOAL_N, 1u);
                            if (pos < 1024u)
                                F.goal_list[pos] = id;
                            else
                                atomicOr((uint32_t*)&sh[FR_FLAGS], FRF_GOALS_LOST);  // (more than a thousand candidates in one round: looked up again in the tree below)
                        }
                    } else if (valid) {
                        if ((uint32_t)k > sh[BK_DEPTH]) __hip_atomic_fetch_max((lds_u32*)&sh[BK_DEPTH], (uint32_t)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (rare: one node per level)
                        const lds_mask64* mrow = EE.l_mask + ((size_t)k * n + (NODE_TRIM(packed) - 1)) * nw;
                        uint32_t c = 0;
                        for (int w = 0; w < nw; ++w) c += (uint32_t)__builtin_popcountll(mrow[w]);
                        ex[s] = c != 0u;
                        cnt[s] = c;
                    }
                }
                mine += (unsigned long long)((cnt[s] + 3u) >> 2) | ((unsigned long long)cnt[s] << 32);
            }
            unsigned long long tot = 0;
            const unsigned long long base = wg_scan_excl(mine, wsum64, lane, wave, n_waves, tot);
            const uint32_t NG = (uint32_t)(tot & 0xffffffffull), NC = (uint32_t)(tot >> 32);
            const uint32_t nn_base = sh[FR_NNODES];
            const bool overflow = nn_base + NC > S.max_nodes;
            const bool all_far = sh[FR_NEAR_N] + NC > OC;  // near cannot take this round's children: they wait in far
            const double l_far = all_far ? -1.0 : sh_ld_d(sh, FR_L_FAR);
            BK_TICK(tk_p2)
            BK_OPAQUE_TID
            // ---- P3: expansion items (group of four successors of a collision-free node, lane): expand_node.m:18-90
            for (uint32_t tile0 = 0; tile0 < NG && !overflow; tile0 += VCAP) {  // (uniform; one tile unless a round has more than VCAP groups)
                if (tile0) __syncthreads();  // (the previous tile's items have read the lists)
                {
                    uint32_t g = (uint32_t)(base & 0xffffffffull), c = (uint32_t)(base >> 32);
#pragma unroll
                    for (int s = 0; s < BK_P2; ++s) {
                        const uint32_t ng = (cnt[s] + 3u) >> 2;
                        for (uint32_t q = 0; q < ng; ++q) {
                            const uint32_t gi = g + q - tile0;  // (unsigned: groups of earlier tiles are far outside)
                            if (gi < VCAP) {
                                vlist[gi] = rr[s] | (q << 16);
                                voffs[gi] = c;
                            }
                        }
                        g += ng;
                        c += cnt[s];
                    }
                }
                __syncthreads();
                const uint32_t ngt = NG - tile0 < VCAP ? NG - tile0 : VCAP, items = ngt * 4u;
                for (uint32_t b0 = 0; b0 < items; b0 += (uint32_t)bd) {  // (uniform trip count: wave-wide appends inside)
                    const uint32_t item = b0 + (uint32_t)tid;
                    const bool in = item < items;
                    const uint32_t gi = in ? item >> 2 : 0u, ent = vlist[gi];
                    const uint32_t r = ent & 0xffffu, rank = (ent >> 16) * 4u + (item & 3u);
                    const uint32_t id = ready[r], i0 = id - 1u;
                    NodeBits cu;
#pragma unroll
                    for (int q = 0; q < 4; ++q) cu.q[q] = node_piece(S, i0, q);
                    const NodeRec& cn = cu.r;
                    const int cTrim = NODE_TRIM(cn.packed), cK = NODE_K(cn.packed);
                    const int k_exp = cK + 1;            // expand_node.m:13
                    const int steps_to_go = Hp - k_exp;  // :37
                    const lds_mask64* mrow = EE.l_mask + ((size_t)cK * n + (cTrim - 1)) * nw;
                    // the rank-th successor (ascending trim: expand_node.m:18): its mask word and its place in that word
                    int w = 0;
                    uint32_t rk = rank;
                    uint64_t mask = mrow[0];
                    for (int q = 1; q < nw && rk >= (uint32_t)__builtin_popcountll(mask); ++q) {
                        rk -= (uint32_t)__builtin_popcountll(mask);
                        mask = mrow[q];
                        w = q;
                    }
                    const bool active = in && rk < (uint32_t)__builtin_popcountll(mask);
                    double f = 0.0;
                    uint32_t ci = 0;
                    if (active) {
                        const int t2 = w * 64 + nth_bit(mask, (int)rk);  // 0-based successor trim
                        const int m = (int)EE.l_mi[(cTrim - 1) * n + t2];
                        const double dx = EE.l_pose[m].dx, dy = EE.l_pose[m].dy, dyaw = EE.l_pose[m].dyaw;
                        const int ncols = EE.l_pose[m].n_cols;
                        NodeRec ch;
                        ch.x = cn.cs * dx - cn.sn * dy + cn.x;  // :53
                        ch.y = cn.sn * dx + cn.cs * dy + cn.y;  // :54
                        ch.yaw = cn.yaw + dyaw;                 // :55
                        ch.cs = 0.0;
                        ch.sn = 0.0;
                        ch.parent = id;
                        ch.packed = (uint32_t)(t2 + 1) | ((uint32_t)k_exp << 10) | ((uint32_t)m << 15) | ((uint32_t)ncols << 27);
                        // cost-to-come (:57-61) and cost-to-go (:66-73), summed in the reference's order
                        {
                            const double ddx = ch.x - EE.l_rx[k_exp - 1], ddy = ch.y - EE.l_ry[k_exp - 1];
                            const double nrm = sqrt(ddx * ddx + ddy * ddy);
                            ch.g = cn.g + nrm * nrm;  // :61
                        }
                        double expH = 0.0;
                        for (int it = 1; it <= steps_to_go; ++it) {
                            const double ddx = ch.x - EE.l_rx[k_exp - 1 + it], ddy = ch.y - EE.l_ry[k_exp - 1 + it];
                            const double nrm = sqrt(ddx * ddx + ddy * ddy);
                            const double df = nrm - EE.l_dcum[(k_exp - 1) * PDMPC_HP_MAX + (it - 1)];
                            const double m0 = (df > 0) ? df : 0.0;
                            expH = expH + m0 * m0;
                        }
                        ch.h = expH;
                        f = ch.g * 1 + expH * 1;  // GraphSearch.m:100-102
                        ci = nn_base + voffs[gi] + rank;  // 0-based index of the child (Tree.add_nodes, Tree.m:61)
                        node_store(S, ci, ch);
                        vs_store(VS, ci, 0);  // validity unknown
                        F.gkey[ci] = f;
                        F.glink[ci] = (unsigned long long)ch.parent | ((unsigned long long)ch.packed << 32);
                        ((ulonglong2*)A.arena.walk + voff)[ci].y = 0ull;  // (no stamp: the arena holds other searches' leftovers, see bk_classify_wave)
                        (A.arena.child0 + voff)[ci] = 0u;                 // (no children yet)
                        if (rank == 0u) (A.arena.child0 + voff)[i0] = ci + 1u;
                    }
                    to_near(active && !(f > l_far), f, ci + 1u);
                    to_far(active && f > l_far, f, ci + 1u);
                }
            }
            if (tid == 0) {  // (nobody reads these words before the barrier that ends the round)
                sh[FR_NNODES] = nn_base + (overflow ? 0u : NC);
                sh[FR_PROCESSED] = sh[FR_PROCESSED] + Rn;
                if (A.debug_tail == 1 && sh[FR_ROUNDS] < 32u) X.O->path_nodes[PDMPC_HP_MAX - 7 + (int)(sh[FR_ROUNDS] >> 3)][sh[FR_ROUNDS] & 7u] = (double)Rn;  // (diagnostics: the sizes of the first thirty-two rounds in rows HP_MAX - 7 .. HP_MAX - 4)
                sh[FR_ROUNDS] = sh[FR_ROUNDS] + 1u;
                if (overflow) sh[FR_FLAGS] = sh[FR_FLAGS] | FRF_OVERFLOW;
            }
            Rn = 0;
            BK_TICK(tk_p3)
        }
        BK_OPAQUE_TID
        // the key ranges the appends of this round and the selection before it have met
        flush_near();
        flush_far();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // ---- the barrier that ends the round
        if (sh[FR_GOAL_N] > 1024u) {  // (uniform; the list holds the first 1024)
            __syncthreads();
            if (tid == 0) sh[FR_GOAL_N] = 1024u;
            __syncthreads();
        }
        fr_resolve_goals(F, S, tid, lane, wave);
        if (sh[FR_FLAGS] & FRF_GOALS_LOST) {  // (uniform) candidates were lost: every collision-free node at the horizon is offered again
            __syncthreads();
            if (tid == 0) {
                sh[FR_BEST_ID] = 0;
                sh[FR_PATH_FOR] = 0;
                sh[FR_FLAGS] = sh[FR_FLAGS] & ~FRF_GOALS_LOST;
            }
            __syncthreads();
            uint32_t nn = sh[FR_NNODES];
            nn = nn < S.max_nodes ? nn : S.max_nodes;
            for (uint32_t base = 0; base < nn; base += (uint32_t)bd) {  // (uniform trip count: barriers inside)
                const uint32_t b = base + (uint32_t)wave * PDMPC_WAVE;
                const uint32_t i0 = b + (uint32_t)lane;
                const bool in = i0 < nn;
                const uint32_t j0 = in ? i0 : 0u;
                const bool cand = in && vs_load(VS, j0) == VS_VALID && NODE_K(((const uint32_t*)(S.gn + j0))[15]) == Hp;
                bk_offer_goals_lanes(F, S, VS, cand, j0 + 1u, lane);
                __syncthreads();  // at most blockDim candidates per pass: the list cannot overflow
                fr_resolve_goals(F, S, tid, lane, wave);
            }
        }
        BK_TICK(tk_work)

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
        // Predecessors that finished meanwhile (PrioritizedController.m:476-491).  An arrival is two things.  (1) The COPY: the
        // predecessor's areas replace its expected ones in the soup — cheap, done here at every round boundary —; from then on edges are
        // checked against the real areas (the predecessor leaves SH_PEND) and the predecessor is due for (2) the VERIFICATION (BK_FD):
        // every collision-free edge evaluated before the copy is re-checked against the new areas, parked nodes come back into the
        // open set.  A finished search that waits (bk_wait_done) copies and checks its plan's path only, and the verification follows
        // for everybody who has arrived meanwhile; a running search verifies right behind the copy (measured: putting it off until
        // the search has nothing else to do leaves parked nodes parked and dead subtrees alive — C2's heavy steps 1.35 -> 1.75 ms).
        BK_MARK2
        // (bk_flags bits 2-4, PDMPC_TUNING=poll_every=K: a search that has just run a round looks for arrivals at every K-th boundary
        // only — an arrival event costs a running light search 15-25 us whatever it brings; a search that stalls or is done polls at once)
        const uint32_t poll_k = ((uint32_t)A.bk_flags >> 2) & 7u;
        const bool skip_poll = ran_round && poll_k != 0u && (sh[FR_ROUNDS] % (poll_k + 1u)) != 0u && sh[SH_STATE] != ST_ARRIVED;
        if (!skip_poll && sh_load64(sh, SH_PEND_LO) != 0ull) {  // (uniform: written by thread 0 between barriers)
            if (wave == 0 && sh[SH_STATE] != ST_ARRIVED) (void)bk_poll_predecessors(A, P, sh, 0ull, lane);  // (a waiting search has polled already: bk_wait)
            __syncthreads();
            if (sh[SH_STATE] == ST_ARRIVED) {  // (uniform)
                const unsigned long long arr = sh_load64(sh, SH_ARR_LO);
                bk_incorporate(P.out, P.pred, P.l_soup, P.l_soff, P.l_lit, Hp, arr, tid, bd, chg);
                __syncthreads();
                if (tid == 0) {
                    const unsigned long long pend = sh_load64(sh, SH_PEND_LO) & ~arr, fd = sh_load64(sh, BK_FD_LO) | arr;
                    sh[SH_PEND_LO] = (uint32_t)pend;
                    sh[SH_PEND_HI] = (uint32_t)(pend >> 32);
                    sh[BK_FD_LO] = (uint32_t)fd;
                    sh[BK_FD_HI] = (uint32_t)(fd >> 32);
                    sh[SH_ARR_LO] = 0;
                    sh[SH_ARR_HI] = 0;
                    sh[SH_STATE] = ST_RUN;
                    if (ticking) {  // (diagnostics: the last arrival that met a running search, and the rounds done by then)
                        X.O->path_nodes[PDMPC_HP_MAX - 2][6] = (double)sh[FR_ROUNDS];
                        X.O->path_nodes[PDMPC_HP_MAX - 2][7] = (double)__builtin_amdgcn_s_memrealtime();
                    }
                }
                __syncthreads();
                // (putting the whole verification off until the search stalls or is done: C2 1 382 -> 1 212, C3 1 100 -> 939, C4 82.8 -> 71.4
                // steps/s — parked nodes stayed parked.  bk_flags bit 1, PDMPC_TUNING=lazy_verify=1: the parked nodes come back at once,
                // only the re-check of the collision-free nodes — which can take edges away, never add any — waits)
                if (A.bk_flags & 2)
                    unpark_req = true;
                else
                    verify_req = true;
            }
        }
        BK_TICK2(0)
        if (unpark_req && !verify_req) {  // (uniform) the arrival's cheap half: parked nodes are open again and meet the real areas in a round
            const uint32_t n_parked = sh[BK_NTENT];
            uint32_t nn = sh[FR_NNODES];
            nn = nn < S.max_nodes ? nn : S.max_nodes;
            if (n_parked) {  // (uniform)
                const bool fits = sh[FR_NEAR_N] + n_parked <= OC;
                const double l_far = sh_ld_d(sh, FR_L_FAR);
                for (uint32_t base = 0; base < nn; base += (uint32_t)bd) {  // (uniform trip count: wave-wide appends inside)
                    const uint32_t i0 = base + (uint32_t)tid;
                    const bool tent = i0 < nn && vs_load(VS, i0 < nn ? i0 : 0u) == VS_TENT;
                    const double k = tent ? F.gkey[i0] : 0.0;
                    if (tent) vs_store(VS, i0, VS_UNKNOWN);
                    to_near(tent && fits && !(k > l_far), k, i0 + 1u);
                    to_far(tent && !(fits && !(k > l_far)), k, i0 + 1u);
                }
                flush_near();
                flush_far();
                vs_copied = false;
                __syncthreads();
                if (tid == 0) {
                    sh[BK_NTENT] = 0;
                    sh_st_d(sh, BK_TENT_MIN, inf);
                }
                __syncthreads();
            }
        }
        unpark_req = false;
        if (verify_req) {  // (uniform)
            verify_req = false;
            const unsigned long long arr = sh_load64(sh, BK_FD_LO);  // everybody whose areas were copied since the last verification
            uint32_t nn = sh[FR_NNODES];
            nn = nn < S.max_nodes ? nn : S.max_nodes;
            // Only collision-free nodes can lose their edge, and only to an area that differs from the one their edge was checked
            // against (chg: for a predecessor that carries on with its last plan that is the horizon's step alone).  They are gathered
            // first, a list's worth of the tree at a time, so that the items are dense: an item is a chain of two dependent reads of
            // the tree and seven segment tests, and lanes that skip theirs cost what the busy lanes of their wave cost (C2's last
            // vehicles, 15 predecessors each: 25-30 us per arrival event with every node x predecessor an item).
            uint32_t maxc = 0;
            for (int k = 0; k < Hp; ++k) {  // (uniform)
                const uint32_t c = (uint32_t)__builtin_popcountll(chg[k] & arr);
                maxc = c > maxc ? c : maxc;
            }
            if (ticking) {
                unsigned long long due = 0;
                for (int k = 0; k < Hp; ++k) due += (unsigned long long)__builtin_popcountll(chg[k] & arr);
                atomicAdd(A.work_count + 14, due);
                atomicAdd(A.work_count + 15, (unsigned long long)(Hp * __builtin_popcountll(arr)));
            }
#pragma unroll 1
            for (uint32_t base0 = 0; base0 < nn && maxc != 0u;) {  // (uniform trip counts: barriers inside)
                if (tid == 0) sh[FR_VLIST_N] = 0;
                __syncthreads();
                uint32_t b = base0;
                for (; b < nn && b - base0 + (uint32_t)bd <= (uint32_t)FR_NBINS; b += (uint32_t)bd) {  // (the list holds FR_NBINS nodes)
                    const uint32_t i0 = b + (uint32_t)tid;
                    const uint32_t j0 = i0 < nn ? i0 : 0u;
                    const int k = NODE_K((uint32_t)(F.glink[j0] >> 32));
                    const bool v = i0 < nn && k > 0 && vs_load(VS, j0) == VS_VALID && (chg[k > 0 ? k - 1 : 0] & arr) != 0ull;
                    const unsigned long long bal = __ballot(v);
                    if (bal) {
                        const uint32_t pos0 = sh_add_uniform(sh, FR_VLIST_N, (uint32_t)__builtin_popcountll(bal), lane);
                        if (v) hist[pos0 + lane_rank(bal, lane)] = i0;
                    }
                }
                __syncthreads();
                base0 = b;
                BK_TICK2(5)
                const uint32_t cnt = sh[FR_VLIST_N];
                bk_recheck_items<CHECKER>(S, VS, CK, P, hist, cnt, arr, chg, maxc, sh, tid, bd);
                __syncthreads();
            }
            BK_TICK2(1)
            flags = sh[FR_FLAGS];
            // Nodes lost their edges.  The best candidate survives unless one of its own path did (a published plan's path always
            // does: it was checked before the areas went out): then it stays the best — what comes after it still does, nothing that
            // was dropped comes back, no candidate has to be looked at again — and only the counts are due again (phase B).
            bool best_alive = false;
            if ((flags & FRF_INVALIDATED) && sh[FR_BEST_ID] != 0u) {  // (uniform; every thread walks the same Hp nodes)
                const uint32_t g = sh[FR_BEST_ID];
                double b1;
                best_alive = vs_load(VS, g - 1u) == VS_VALID && fr_goal_path(S, VS, F.gkey, g, b1);
            }
            const bool lost_best = (flags & FRF_INVALIDATED) && !best_alive;
            const bool reopen = lost_best && sh[FR_DROPPED] != 0u;
            // parked nodes (their edges crossed expected areas only) come back into the open set: never evaluated, as far as anybody
            // can tell — a round will check them against what the soup holds then.  (reopen: the rebuild below finds them in the tree)
            const uint32_t n_parked = sh[BK_NTENT];
            if (n_parked) {  // (uniform)
                const bool fits = sh[FR_NEAR_N] + n_parked <= OC;
                const double l_far = sh_ld_d(sh, FR_L_FAR);
                for (uint32_t base = 0; base < nn; base += (uint32_t)bd) {  // (uniform trip count: wave-wide appends inside)
                    const uint32_t i0 = base + (uint32_t)tid;
                    const bool tent = i0 < nn && vs_load(VS, i0 < nn ? i0 : 0u) == VS_TENT;
                    const double k = tent ? F.gkey[i0] : 0.0;
                    if (tent) vs_store(VS, i0, VS_UNKNOWN);
                    const bool push = tent && !reopen;
                    to_near(push && fits && !(k > l_far), k, i0 + 1u);
                    to_far(push && !(fits && !(k > l_far)), k, i0 + 1u);
                }
            }
            BK_TICK2(2)
            if (flags & FRF_INVALIDATED) {
                pb_valid = false;
                rec_valid = false;
            }
            vs_copied = false;  // (verdicts may have changed, parked nodes have come back)
            __syncthreads();
            if (tid == 0) {
                sh[BK_ARRIVALS] = sh[BK_ARRIVALS] + 1u;  // (reported at the end: a global atomic here sits on every level's hand-over)
                sh[BK_FD_LO] = 0;
                sh[BK_FD_HI] = 0;
                for (int k = 0; k < Hp; ++k) chg[k] = 0ull;
                sh[BK_NTENT] = 0;
                sh_st_d(sh, BK_TENT_MIN, inf);
                if (flags & FRF_INVALIDATED) {
                    sh[FR_EVER_INVAL] = 1;
                    if (lost_best) {  // look at all the candidates again
                        sh[FR_BEST_ID] = 0;
                        sh[FR_PATH_FOR] = 0;
                    }
                    sh[FR_FLAGS] = flags & ~FRF_INVALIDATED;
                }
                if (reopen) {
                    // open entries were dropped because they come after a candidate that may be gone now: rebuild the open set
                    // from the tree (every generated node that was never evaluated is open)
                    sh[FR_NEAR_N] = 0;
                    sh[FR_FAR_N] = 0;
                    sh[FR_DROPPED] = 0;
                    sh_st_d(sh, FR_NEAR_MIN, inf);
                    sh_st_d(sh, FR_NEAR_MAX, 0.0);
                    sh_st_d(sh, FR_FAR_MIN, inf);
                    sh_st_d(sh, FR_FAR_MAX, 0.0);
                    sh_st_d(sh, FR_L_FAR, -1.0);  // (everything goes to far until the next refill)
                    sh[BK_MID_N] = 0;
                    sh_st_d(sh, BK_MID_MIN, inf);
                    sh_st_d(sh, BK_L_MID, -1.0);
                }
            }
            __syncthreads();
            if (reopen) l_mid = -1.0;  // (uniform)
            if (lost_best) {
                for (uint32_t base = 0; base < nn; base += (uint32_t)bd) {  // (uniform trip count: barriers inside)
                    const uint32_t b = base + (uint32_t)wave * PDMPC_WAVE;
                    const uint32_t i0 = b + (uint32_t)lane;
                    const bool in = i0 < nn;
                    const uint32_t j0 = in ? i0 : 0u;  // (straight-line code: every lane loads something valid)
                    const uint32_t vst = vs_load(VS, j0);
                    const uint32_t par = node_parent(S, j0);
                    const bool cand = in && vst == VS_VALID && NODE_K(((const uint32_t*)(S.gn + j0))[15]) == Hp;
                    const bool open = reopen && in && vst == VS_UNKNOWN && par != 0u && vs_load(VS, par ? par - 1u : 0u) == VS_VALID;
                    if (reopen) to_far(open, F.gkey[j0], j0 + 1u);
                    bk_offer_goals_lanes(F, S, VS, cand, j0 + 1u, lane);
                    __syncthreads();  // at most blockDim candidates per pass: the list cannot overflow
                    fr_resolve_goals(F, S, tid, lane, wave);
                }
                flush_far();
                __syncthreads();
            }
            if (n_parked) {  // (uniform) the key ranges of what came back
                flush_near();
                flush_far();
                __syncthreads();
            }
            flags = sh[FR_FLAGS];
            BK_TICK2(3)
        }
        if (flags & FRF_TIE) {  // (uniform) equal keys where the order decides: from here on the search is headed for the replay
            __syncthreads();
            if (tid == 0) {
                sh[BK_TIEMODE] = 1u;
                sh[FR_FLAGS] = sh[FR_FLAGS] & ~FRF_TIE;
            }
            __syncthreads();
        }
        const bool tie_mode = sh[BK_TIEMODE] != 0u;
        BK_TICK(tk_arrival)
        BK_MARK3

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

        // are we done?  Open entries above the candidate's path maximum come after it; the others are looked at one by one when a
        // round selects them.  An empty open set without a candidate is exhaustion (GraphSearch.m:57-61).
        // (far_n / far_min: the open entries outside LDS, mid and far together)
        const uint32_t near_n = sh[FR_NEAR_N], mid_n = sh[BK_MID_N], far_n = sh[FR_FAR_N] + mid_n;
        const double near_min = near_n ? sh_ld_d(sh, FR_NEAR_MIN) : inf;
        double far_min = far_n != mid_n ? sh_ld_d(sh, FR_FAR_MIN) : inf;
        {
            const double mm = mid_n ? sh_ld_d(sh, BK_MID_MIN) : inf;
            far_min = mm < far_min ? mm : far_min;
        }
        const double open_min = near_min < far_min ? near_min : far_min;
        const double bb = best ? sh_ld_d(sh, FR_BEST_B1) : inf;
        // parked nodes (their edges cross only areas a pending predecessor is expected to take) count as open
        const uint32_t n_tent = sh[BK_NTENT];
        const double tent_min = n_tent ? sh_ld_d(sh, BK_TENT_MIN) : inf;
        bool done = false, stalled = false;
        if (best) {
            // (bb == open_min: a tie between an open node and a node of the best path — not finished: the round takes the entry, its
            // classification marks the search)
            done = bb < open_min && bb < tent_min;
            stalled = bb < open_min && !done;  // nothing open comes before the candidate, but a parked node may: only an arrival tells
        } else {
            done = near_n == 0u && far_n == 0u && n_tent == 0u;
            stalled = near_n == 0u && far_n == 0u && !done;
        }
        const bool unverified = sh_load64(sh, BK_FD_LO) != 0ull;  // (uniform) areas have been copied since the last verification
        if (stalled && unverified) {  // the parked nodes are judged against what has arrived
            verify_req = true;
            continue;
        }
        if (stalled) {  // (uniform) wait for a predecessor
            if (bk_wait()) {
                dep_timeout = true;
                status = PDMPC_EXHAUSTED;
                break;
            }
            BK_TICK(tk_wait)
            continue;
        }
        if (done && tie_mode && !dep_timeout) {
            // A search that has met equal keys ends on the replay: its tree — every node the reference's heap can pop has been
            // evaluated by now — is popped once more in the order of the libstdc++ heap (bk_replay), which yields the goal, the counts
            // and the ids.  The replay wants final verdicts: every predecessor arrived and verified first.
            if (unverified) {
                verify_req = true;
                continue;
            }
            if (sh_load64(sh, SH_PEND_LO) != 0ull) {
                if (bk_wait()) {
                    dep_timeout = true;
                    status = PDMPC_EXHAUSTED;
                    break;
                }
                BK_TICK(tk_wait)
                continue;
            }
            __syncthreads();
            if (tid == 0) {  // nothing that is still open can be popped before the goal (finished): the open set's room serves the heap
                sh[FR_NEAR_N] = 0;
                sh[FR_FAR_N] = 0;
                sh[BK_MID_N] = 0;
                sh_st_d(sh, FR_NEAR_MIN, inf);
                sh_st_d(sh, FR_NEAR_MAX, 0.0);
                sh_st_d(sh, FR_FAR_MIN, inf);
                sh_st_d(sh, FR_FAR_MAX, 0.0);
                sh_st_d(sh, FR_L_FAR, inf);
                sh_st_d(sh, BK_MID_MIN, inf);
                sh_st_d(sh, BK_L_MID, -1.0);
            }
            l_mid = -1.0;
            __syncthreads();
            if (wave == 0) bk_replay<NW>(A, X, F, EE, near_key, near_id, OC, mid_id, F.near_id, A.arena.child0 + voff, ref_ids, lane);
            __syncthreads();
            BK_TICK(tk_pb)
            if (sh[FR_FLAGS] & FRF_BUG) {
                dep_timeout = true;
                status = PDMPC_EXHAUSTED;
                break;
            }
            const uint32_t need = sh[BK_RP_NEED];
            if (need) {  // (uniform) a node the heap pops was dropped as coming after a candidate that tied with the goal: a round of its own
                if (tid == 0) {
                    ready[0] = need;
                    r_flag[0] = 0u;
                }
                Rn = 1;
                __syncthreads();
                continue;
            }
            goal = sh[BK_RP_GOAL];
            if (sh[BK_PUBLISHED] != 0u && goal != best) atomicOr((uint32_t*)&sh[FR_FLAGS], FRF_BUG);  // (the areas that went out are the candidate's: must never happen)
            R.n_popped = sh[BK_RP_NPOP];
            R.n_expanded = sh[BK_RP_NREF];
            pb_valid = true;
            rec_valid = false;
            tie_replayed = true;
            status = goal ? PDMPC_OK : PDMPC_EXHAUSTED;
            __syncthreads();
            if (sh[FR_FLAGS] & FRF_BUG) dep_timeout = true;
            break;
        }
        BK_MARK2
        const bool publish_first = done && !unverified && !pb_valid && sh[BK_PUBLISHED] == 0u && sh_load64(sh, SH_PEND_LO) == 0ull;
        if (done && (unverified || publish_first) && !dep_timeout) {
            // Finished as far as the verified areas go.  What the successors wait for comes first: the record with the plan's areas (its
            // counts are rewritten after the verification if that takes edges away), the areas that were copied since against the path,
            // the done flag if they pass and nobody is outstanding (bk_wait_done without waiting); then the verification.
            // (publish_first: every predecessor has arrived and has been verified — the areas go out before phase B counts the pops,
            // which only the host waits for: 17-35 us per hop of C2's level chain)
            if (!rec_valid && sh[BK_PUBLISHED] == 0u) {
                if (wave == 0) bk_write_record(A, X, best, best ? PDMPC_OK : PDMPC_EXHAUSTED, false, R.n_popped, R.n_expanded, false, nullptr, rec_written, false, lane);
                rec_written = true;
            }
            // (bk_wait_done rewrites the pending set, the published flag, ...: every wavefront has made this boundary's decisions on
            // them before the first one does — a wavefront can lag arbitrarily far behind the last barrier)
            __syncthreads();
            if (wave == 0 && sh[BK_PUBLISHED] == 0u)
                bk_wait_done(A.done_flag, P.pred, P.out, A.epoch, (uint32_t)slot, P.l_soup, P.l_soff, P.l_lit, (lds_d2*)(X.lsm + PDMPC_LK_PSHAPE), sh, Hp, P.n_pred, best != 0u,
                             ticking ? (lds_vu64*)(tk + tk_pub) : (lds_vu64*)nullptr, sh_load64(sh, BK_FD_LO), 0u, CHECKER == PDMPC_CHECK_SAT, lane);
            __syncthreads();
            BK_TICK2(4)
            if (unverified) {
                verify_req = true;
                BK_TICK(tk_arrival)
                continue;
            }
        }
        if (done) {
            // Phase B right away, also when predecessors are still planning: an arrival that invalidates nothing leaves the tree,
            // hence the counts and ids, as they are, and the result goes out as soon as the last predecessor has been looked at.
            if (!pb_valid && !dep_timeout) {
                __syncthreads();
                R = fr_phase_b<NW>(A, X, F, EE, best, ref_ids, (LDS_AS unsigned char*)(X.lsm + PDMPC_LK_CAND), F.near_key, F.near_id, gp_path);
                pb_valid = true;
                BK_TICK(tk_pb)
                const uint32_t pflags = sh[FR_FLAGS];
                __syncthreads();
                if (pflags & FRF_TIE) {  // (uniform) equal keys among the nodes in front of the goal: the counts and ids come from the replay
                    pb_valid = false;
                    continue;  // (the round boundary turns the flag into BK_TIEMODE)
                }
                if (pflags & FRF_BUG) dep_timeout = true;  // reported as an error status: must never happen
            }
            if (sh_load64(sh, SH_PEND_LO) == 0ull || dep_timeout) {
                goal = best;
                status = best ? PDMPC_OK : PDMPC_EXHAUSTED;
                break;
            }
            // finished, but predecessors that are still planning may yet invalidate what we found: the record is written meanwhile
            if (!rec_valid && !dep_timeout) {
                if (wave == 0) bk_write_record(A, X, best, best ? PDMPC_OK : PDMPC_EXHAUSTED, false, R.n_popped, R.n_expanded, best != 0u, ref_ids, rec_written, sh[BK_PUBLISHED] != 0u, lane);
                rec_valid = true;
                rec_written = true;
            }
            // (c) the validity bytes of the LDS-resident nodes go to HBM now as well (debug read-back of the tree): not behind the last arrival
            if (!vs_copied) {
                uint32_t nn1 = sh[FR_NNODES];
                nn1 = nn1 < S.max_nodes ? nn1 : S.max_nodes;
                const uint32_t nv = VS.NV < nn1 ? VS.NV : nn1;
                for (uint32_t i = (uint32_t)tid; i < nv; i += (uint32_t)bd) VS.g[i] = VS.l[i];
                vs_copied = true;
            }
            if (A.bk_fast_arrival && !dep_timeout) {
                __syncthreads();  // (as above: everybody has read the pending set this wait is about to rewrite)
                if (wave == 0)
                    bk_wait_done(A.done_flag, P.pred, P.out, A.epoch, (uint32_t)slot, P.l_soup, P.l_soff, P.l_lit, (lds_d2*)(X.lsm + PDMPC_LK_PSHAPE), sh, Hp, P.n_pred, best != 0u,
                                 ticking ? (lds_vu64*)(tk + tk_pub) : (lds_vu64*)nullptr, 0ull, 4096u, CHECKER == PDMPC_CHECK_SAT, lane);
                __syncthreads();
                if (ticking && sh[BK_WAITRES] == 1u) {  // (diagnostics: an arrival crossed the finished plan's path)
                    X.O->path_nodes[PDMPC_HP_MAX - 2][6] = (double)sh[FR_ROUNDS];
                    X.O->path_nodes[PDMPC_HP_MAX - 2][7] = (double)__builtin_amdgcn_s_memrealtime();
                }
                if (sh[BK_IDLE] > A.spin_limit) dep_timeout = true;
            } else if (bk_wait()) {
                dep_timeout = true;  // a predecessor never finished: give up on it (reported as an error status)
            }
            BK_TICK(tk_wait)
            continue;
        }

        BK_OPAQUE_TID
        // A round takes the smallest open keys: bk_round0 while the search is young (a round costs about the same for one node as for a
        // few dozen: the items of a small round run side by side), growing with the work done up to bk_round.
        // (rounds stay at bk_round0 while every round gets one level deeper — a light search is over after Hp + 1 of them and what a
        // round takes beyond what the reference pops is wasted; a search that stalls, or goes on beyond Hp + 2 rounds, is not light:
        // its rounds grow with the work done)
        uint32_t round_target;
        {
            const uint32_t done_so_far = sh[FR_PROCESSED];
            const uint32_t depth_now = sh[BK_DEPTH];
            heavy = heavy || depth_now == depth_seen || sh[FR_ROUNDS] > (uint32_t)Hp + 1u;
            depth_seen = depth_now;
            const uint32_t ramp = (uint32_t)A.bk_round0 + (heavy ? done_so_far / (uint32_t)A.bk_ramp : 0u);
            round_target = ramp < (uint32_t)A.bk_round ? ramp : (uint32_t)A.bk_round;
        }
        // ---- near holds fewer entries than the round wants (or nothing below far's smallest key): it is topped up from far with the
        // smallest entries far holds, as many as leave room for a round's children.  (Topping up when near is EMPTY only made a heavy
        // search alternate between rounds of a thousand nodes, the few hundred those left behind, and a handful of new children.)
        uint32_t nn_near = near_n;
        const bool merge = nn_near != 0u && far_min < near_min;
        if (far_n != 0u && (nn_near < round_target || merge)) {
            if (merge) {  // (rare: near goes into far first so that the refill sees every open entry)
                for (uint32_t b0 = 0; b0 < nn_near; b0 += (uint32_t)bd) {
                    const uint32_t e = b0 + (uint32_t)tid;
                    const bool in = e < nn_near;
                    to_far(in, in ? near_key[e] : 0.0, in ? near_id[e] : 0u);
                }
                flush_far();
                __syncthreads();
                if (tid == 0) sh[FR_NEAR_N] = 0;
                __syncthreads();
                nn_near = 0;
            }
            // A heavy search's far list holds a hundred thousand entries and more; a pass over all of it for every refill of near (every
            // other round) cost as much as the checks.  Beyond bk_mid_min entries far feeds near through the mid list: stage 0 moves a
            // band of far's smallest keys (about bk_mid_fill entries, up to the key l_mid) to mid, stage 1 refills near from mid; until
            // mid runs short again only mid is scanned, and open entries up to l_mid that leave LDS go to mid (to_far).
            const uint32_t room = OC - OC / 6u;  // (a sixth of near stays free for the children of the rounds to come)
            const uint32_t fill_near = room > nn_near + 64u ? room - nn_near : 64u;
            int stage = (sh[FR_FAR_N] > (uint32_t)A.bk_mid_min && sh[BK_MID_N] < fill_near) ? 0 : 1;
#pragma unroll 1
            for (; stage < 2; ++stage) {
                const bool to_mid = stage == 0;
                const bool from_mid = !to_mid && sh[BK_MID_N] != 0u;  // (uniform)
                if (!to_mid && !from_mid && sh[FR_FAR_N] == 0u) break;  // (nothing to refill from)
                double* const src_key = from_mid ? mid_key : F.far_key;
                uint32_t* const src_id = from_mid ? mid_id : F.far_id;
                const int SRC_N = from_mid ? BK_MID_N : FR_FAR_N;
                const uint32_t fn = sh[SRC_N];
                const uint32_t fill = to_mid ? (uint32_t)A.bk_mid_fill : fill_near;
                const double l_mid_old = sh_ld_d(sh, BK_L_MID);
                const uint32_t far_left = from_mid ? sh[FR_FAR_N] : 0u;  // (entries beyond the source)
                double lo = from_mid ? sh_ld_d(sh, BK_MID_MIN) : sh_ld_d(sh, FR_FAR_MIN), hi = from_mid ? l_mid_old : sh_ld_d(sh, FR_FAR_MAX);
                const double hi_src = hi;
                uint32_t bsel = FR_NBINS - 1;
                double scale = 0.0;
                // (a list of tens of thousands of entries is histogrammed on a sample: every stride-th entry.  The threshold only has to
                // bring about `fill` entries; what the chosen bins hold beyond near's room goes back, what they hold less comes next time)
                const uint32_t stride = fn > 16384u ? fn / 8192u : 1u;
                const uint32_t fill_s = stride > 1u ? (fill / stride > 16u ? fill / stride : 16u) : fill;
                for (int zoom = 0; zoom < 6; ++zoom) {
                    scale = hi > lo ? (double)FR_NBINS / (hi - lo) : 0.0;
                    for (int i = tid; i < FR_NBINS; i += bd) hist[i] = 0;
                    __syncthreads();
                    if (stride == 1u) {
                        fr_histogram(F, src_key, fn, lo, scale);
                    } else {
                        for (uint32_t e = (uint32_t)tid * stride; e < fn; e += (uint32_t)bd * stride)
                            __hip_atomic_fetch_add(&hist[fr_bin(src_key[e], lo, scale)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    __syncthreads();
                    if (wave == 0) bk_far_select(F, fill_s, lane);
                    __syncthreads();
                    bsel = sh[FR_SEL_BIN];
                    const uint32_t cum = sh[FR_SEL_CUM] * stride;
                    __syncthreads();
                    if (bsel != 0u || scale == 0.0 || (to_mid ? cum <= 4u * fill : nn_near + cum <= OC - 64u)) break;
                    hi = lo + (hi - lo) / (double)FR_NBINS;  // nearly everything sits in the first bin: look closer
                }
                const bool all = bsel >= FR_NBINS - 1 || scale == 0.0;  // every entry of the source is taken
                const double l_new = all ? inf : lo + (double)(bsel + 1u) / scale;
                if (tid == 0) {
                    if (from_mid) {
                        sh_st_d(sh, BK_MID_MIN, inf);
                    } else {
                        sh_st_d(sh, FR_FAR_MIN, inf);
                        sh_st_d(sh, FR_FAR_MAX, 0.0);
                    }
                    if (to_mid) {
                        sh_st_d(sh, BK_L_MID, all ? hi_src : l_new);  // (from now on what leaves LDS with a key up to here joins mid)
                    } else {
                        if (!from_mid) sh_st_d(sh, BK_L_MID, -1.0);  // (near is fed by far directly: no mid list)
                        if (nn_near == 0u) {  // (else near's key range stays and takes the new entries in)
                            sh_st_d(sh, FR_NEAR_MIN, inf);
                            sh_st_d(sh, FR_NEAR_MAX, 0.0);
                        }
                        // children up to this key join near from now on (what near still holds lies below the old limit, what the source
                        // held above it); with all of mid taken and far not empty, near's limit is mid's: far's keys lie above it
                        const double l_far_new = (all && far_left != 0u) ? l_mid_old : l_new;
                        const double l_old = sh_ld_d(sh, FR_L_FAR);
                        sh_st_d(sh, FR_L_FAR, (nn_near != 0u && l_old > l_far_new && l_old < inf) ? l_old : l_far_new);
                    }
                }
                __syncthreads();
                load_l_mid();
                const double lo_c = lo, scale_c = scale;
                const uint32_t kept = fr_partition(
                    src_key, src_id, fn, wsum, n_waves, [&](double k, uint32_t i) -> int { return i == 0u ? -1 : (fr_bin(k, lo_c, scale_c) <= bsel ? 1 : 0); },
                    [&](int c, double k, uint32_t i) {
                        const unsigned long long b = __ballot(c == 1);
                        bool back = false;
                        if (b && to_mid) {  // (uniform) far's band joins mid
                            const uint32_t base = sh_add_uniform(sh, BK_MID_N, (uint32_t)__builtin_popcountll(b), lane);
                            if (c == 1) {
                                const uint32_t pos = base + lane_rank(b, lane);
                                mid_key[pos] = k;
                                mid_id[pos] = i;
                                mid_mn = k < mid_mn ? k : mid_mn;
                            }
                        } else if (b) {
                            const uint32_t base = sh_add_uniform(sh, FR_NEAR_N, (uint32_t)__builtin_popcountll(b), lane);
                            const uint32_t pos = base + lane_rank(b, lane);
                            const bool fits = c == 1 && pos < OC;
                            if (fits) {
                                near_key[pos] = k;
                                near_id[pos] = i;
                                near_mn = k < near_mn ? k : near_mn;
                                near_mx = k > near_mx ? k : near_mx;
                            }
                            back = c == 1 && !fits;
                        }
                        // (an entry that does not fit cannot go back into the list that is being compacted: it is appended behind the
                        // old end of its list and moved down afterwards — only if more than near's capacity of keys share the first bins)
                        if (!to_mid) to_far(back, k, i);
                        if (c < 0 && i != 0u) {  // kept entries (i is their node, never 0)
                            if (from_mid) {
                                mid_mn = k < mid_mn ? k : mid_mn;
                            } else {
                                far_mn = k < far_mn ? k : far_mn;
                                far_mx = k > far_mx ? k : far_mx;
                            }
                        }
                    });
                flush_far();
                flush_near();
                // entries that did not fit into near were appended behind the source's old end (their keys lie in the chosen bins: below
                // l_mid for a source mid, and without a mid list everything goes to far)
                const uint32_t extra = sh[SRC_N] - fn;
                for (uint32_t e = (uint32_t)tid; e < extra; e += (uint32_t)bd) {  // (kept + extra <= fn: the ranges do not overlap)
                    src_key[kept + e] = src_key[fn + e];
                    src_id[kept + e] = src_id[fn + e];
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    sh[SRC_N] = kept + extra;
                    if (sh[FR_NEAR_N] > OC) sh[FR_NEAR_N] = OC;
                }
                __syncthreads();
            }
            nn_near = sh[FR_NEAR_N];
            BK_TICK(tk_refill)
        }

        BK_OPAQUE_TID
        // ---- a SMALL open set (at most BK_PER entries per lane of one wavefront: every light search, and the young rounds of every
        // other): the first wavefront picks the round by itself.  It holds all of near in registers, finds a key below which about a
        // round's worth of entries lies by bisection (a ballot and a population count per step; any threshold will do: which open
        // entries a round takes decides when they are processed, not what the search finds), classifies and compacts with ballots.
        // No histogram, no workgroup scan, two barriers — the sixteen-wavefront selection below costs a light round 3.6-4.7 of its
        // 13-17 us (profiles/r05_step_profile_passes.txt).
        if (nn_near <= (uint32_t)(BK_FAST_PER * PDMPC_WAVE) && (A.bk_flags & 1)) {  // (uniform)
            BK_TICK3(4)
            // (the other wavefronts may be anywhere behind the last barrier, still reading the shared words the boundary's decisions
            // rest on — near's count and key range, the flags —: they are through before the first wavefront rewrites them)
            __syncthreads();
            if (wave == 0) {
                const double lo = sh_ld_d(sh, FR_NEAR_MIN), hi = sh_ld_d(sh, FR_NEAR_MAX);
                double kk[BK_FAST_PER];
                uint32_t ii[BK_FAST_PER];
#pragma unroll
                for (int j = 0; j < BK_FAST_PER; ++j) {
                    const uint32_t e = (uint32_t)j * PDMPC_WAVE + (uint32_t)lane;
                    kk[j] = e < nn_near ? near_key[e] : 0.0;
                    ii[j] = e < nn_near ? near_id[e] : 0u;
                }
                double thr = hi;  // (near's keys do not exceed FR_NEAR_MAX: everything)
                if (nn_near > round_target) {
                    double a = lo, b = hi;
                    for (int it = 0; it < 48; ++it) {  // (uniform: every quantity below is the same in all lanes)
                        const double mid = a + (b - a) * 0.5;
                        if (!(mid > a) || !(mid < b)) break;  // the interval has collapsed (equal keys): take what the last threshold took
                        uint32_t c = 0;
#pragma unroll
                        for (int j = 0; j < BK_FAST_PER; ++j) c += (uint32_t)__builtin_popcountll(__ballot(ii[j] != 0u && kk[j] <= mid));
                        if (c >= round_target) {
                            thr = mid;
                            b = mid;
                            if (c <= 2u * round_target + 16u) break;
                        } else {
                            a = mid;
                        }
                    }
                }
                const bool have_goal = best != 0u;
                const bool check_alive = sh[FR_EVER_INVAL] != 0u;
                const uint32_t epoch_now = sh[BK_ARRIVALS] & 0xffffu;
                ulonglong2* const wcache = (ulonglong2*)A.arena.walk + voff;
                uint32_t pk = 0, pr = 0, n_dead = 0, n_drop = 0;
#pragma unroll
                for (int j = 0; j < BK_FAST_PER; ++j) {  // (as below: straight-line code around the wave-wide walk)
                    const double k = kk[j];
                    const uint32_t i = ii[j];
                    const bool sel = i != 0u && k <= thr;
                    const bool above = have_goal && k > bb;
                    const bool walk = sel && !above && (have_goal || check_alive);
                    if (sel && have_goal && k == bb) atomicOr((uint32_t*)&sh[FR_FLAGS], FRF_TIE);
                    int r = 1;
                    if (have_goal || check_alive) r = bk_classify_wave(F.glink, wcache, VS, F.gkey, gp_path, gp_mp, have_goal, check_alive, best, epoch_now, walk ? i : 0u, k, sh);  // (uniform condition)
                    const int c = i == 0u ? -1 : (sel ? (above ? 3 : r) : 0);
                    const unsigned long long bk = __ballot(c == 0), br = __ballot(c == 1);
                    n_dead += (uint32_t)__builtin_popcountll(__ballot(c == 4));
                    n_drop += (uint32_t)__builtin_popcountll(__ballot(c == 3));
                    if (c == 0) {  // (every entry is in a register: the compacted list may overwrite the old one)
                        const uint32_t pos = pk + lane_rank(bk, lane);
                        near_key[pos] = k;
                        near_id[pos] = i;
                        near_mn = k < near_mn ? k : near_mn;  // (folded into the shared words at the end of the round)
                        near_mx = k > near_mx ? k : near_mx;
                    } else if (c == 1) {
                        const uint32_t pos = pr + lane_rank(br, lane);  // (at most BK_FAST_PER * 64 entries: the ready list holds them)
                        ready[pos] = i;
                        r_flag[pos] = 0u;
                    }
                    pk += (uint32_t)__builtin_popcountll(bk);
                    pr += (uint32_t)__builtin_popcountll(br);
                }
                if (lane == 0) {
                    sh[FR_NEAR_N] = pk;
                    sh_st_d(sh, FR_NEAR_MIN, inf);
                    sh_st_d(sh, FR_NEAR_MAX, 0.0);
                    sh[FR_SEL_BIN] = pr;  // (the refill's word, free here: the round's size for everybody)
                    if (n_dead) sh[FR_DEAD] = sh[FR_DEAD] + n_dead;
                    if (n_drop) sh[FR_DROPPED] = sh[FR_DROPPED] + n_drop;  // comes after the candidate: never popped
                }
            }
            __syncthreads();  // the ready list and the compacted near are in place
            Rn = sh[FR_SEL_BIN];
            BK_TICK3(5)
            BK_TICK(tk_select)
            continue;
        }
        // ---- this round's entries: the smallest keys of near.  Every thread holds BK_PER entries in registers.
        {
            BK_TICK3(4)
            // (the classification below may raise FRF_TIE: behind everybody's boundary decisions.  With a histogram its barrier is
            // in front of the classification anyway; the scan's barrier is in front of thread 0's rewrite of near's count)
            if (!(nn_near > round_target)) __syncthreads();
            const double lo = sh_ld_d(sh, FR_NEAR_MIN);
            double hi = sh_ld_d(sh, FR_NEAR_MAX);
            double kk[BK_PER];
            uint32_t ii[BK_PER];
#pragma unroll
            for (int j = 0; j < BK_PER; ++j) {
                const uint32_t e = (uint32_t)j * (uint32_t)bd + (uint32_t)tid;
                kk[j] = e < nn_near ? near_key[e] : 0.0;
                ii[j] = e < nn_near ? near_id[e] : 0u;
            }
            uint32_t bsel = BK_NB - 1;
            double scale = 0.0;
            const bool use_hist = nn_near > round_target;  // (a round that takes all of near needs no histogram: scale 0 puts every key into bin 0)
            if (use_hist) {
                for (int zoom = 0; zoom < 8; ++zoom) {
                    scale = hi > lo ? (double)BK_NB / (hi - lo) : 0.0;
#pragma unroll
                    for (int j = 0; j < BK_PER; ++j)
                        if (ii[j]) __hip_atomic_fetch_add(&bins[bk_bin(kk[j], lo, scale)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __syncthreads();
                    uint32_t cum;
                    bk_select(bins, round_target, lane, bsel, cum);  // (every wave for itself)
                    if (cum <= 2u * round_target + 16u || scale == 0.0) break;
                    hi = lo + (hi - lo) / (double)BK_NB;  // too many entries share the first bins: look closer
                    __syncthreads();                       // every wave has read the bins ...
                    if (tid < BK_NB) bins[tid] = 0u;
                    __syncthreads();                       // ... and they are clean again
                }
            }
            const bool have_goal = best != 0u;
            const bool check_alive = sh[FR_EVER_INVAL] != 0u;  // some node lost its edge to late areas: its descendants are dead
            const uint32_t epoch_now = sh[BK_ARRIVALS] & 0xffffu;
            ulonglong2* const wcache = (ulonglong2*)A.arena.walk + voff;
            int cls[BK_PER];
            unsigned long long mine = 0;
            uint32_t n_dead = 0, n_drop = 0;
#pragma unroll
            for (int j = 0; j < BK_PER; ++j) {  // (every lane of the wave runs this: straight-line code around the wave-wide walk)
                const double k = kk[j];
                const uint32_t i = ii[j];
                const uint32_t b = bk_bin(k, lo, scale);
                const bool sel = i != 0u && b <= bsel;
                const bool above = have_goal && k > bb;  // above the candidate's path maximum: comes after it
                const bool walk = sel && !above && (have_goal || check_alive);
                if (sel && have_goal && k == bb) atomicOr((uint32_t*)&sh[FR_FLAGS], FRF_TIE);
                int r = 1;
                if (have_goal || check_alive) r = bk_classify_wave(F.glink, wcache, VS, F.gkey, gp_path, gp_mp, have_goal, check_alive, best, epoch_now, walk ? i : 0u, k, sh);  // (uniform condition)
                cls[j] = i == 0u ? -1 : (sel ? (above ? 3 : r) : 0);
                mine += (cls[j] == 0 ? 1ull : 0ull) | (cls[j] == 1 ? (1ull << 32) : 0ull);
                n_dead += cls[j] == 4 ? 1u : 0u;
                n_drop += cls[j] == 3 ? 1u : 0u;
            }
            unsigned long long tot = 0;
            const unsigned long long base = wg_scan_excl(mine, wsum64 + 16, lane, wave, n_waves, tot);  // (its barrier: every entry and every bin has been read)
            if (use_hist && tid < BK_NB) bins[tid] = 0u;  // (clean for the next selection)
            const uint32_t n_keep = (uint32_t)(tot & 0xffffffffull), n_rdy = (uint32_t)(tot >> 32);
            if (tid == 0) {  // (the old list's key range has been read by everybody; nobody touches these words before the next barrier)
                sh[FR_NEAR_N] = n_keep;
                sh_st_d(sh, FR_NEAR_MIN, inf);
                sh_st_d(sh, FR_NEAR_MAX, 0.0);
            }
            {
                uint32_t pk = (uint32_t)(base & 0xffffffffull), pr = (uint32_t)(base >> 32);
                bool over[BK_PER];
#pragma unroll
                for (int j = 0; j < BK_PER; ++j) {
                    over[j] = false;
                    if (cls[j] == 0) {
                        near_key[pk] = kk[j];
                        near_id[pk] = ii[j];
                        near_mn = kk[j] < near_mn ? kk[j] : near_mn;  // (folded into the shared words at the end of the round)
                        near_mx = kk[j] > near_mx ? kk[j] : near_mx;
                        pk += 1u;
                    } else if (cls[j] == 1) {
                        if (pr < RC) {
                            ready[pr] = ii[j];
                            r_flag[pr] = 0u;
                        } else {
                            over[j] = true;  // (only if hundreds of keys are equal to the last bit: they wait in far)
                        }
                        pr += 1u;
                    }
                }
                if (n_rdy > RC) {  // (uniform)
#pragma unroll
                    for (int j = 0; j < BK_PER; ++j) to_far(over[j], kk[j], ii[j]);
                }
            }
            if (n_dead) sh_add(sh, FR_DEAD, n_dead);
            if (n_drop) sh_add(sh, FR_DROPPED, n_drop);  // comes after the candidate: never popped
            Rn = n_rdy < RC ? n_rdy : RC;
            __syncthreads();  // the ready list and the compacted near are in place
            BK_TICK3(5)
            BK_TICK(tk_select)
        }
    }

    // ================= results =================
    int tid_r = tid_k, lane_r = lane_k;  // (opaque once more: see the round loop)
    asm volatile("" : "+v"(tid_r), "+v"(lane_r));
    uint32_t nnodes_raw = sh[FR_NNODES];
    nnodes_raw = nnodes_raw < S.max_nodes ? nnodes_raw : S.max_nodes;
    __syncthreads();
    const bool pb_ran = pb_valid;
    if (!pb_valid) {
        R.n_popped = 0;
        R.n_expanded = nnodes_raw;
    }
    // validity bytes of the LDS-resident nodes go to HBM with the rest (debug read-back of the tree, pdmpc_debug_tree)
    {
        const uint32_t nv = vs_copied ? 0u : (VS.NV < nnodes_raw ? VS.NV : nnodes_raw);
        for (uint32_t i = (uint32_t)tid_r; i < nv; i += (uint32_t)bd) VS.g[i] = VS.l[i];
    }
    {  // work counters: one atomic per wave
        unsigned long long a = t_checks, b = t_pairs;  // (a thread's share stays far below 2^32)
#pragma unroll
        for (int o = PDMPC_WAVE / 2; o > 0; o >>= 1) {
            a += __shfl_xor(a, o);
            b += __shfl_xor(b, o);
        }
        if (lane_r == 0) {
            atomicAdd(A.work_count + 0, a);
            atomicAdd(A.work_count + 1, b);
        }
    }
    if (tid_r == 0) {
        atomicAdd(P.counters + 2, (int)sh[BK_ARRIVALS]);
        atomicAdd(A.work_count + 2, (unsigned long long)sh[FR_PROCESSED]);
        atomicAdd(A.work_count + 3, (unsigned long long)sh[FR_ROUNDS]);
        A.tree_size[slot] = (int32_t)(nnodes_raw | 0x40000000u | (tie_replayed ? 0x20000000u : 0u));  // marks the arena as a raw tree (api.cpp reconstructs the reference's; replayed: the pop sequence is there too)
    }
    if (tid_r == 0 && A.debug_tail) {  // diagnostics in the unused tail of the record (rows HP_MAX - 2 .. HP_MAX of path_nodes); PDMPC_DEBUG_TAIL=1
        double* dbg = X.O->path_nodes[PDMPC_HP_MAX];
        dbg[0] = (double)sh[FR_ROUNDS];
        dbg[1] = (double)sh[FR_PROCESSED];
        dbg[2] = (double)nnodes_raw;
        dbg[3] = (double)sh[FR_NEAR_N];
        dbg[4] = (double)(sh[FR_FAR_N] + sh[BK_MID_N]);
        dbg[5] = (double)sh[FR_FLAGS];
        dbg[6] = (double)sh[BK_ARRIVALS];  // verification events
        dbg[7] = (double)(__builtin_amdgcn_s_memrealtime() - tk[TK_START]);
        for (int i = 0; i < 6; ++i) X.O->path_nodes[PDMPC_HP_MAX - 3][i] = (double)tk2[i];
        if (A.n_helpers > 0) {  // (the seats this search has given out)
            const unsigned long long sw = __hip_atomic_load(board + PDMPC_HB_SEATS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            X.O->path_nodes[PDMPC_HP_MAX - 3][7] = (double)(sw & 0xffffffffull);
        }
        X.O->path_nodes[PDMPC_HP_MAX - 2][0] = (double)(tk[TK_START] - X.rt_kernel_start);
        X.O->path_nodes[PDMPC_HP_MAX - 2][1] = (double)tk[tk_p1];
        X.O->path_nodes[PDMPC_HP_MAX - 2][2] = (double)tk[tk_p2];
        X.O->path_nodes[PDMPC_HP_MAX - 2][3] = (double)tk[tk_p3];
        X.O->path_nodes[PDMPC_HP_MAX - 2][4] = (double)tk[tk_pb];
        X.O->path_nodes[PDMPC_HP_MAX - 2][5] = (double)tk[tk_refill];
        X.O->path_nodes[PDMPC_HP_MAX - 1][0] = (double)(tk[tk_work] + tk[tk_p1] + tk[tk_p2] + tk[tk_p3]);
        X.O->path_nodes[PDMPC_HP_MAX - 1][1] = (double)tk[tk_arrival];
        X.O->path_nodes[PDMPC_HP_MAX - 1][2] = (double)tk[tk_select];
        X.O->path_nodes[PDMPC_HP_MAX - 1][3] = (double)tk[tk_wait];
        X.O->path_nodes[PDMPC_HP_MAX - 1][4] = (double)(tk[TK_MARK] - tk[TK_START]);
        X.O->path_nodes[PDMPC_HP_MAX - 1][5] = (double)tk[tk_pub];                // when the done flag was set ahead of the end (0: at the end), 100 MHz device clock
        X.O->path_nodes[PDMPC_HP_MAX - 1][6] = (double)__builtin_amdgcn_s_memrealtime();  // ... and now (the flag follows within a microsecond unless it is out)
        X.O->path_nodes[PDMPC_HP_MAX - 1][7] = (double)X.rt_kernel_start;
    }
    X.status = status;
    X.n_popped = (int)R.n_popped;
    X.path_ready = pb_ran && goal != 0u;  // (l_path holds G's path: the record need not walk it again)
    X.goal = goal;
    X.nnodes = R.n_expanded;
    X.dep_timeout = dep_timeout;
    X.rec_valid = rec_valid && !dep_timeout;
    X.rec_written = rec_written;
    X.published = sh[BK_PUBLISHED] != 0u;
    return false;
}


// ---------------------------------------------------------------------------------------------------
// Helper workgroups (the workgroups of a launch behind its searches, bulk_body).  A helper takes a SEAT at a search that shares its
// rounds (the board's WANT word; the search with the fewest seats; one fetch-and-add, once), mirrors that search's obstacle soup in
// its own LDS (literal obstacles, lanelet boundary, the areas of the predecessors the owner has incorporated, the expected areas of
// the others) and from then on polls ONE word that nobody else polls: its seat's assignment.  An assignment is a range of the
// round's posted records; the helper runs the owner's own check items on it, leaves one verdict word per entry and the round's number
// in its seat's done word.  It keeps the seat until the search ends, then looks for another search.  A helper never waits for
// anything but memory, so an owner that waits for its seats always gets them; helpers leave when every search of the launch has ended.
template <int CHECKER>
__device__ __forceinline__ void bulk_helper_body(const KernelArgs& A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LDS_AS unsigned char* lsm = (LDS_AS unsigned char*)smem;
    const int tid = threadIdx.x, lane = tid & (PDMPC_WAVE - 1), wave = uni_i(tid >> 6), bd = (int)blockDim.x;
    const int Hp = A.Hp, n_s = A.n_searches;
    const uint32_t CAP = (uint32_t)A.bk_tile;  // records of a range (what fits the staging area)
    // the owners' carve (search_prologue): only the regions a check item reads are filled
    lds_u32* l_path = (lds_u32*)(lsm + PDMPC_LK_PATH);
    lds_i32* l_soff = (lds_i32*)(l_path + PDMPC_HP_MAX + 2);
    lds_i32* l_hoff = l_soff + PDMPC_HP_MAX + 1;
    volatile lds_u32* hs = (volatile lds_u32*)(l_hoff + PDMPC_HP_MAX + 1);
    lds_i32* l_lit = (lds_i32*)(hs + SH_WORDS);
    lds_d2* l_soup = (lds_d2*)(lsm + A.lds.soup);
    lds_d2* t_rec = (lds_d2*)(lsm + PDMPC_LK_NEAR_KEY);              // [bk_tile][3] the range's posted records
    volatile lds_u32* t_flag = (volatile lds_u32*)(lsm + PDMPC_LK_READY);  // [bk_tile] collision flags
    lds_u32* chm = (lds_u32*)(lsm + PDMPC_LK_MISC) + 192;
    BkCheck CK;
    CK.l_area = (const lds_d2*)(lsm + A.lds.area);
    CK.g_area = (const d2*)A.man_area;
    CK.l_soup = l_soup;
    CK.l_soff = l_soff;
    CK.l_hoff = l_hoff;
    CK.l_lit = l_lit;
    CK.areas_in_lds = A.areas_in_lds;
    CK.ll_base = 0;
    CK.ll_len = 0;
    CK.Hp = Hp;
    if (A.areas_in_lds) stage16(lsm + A.lds.area, A.man_area, A.n_man * 3 * PDMPC_VMAX, tid);
    if (tid < SH_WORDS) hs[tid] = 0;
    __syncthreads();
    int my_slot = -1, my_seat = -1;  // (uniform)
    uint32_t last_seq = 0;
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
    // (PDMPC_TUNING=debug_tail=1: where a helper's time goes, 100 MHz ticks summed over its ranges: idle, from the assignment to the
    // soup in place, the range's records, its check items, verdicts + report; work_count[8..12], [13] = ranges)
    const bool hticking = A.debug_tail != 0 && tid == 0;
    unsigned long long hk_mark = hticking ? __builtin_amdgcn_s_memrealtime() : 0ull, hk[5] = {0, 0, 0, 0, 0}, hk_tiles = 0;
    const unsigned long long hk_start = hk_mark;
#define HK_TICK(i)                                                        \
    if (hticking) {                                                       \
        const unsigned long long n__ = __builtin_amdgcn_s_memrealtime(); \
        hk[i] += n__ - hk_mark;                                           \
        hk_mark = n__;                                                    \
    }
    for (;;) {
        // ================= no seat: look for a search that shares its rounds =================
        if (my_slot < 0) {
            if (wave == 0) {
                // one lane per search: candidates say so in their WANT word; the one with the most work per seat is joined (work = nodes
                // processed so far: a step takes as long as its heaviest search, and that is where the helpers belong)
                float best_score = -1.0f;
                int best_rel = -1;
                for (int base = 0; base < n_s; base += PDMPC_WAVE) {  // (uniform trip count)
                    const int k = base + lane;
                    const int s_rel = k < n_s ? (pref + k) % n_s : 0;
                    const unsigned long long* b = A.help_board + (size_t)(A.first + s_rel) * PDMPC_HB_WORDS;
                    const unsigned long long want = __hip_atomic_load(b + PDMPC_HB_WANT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long sw = __hip_atomic_load(b + PDMPC_HB_SEATS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long wt = __hip_atomic_load(b + PDMPC_HB_WEIGHT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t cnt = (uint32_t)(sw >> 32) == A.launch_id ? (uint32_t)(sw & 0xffffffffull) : 0u;
                    // (a seat is kept until the search ends, so a search only gets the seats its work so far entitles it to: its share of the
                    // launch's helpers per 256 nodes processed — plenty of helpers per search (C2): every seat at once; a fifth of a helper
                    // per search (C4): a medium search holds one or two, the 10^5-node search of the step collects all 64 as it grows)
                    uint32_t allowed = 1u + (uint32_t)((wt * (unsigned long long)A.n_helpers) / ((unsigned long long)n_s * (unsigned long long)A.bk_seat_nodes));
                    if (idle >= 8u) allowed = (uint32_t)PDMPC_HB_SEATS_MAX;  // (this helper has found nothing it was entitled to for a while: better seated than idle)
                    allowed = allowed < (uint32_t)PDMPC_HB_SEATS_MAX ? allowed : (uint32_t)PDMPC_HB_SEATS_MAX;
                    const bool cand = k < n_s && want == (unsigned long long)A.launch_id && cnt < allowed;
                    const float score = cand ? (float)(wt + 1ull) / (float)(cnt + 1u) : -1.0f;
                    float c = score;
#pragma unroll
                    for (int o = PDMPC_WAVE / 2; o > 0; o >>= 1) {
                        const float v = __shfl_xor(c, o);
                        c = v > c ? v : c;
                    }
                    const unsigned long long m = __ballot(cand && score == c);
                    if (m && c > best_score) {  // (uniform)
                        best_score = c;
                        best_rel = (pref + base + (int)__builtin_ctzll(m)) % n_s;
                    }
                }
                uint32_t got = 0xffffffffu;
                if (best_rel >= 0) {  // (uniform) take a seat: one fetch-and-add; a stale word (another launch's) gives nothing
                    unsigned long long old = 0;
                    if (lane == 0)
                        old = __hip_atomic_fetch_add(A.help_board + (size_t)(A.first + best_rel) * PDMPC_HB_WORDS + PDMPC_HB_SEATS, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t lo32 = uni_u((uint32_t)old), hi32 = uni_u((uint32_t)(old >> 32));
                    if (hi32 == A.launch_id && lo32 < (uint32_t)PDMPC_HB_SEATS_MAX) got = lo32;
                }
                uint32_t cmd = 0;
                if (got != 0xffffffffu) {
                    cmd = 1;
                } else {
                    const uint32_t fin = __hip_atomic_load(A.help_finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (fin - A.help_fin_base >= (uint32_t)n_s) cmd = 2;
                }
                if (lane == 0) {
                    hs[HS_SLOT] = (uint32_t)(A.first + (best_rel >= 0 ? best_rel : 0));
                    hs[HS_FIRST] = got;
                    hs[HS_CMD] = cmd;
                }
            }
            __syncthreads();
            const uint32_t cmd = hs[HS_CMD];
            if (cmd == 2u) break;
            if (cmd == 0u) {  // nobody shares: look again in a while (a search announces itself once; nothing is lost by being a few microseconds late)
                __builtin_amdgcn_s_sleep(127);
                __builtin_amdgcn_s_sleep(127);
                if (++idle > (A.spin_limit >> 4)) break;  // (uniform) the searches never came: leave; they do without helpers
                __syncthreads();
                continue;
            }
            my_slot = (int)hs[HS_SLOT];
            my_seat = (int)hs[HS_FIRST];
            last_seq = 0;
            idle = 0;
            // ---- the search's obstacle soup (search_prologue, parts 3 and 4); its predecessors' slots start empty
            const DevVehicle* __restrict__ V = A.veh + my_slot;
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
            CK.ll_base = off;
            CK.ll_len = V->ll_len;
            __syncthreads();
            const d2 nanpt = d2{__longlong_as_double(0x7ff8000000000000LL), __longlong_as_double(0x7ff8000000000000LL)};
            for (int idx = tid; idx < Hp * pred_cols; idx += bd) {
                const int k = idx / pred_cols;
                l_soup[l_soff[k] + l_lit[k] + (idx - k * pred_cols)] = nanpt;
            }
            __syncthreads();
            P.pred = A.pred + V->pred_off;
            P.n_pred = V->n_pred;
            if (A.bk_tentative) bk_tentative_areas(A, P, V->n_pred >= 64 ? ~0ull : ((1ull << V->n_pred) - 1ull), tid, bd);  // (as the owner: expected areas until the real ones are in)
            cur_mask = 0;
            __syncthreads();
            if (tid < 8) {  // the chunk table of this soup (bulk_search)
                const int ls = tid;
                uint32_t mx = 0;
                for (int k = 1; k <= Hp; ++k) {
                    const int M_k = l_soff[k] - l_soff[k - 1], Hk = l_hoff[k] - l_hoff[k - 1];
                    int n0, n1, n2;
                    bk_item_counts<CHECKER>(M_k, Hk, CK.ll_len, n0, n1, n2);
                    const int Sg = 1 << ls;
                    const uint32_t ch = (uint32_t)(((n0 + Sg - 1) >> ls) + ((n1 + Sg - 1) >> ls) + ((n2 + Sg - 1) >> ls));
                    mx = ch > mx ? ch : mx;
                }
                chm[ls] = mx;
            }
            __syncthreads();
            if (hticking) hk_mark = __builtin_amdgcn_s_memrealtime();
            continue;
        }
        // ================= seated: this seat's assignment word =================
        unsigned long long* board = A.help_board + (size_t)my_slot * PDMPC_HB_WORDS;
        if (wave == 0) {
            const unsigned long long w = __hip_atomic_load(board + PDMPC_HB_ASSIGN + my_seat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t seq = (uint32_t)(w >> 40);
            uint32_t cmd = 0;
            if (seq != last_seq && seq != 0u) {
                cmd = 1;
#if PDMPC_BK_POST_FENCES
                // what the owner wrote before it assigned (and the predecessors it had seen) is visible from here on; not on an idle
                // poll: the fence empties this XCD's L2, which searches on neighbouring CUs share
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
                const unsigned long long mask = __hip_atomic_load(board + PDMPC_HB_MASK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 0) {
                    hs[HS_FIRST] = (uint32_t)((w >> 20) & 0xfffffull);
                    hs[HS_COUNT] = (uint32_t)(w & 0xfffffull);
                    hs[HS_TICKET] = seq;
                    hs[HS_MASK_LO] = (uint32_t)mask;
                    hs[HS_MASK_HI] = (uint32_t)(mask >> 32);
                }
            } else if ((idle & 15u) == 15u) {  // (now and then) has the search ended?
                const unsigned long long want = __hip_atomic_load(board + PDMPC_HB_WANT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (want != (unsigned long long)A.launch_id) cmd = 3;
            }
            if (lane == 0) hs[HS_CMD] = cmd;
        }
        __syncthreads();
        const uint32_t cmd = hs[HS_CMD];
        if (cmd == 3u) {  // the seat's search is over: look for another one
            my_slot = -1;
            my_seat = -1;
            idle = 0;
            __syncthreads();
            continue;
        }
        if (cmd == 0u) {
            if (idle < 256u)
                __builtin_amdgcn_s_sleep(1);
            else
                __builtin_amdgcn_s_sleep(16);
            if (++idle > A.spin_limit) break;  // (uniform) must never happen: the search ends and says so
            __syncthreads();
            continue;
        }
        idle = 0;
        HK_TICK(0)
        const uint32_t first = hs[HS_FIRST], Rt = hs[HS_COUNT] < CAP ? hs[HS_COUNT] : CAP, seq = hs[HS_TICKET];
        const unsigned long long mask = ((unsigned long long)hs[HS_MASK_HI] << 32) | hs[HS_MASK_LO];
        last_seq = seq;
        if (mask != cur_mask) {  // (within a launch a search's set of incorporated predecessors only grows)
            bk_incorporate(P.out, P.pred, P.l_soup, P.l_soff, P.l_lit, Hp, mask & ~cur_mask, tid, bd, nullptr);  // (loads that are coherent by themselves)
            cur_mask = mask;
        }
        HK_TICK(1)
        // ---- the range: its records into LDS, its check items, its verdicts
        {
            const d2* post = (const d2*)A.bk_post + ((size_t)my_slot * (size_t)A.bk_ready_cap + first) * 3u;
            for (uint32_t i = (uint32_t)tid; i < Rt * 3u; i += (uint32_t)bd) t_rec[i] = bk_post_load(post + i);
            for (uint32_t i = (uint32_t)tid; i < Rt; i += (uint32_t)bd) t_flag[i] = 0u;
        }
        __syncthreads();
        HK_TICK(2)
        if (Rt) {
            const BkPostSrc psrc{t_rec};
            const int ls = bk_chunk_shift(chm, Rt, (uint32_t)bd);
            const unsigned long long pend = A.bk_tentative ? (P.n_pred >= 64 ? ~0ull : ((1ull << P.n_pred) - 1ull)) & ~mask : 0ull;
            bk_check_items<CHECKER>(CK, psrc, t_flag, 0u, Rt, ls, chm[ls], pend, tid, bd);
        }
        __syncthreads();
        HK_TICK(3)
        {
            uint32_t* verdict = A.help_verdict + (size_t)my_slot * PDMPC_HELP_CAP + first;
            for (uint32_t i = (uint32_t)tid; i < Rt; i += (uint32_t)bd) {  // 1 collision-free, 2 collides, 3 crosses expected areas only
                const uint32_t fl = t_flag[i];
                __hip_atomic_store(verdict + i, (fl & 1u) ? 2u : ((fl & 2u) ? 3u : 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's verdicts have been acknowledged (the barrier alone does not wait for them) ...
        __syncthreads();                                     // ... every wave's have: the seat's done word can say so
        if (tid == 0) {
            __hip_atomic_store(board + PDMPC_HB_DONE + my_seat, (unsigned long long)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hs[HS_CMD] = 0;
        }
        __syncthreads();
        HK_TICK(4)
        hk_tiles += 1;
    }
    if (hticking) {
        atomicAdd(A.work_count + 7, __builtin_amdgcn_s_memrealtime() - hk_start);  // (this helper's lifetime)
        for (int i = 0; i < 5; ++i) atomicAdd(A.work_count + 8 + i, hk[i]);
        atomicAdd(A.work_count + 13, hk_tiles);
    }
#undef HK_TICK
}

template <int NW, int CHECKER>
__device__ __forceinline__ void bulk_body(const KernelArgs& A) {
    // The workgroups behind the searches are their helpers: one launch, so the helpers are dispatched with (for launches with more
    // searches than CUs: right behind) the searches they serve — a helper kernel of its own on a second stream now and then shared a
    // hardware queue with the launch stream and started when the searches were through (one step in a hundred without helpers).
    // (bk_helpers_first of the helpers come in front of the searches: a launch of more searches than CUs, whose finished searches hold
    // their CUs while they wait for predecessors, leaves the helpers behind the searches no CU until the step is nearly over)
    if ((int)blockIdx.x < A.bk_helpers_first || (int)blockIdx.x >= A.bk_helpers_first + A.n_searches) {  // (uniform over the workgroup)
        bulk_helper_body<CHECKER>(A);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Ctx X;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    search_prologue(A, X, (LDS_AS unsigned char*)smem);
    X.rt_kernel_start = rt0;
    const int wave = X.wave;
    lds_u32* ref_ids = (lds_u32*)(X.lsm + PDMPC_LK_MISC) + 224;  // behind the chunk table (nothing else uses those words)
    const bool tie = bulk_search<NW, CHECKER>(A, X, ref_ids);
    (void)tie;  // (equal keys are resolved inside the search: bk_replay)
    __syncthreads();
    if (wave != 0) return;
    {  // (opaque, as in the round loop: nothing derived from the lane index in the prologue is worth a register until here)
        int l__ = X.lane;
        asm volatile("" : "+v"(l__));
        X.lane = l__;
    }
    if (!X.rec_valid) bk_write_record(A, X, X.goal, X.status, X.dep_timeout, (uint32_t)X.n_popped, X.nnodes, X.path_ready, ref_ids, X.rec_written, X.published, X.lane);
    bk_publish(A, X, X.status, X.dep_timeout);
    if (X.lane == 0 && A.n_helpers > 0) {
        __hip_atomic_store(A.help_board + (size_t)X.slot * PDMPC_HB_WORDS + PDMPC_HB_WANT, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (the seats are free to go)
        atomicAdd(A.help_finished, 1u);
    }
}

}  // namespace

// One kernel per (checker, successor-mask words) in a translation unit of its own (bulk_kernel.hip: InterX, one mask word — every
// BASELINE road-network configuration; bulk_kernel_wide.hip: InterX, automata of more than 64 trims; bulk_kernel_sat.hip: the
// separating-axis checker, any automaton), so that they compile side by side and each gets its own register allocation.
#ifndef PDMPC_BULK_KERNEL_ATTR
#define PDMPC_BULK_KERNEL_ATTR
#endif
#define PDMPC_BULK_KERNEL(NAME, LAUNCHER, NW, CHECKER, MAXWAVES)                                                                                   \
    extern "C" __global__ __launch_bounds__(PDMPC_WAVE * (MAXWAVES)) PDMPC_BULK_KERNEL_ATTR void NAME(const KernelArgs A) { bulk_body<NW, CHECKER>(A); } \
    extern "C" int LAUNCHER(const KernelArgs* args, int count, void* stream, uint32_t* lds_high_water) {                                \
        if (count <= 0) return 0;                                                                                                        \
        typedef void (*kernel_t)(const KernelArgs);                                                                                      \
        kernel_t fn = NAME;                                                                                                              \
        /* (the attribute is a maximum: raised when a launch needs more than any before it; the high-water mark lives in the handle) */ \
        uint32_t& have = lds_high_water[0];                                                                                              \
        if (args->lds.total > have) {                                                                                                    \
            hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)args->lds.total);       \
            if (e != hipSuccess) return (int)e;                                                                                          \
            have = args->lds.total;                                                                                                      \
        }                                                                                                                                \
        hipLaunchKernelGGL(fn, dim3(count + (args->n_helpers > 0 ? args->n_helpers : 0)), dim3(PDMPC_WAVE * args->n_waves), args->lds.total, (hipStream_t)stream, *args); \
        return (int)hipGetLastError();                                                                                                   \
    }
