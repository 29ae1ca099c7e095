// search_kernel.hip — one workgroup (one 64-lane wavefront) plans one vehicle.
//
// The kernel restates, for gfx950, the reference's optimal graph search:
//   GraphSearch.do_graph_search      hlc/optimizer/graph_search/GraphSearch.m:23-107
//   GraphSearch.eval_edge_exact      GraphSearch.m:111-196
//   expand_node                      graph_search/expand_node.m:1-91
//   are_constraints_satisfied_sat    graph_search/are_constraints_satisfied_sat.m:1-68  (+ intersect_sat.m,
//                                    common/intersect_lanelet_boundary.m)
//   are_constraints_satisfied_interx graph_search/are_constraints_satisfied_interx.m:1-39 (+ InterX.m:48-110)
//   std::priority_queue semantics    priority_queue/priority_queue_interface_mex.cpp:19-31 (libstdc++ heap)
//   return_path_to / return_path_area / Tree.path_to_root
//
// Mapping to the hardware (DESIGN.md has the full picture):
//   * MPA tables, the vehicle's obstacle "soup", the open list (binary heap) and the first NL tree nodes
//     live in LDS; nodes are written through to HBM (SoA, coalesced across the children of an expansion),
//     heap entries beyond HL and nodes beyond NL spill to HBM.
//   * lanes parallelise the inner loops: one lane per obstacle segment (InterX) / per separating axis
//     (SAT), one lane per successor trim (expansion), one lane per polygon vertex (area transform);
//     wave votes (ballot) reduce the collision predicates.
//   * the sequential part (heap sift, pop -> check -> expand) runs wave-uniform with scalar branches.
//   * floating point: every expression keeps the reference's operation order; the file is compiled with
//     -ffp-contract=off so no FMA is formed, sqrt and division are IEEE-correct, sin/cos come from
//     include/pdmpc_math.h.  Results are bit-identical to the CPU oracle.
#include <hip/hip_runtime.h>

#include "../../include/pdmpc_math.h"
#include "pdmpc_device.h"

namespace {

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
__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }
__device__ __forceinline__ bool is_nan(double v) { return v != v; }

// ---------------------------------------------------------------------------------------------------
// Per-vehicle search state.  Pointers named l* point into LDS, g* into this vehicle's HBM slices.
struct Search {
    // tree nodes
    double *lx, *ly, *lyaw, *lg, *lh, *lcs, *lsn;
    uint32_t* lparent;
    uint16_t* ltk;
    NodeArena g;
    uint32_t NL, max_nodes;
    // open list
    double* lkey;
    uint32_t* lid;
    uint32_t HL;
    uint32_t heap_len;
    int lane;
};

#define NODE_RD(S, field, i) (((i) < (S).NL) ? (S).l##field[(i)] : (S).g.field[(i)])

__device__ __forceinline__ double heap_key_at(const Search& S, uint32_t i) { return i < S.HL ? S.lkey[i] : S.g.heap_key[i]; }
__device__ __forceinline__ uint32_t heap_id_at(const Search& S, uint32_t i) { return i < S.HL ? S.lid[i] : S.g.heap_id[i]; }
__device__ __forceinline__ void heap_set(Search& S, uint32_t i, double key, uint32_t id) {
    if (S.lane == 0) {
        if (i < S.HL) {
            S.lkey[i] = key;
            S.lid[i] = id;
        } else {
            S.g.heap_key[i] = key;
            S.g.heap_id[i] = id;
        }
    }
}
// make lane 0's heap writes visible to the whole wave (LDS: in-order DS queue; HBM spill: same-CU L1)
__device__ __forceinline__ void heap_fence(const Search& S) {
    if (S.heap_len > S.HL) __threadfence_block();
    __builtin_amdgcn_wave_barrier();
}

// std::push_heap: libstdc++ __push_heap(first, hole, top = 0, value) with comp(a, b) = a.key > b.key
// (priority_queue_interface_mex.cpp:23-29).  All indices are wave-uniform.
__device__ void heap_sift_up(Search& S, uint32_t hole, double key, uint32_t id) {
    while (hole > 0) {
        const uint32_t parent = (hole - 1) >> 1;
        const double pk = uni_d(heap_key_at(S, parent));
        if (!(pk > key)) break;  // strict: equal keys do not move up
        const uint32_t pid = uni_u(heap_id_at(S, parent));
        heap_set(S, hole, pk, pid);
        hole = parent;
    }
    heap_set(S, hole, key, id);
}

__device__ void heap_push(Search& S, uint32_t id, double key) {
    const uint32_t hole = S.heap_len;
    S.heap_len = hole + 1;
    heap_sift_up(S, hole, key, id);
    heap_fence(S);
}

// std::pop_heap + pop_back: libstdc++ __pop_heap -> __adjust_heap(first, 0, len, value) -> __push_heap
__device__ void heap_pop(Search& S) {
    const uint32_t len = S.heap_len - 1;  // length after the pop
    S.heap_len = len;
    if (len == 0) return;
    const double vkey = uni_d(heap_key_at(S, len));
    const uint32_t vid = uni_u(heap_id_at(S, len));
    uint32_t hole = 0, child = 0;
    const uint32_t half = (len - 1) >> 1;
    while (child < half) {
        child = 2 * (child + 1);  // right child
        double ck = uni_d(heap_key_at(S, child));
        const double lk = uni_d(heap_key_at(S, child - 1));
        if (ck > lk) {  // comp(right, left): right is worse -> take left
            child--;
            ck = lk;
        }
        const uint32_t cid = uni_u(heap_id_at(S, child));
        heap_set(S, hole, ck, cid);
        hole = child;
    }
    if ((len & 1u) == 0 && child == ((len - 2) >> 1)) {  // lone left child at the bottom
        child = 2 * (child + 1);
        const double ck = uni_d(heap_key_at(S, child - 1));
        const uint32_t cid = uni_u(heap_id_at(S, child - 1));
        heap_set(S, hole, ck, cid);
        hole = child - 1;
    }
    heap_fence(S);
    heap_sift_up(S, hole, vkey, vid);
    heap_fence(S);
}

// ---------------------------------------------------------------------------------------------------
// InterX.m:63-76,108-110 for one curve pair: L1 = shape (V points, LDS), L2 = soup (M points, LDS).
// One lane per L2 segment; strict "< 0" products; NaN separators make every comparison false.
__device__ bool interx_wave(const d2* sh, int V, const d2* L2, int M, int lane) {
    if (M < 2 || V < 2) return false;
    for (int base = 0; base < M - 1; base += PDMPC_WAVE) {
        const int j = base + lane;
        bool hit = false;
        if (j < M - 1) {
            const d2 q0 = L2[j], q1 = L2[j + 1];
            const double dx2 = q1.x - q0.x, dy2 = q1.y - q0.y;
            const double S2 = dx2 * q0.y - dy2 * q0.x;
            d2 p0 = sh[0];
            for (int i = 0; i < V - 1; ++i) {
                const d2 p1 = sh[i + 1];
                const double dx1 = p1.x - p0.x, dy1 = p1.y - p0.y;
                const double S1 = dx1 * p0.y - dy1 * p0.x;
                const double a0 = dx1 * q0.y - dy1 * q0.x;
                const double a1 = dx1 * q1.y - dy1 * q1.x;
                const bool c1 = (a0 - S1) * (a1 - S1) < 0;
                const double b0 = p0.y * dx2 - p0.x * dy2;
                const double b1 = p1.y * dx2 - p1.x * dy2;
                const bool c2 = (b0 - S2) * (b1 - S2) < 0;
                hit = hit || (c1 && c2);
                p0 = p1;
            }
        }
        if (wave_any(hit)) return true;
    }
    return false;
}

// intersect_sat.m:1-42 for shape (V1 points) vs one polygon o (V2 points): one lane per separating axis.
// An axis separates iff min1 - max2 > 0 or min2 - max1 > 0 (:33-40); a zero-length edge gives a NaN axis whose
// comparisons are false.  collide <=> no axis of either polygon separates.
__device__ bool sat_pair_wave(const d2* sh, int V1, const d2* o, int V2, int lane) {
    const int A = V1 + V2;
    for (int base = 0; base < A; base += PDMPC_WAVE) {
        const int a = base + lane;
        bool sep = false;
        if (a < A) {
            d2 e0, e1;
            if (a < V1) {
                e0 = sh[a];
                e1 = sh[(a + 1 == V1) ? 0 : a + 1];
            } else {
                const int b = a - V1;
                e0 = o[b];
                e1 = o[(b + 1 == V2) ? 0 : b + 1];
            }
            const double ex = e1.x - e0.x, ey = e1.y - e0.y;
            const double ax = -ey, ay = ex;
            const double nrm = sqrt(ax * ax + ay * ay);
            const double nx = ax / nrm, ny = ay / nrm;
            double minS = 0, maxS = 0, minO = 0, maxO = 0;
            for (int v = 0; v < V1; ++v) {
                const d2 p = sh[v];
                const double d = nx * p.x + ny * p.y;
                if (v == 0) {
                    minS = d;
                    maxS = d;
                } else {
                    minS = (d < minS) ? d : minS;
                    maxS = (d > maxS) ? d : maxS;
                }
            }
            for (int v = 0; v < V2; ++v) {
                const d2 p = o[v];
                const double d = nx * p.x + ny * p.y;
                if (v == 0) {
                    minO = d;
                    maxO = d;
                } else {
                    minO = (d < minO) ? d : minO;
                    maxO = (d > maxO) ? d : maxO;
                }
            }
            sep = (minS - maxO > 0) || (minO - maxS > 0);
        }
        if (wave_any(sep)) return false;
    }
    return true;
}

// are_constraints_satisfied_sat.m:15-35: every polygon of the step's soup (static then dynamic obstacles).
__device__ bool sat_soup_wave(const d2* sh, int V1, const d2* soup, int M, int lane) {
    int pos = 0;
    while (pos < M) {
        // next NaN separator at or after pos
        int end = M;
        for (int base = pos; base < M; base += PDMPC_WAVE) {
            const int j = base + lane;
            const bool sepr = (j < M) && is_nan(soup[j].x);
            const unsigned long long b = __ballot(sepr);
            if (b) {
                end = base + (int)__builtin_ctzll(b);
                break;
            }
        }
        const int V2 = end - pos;
        if (V2 > 0 && sat_pair_wave(sh, V1, soup + pos, V2, lane)) return true;
        pos = end + 1;
    }
    return false;
}

// intersect_lanelet_boundary.m:1-56 on the soup [left, NaN, right, NaN]: one lane per boundary segment,
// AABB pre-filter (:20,40) then intersect_sat(shape, segment) with the segment as a 2-point polygon.
__device__ bool sat_boundary_wave(const d2* sh, int V1, const d2* ll, int M, int lane) {
    if (M < 2) return false;
    double max_x = sh[0].x, min_x = sh[0].x, max_y = sh[0].y, min_y = sh[0].y;
    for (int v = 1; v < V1; ++v) {
        const d2 p = sh[v];
        max_x = (p.x > max_x) ? p.x : max_x;
        min_x = (p.x < min_x) ? p.x : min_x;
        max_y = (p.y > max_y) ? p.y : max_y;
        min_y = (p.y < min_y) ? p.y : min_y;
    }
    for (int base = 0; base < M - 1; base += PDMPC_WAVE) {
        const int j = base + lane;
        bool hit = false;
        if (j < M - 1) {
            const d2 q0 = ll[j], q1 = ll[j + 1];
            const bool real = !(is_nan(q0.x) || is_nan(q1.x));
            const bool reject = (max_x < q0.x && max_x < q1.x) || (min_x > q0.x && min_x > q1.x) ||
                                (max_y < q0.y && max_y < q1.y) || (min_y > q0.y && min_y > q1.y);
            if (real && !reject) {
                bool sep = false;
                const int A = V1 + 2;
                for (int a = 0; a < A; ++a) {
                    d2 e0, e1;
                    if (a < V1) {
                        e0 = sh[a];
                        e1 = sh[(a + 1 == V1) ? 0 : a + 1];
                    } else if (a == V1) {
                        e0 = q0;
                        e1 = q1;
                    } else {
                        e0 = q1;
                        e1 = q0;
                    }
                    const double ex = e1.x - e0.x, ey = e1.y - e0.y;
                    const double ax = -ey, ay = ex;
                    const double nrm = sqrt(ax * ax + ay * ay);
                    const double nx = ax / nrm, ny = ay / nrm;
                    double minS = 0, maxS = 0;
                    for (int v = 0; v < V1; ++v) {
                        const d2 p = sh[v];
                        const double d = nx * p.x + ny * p.y;
                        if (v == 0) {
                            minS = d;
                            maxS = d;
                        } else {
                            minS = (d < minS) ? d : minS;
                            maxS = (d > maxS) ? d : maxS;
                        }
                    }
                    const double d0 = nx * q0.x + ny * q0.y;
                    const double d1 = nx * q1.x + ny * q1.y;
                    const double minO = (d1 < d0) ? d1 : d0;
                    const double maxO = (d1 > d0) ? d1 : d0;
                    sep = sep || (minS - maxO > 0) || (minO - maxS > 0);
                }
                hit = !sep;
            }
        }
        if (wave_any(hit)) return true;
    }
    return false;
}

// copy `count` 16-byte elements HBM -> LDS, lane-strided (coalesced 1 KiB per wave instruction)
__device__ __forceinline__ void stage16(void* dst_lds, const void* src, int count, int lane) {
    d2* d = (d2*)dst_lds;
    const d2* s = (const d2*)src;
    for (int i = lane; i < count; i += PDMPC_WAVE) d[i] = s[i];
}

}  // namespace

extern "C" __global__ __launch_bounds__(PDMPC_WAVE) void pdmpc_search_kernel(const KernelArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int slot = A.first + blockIdx.x;
    const int Hp = A.Hp;
    const int n = A.n_trims;
    const int nw = A.n_words;
    const DevVehicle* __restrict__ V = A.veh + slot;

    // ---- LDS carve
    uint64_t* l_mask = (uint64_t*)(smem + A.lds.mask);
    int16_t* l_mi = (int16_t*)(smem + A.lds.man_index);
    DevManPose* l_pose = (DevManPose*)(smem + A.lds.pose);
    const d2* area_tab = A.areas_in_lds ? (const d2*)(smem + A.lds.area) : (const d2*)A.man_area;
    double* l_rx = (double*)(smem + A.lds.ref);
    double* l_ry = l_rx + PDMPC_HP_MAX;
    double* l_dtv = l_ry + PDMPC_HP_MAX;
    d2* l_shA = (d2*)(smem + A.lds.shape);
    d2* l_shB = l_shA + PDMPC_VMAX;
    uint32_t* l_path = (uint32_t*)(smem + A.lds.path);
    int32_t* l_soff = (int32_t*)(l_path + PDMPC_HP_MAX + 2);  // soup offsets [Hp+1], hdv offsets [Hp+1]
    int32_t* l_hoff = l_soff + PDMPC_HP_MAX + 1;
    d2* l_soup = (d2*)(smem + A.lds.soup);

    Search S;
    S.lx = (double*)(smem + A.lds.nx);
    S.ly = (double*)(smem + A.lds.ny);
    S.lyaw = (double*)(smem + A.lds.nyaw);
    S.lg = (double*)(smem + A.lds.ng);
    S.lh = (double*)(smem + A.lds.nh);
    S.lcs = (double*)(smem + A.lds.ncs);
    S.lsn = (double*)(smem + A.lds.nsn);
    S.lparent = (uint32_t*)(smem + A.lds.nparent);
    S.ltk = (uint16_t*)(smem + A.lds.ntk);
    S.lkey = (double*)(smem + A.lds.heap_key);
    S.lid = (uint32_t*)(smem + A.lds.heap_id);
    S.NL = (uint32_t)A.NL;
    S.HL = (uint32_t)A.HL;
    S.max_nodes = A.max_nodes;
    S.lane = lane;
    const size_t voff = (size_t)slot * A.max_nodes;
    S.g.x = A.arena.x + voff;
    S.g.y = A.arena.y + voff;
    S.g.yaw = A.arena.yaw + voff;
    S.g.g = A.arena.g + voff;
    S.g.h = A.arena.h + voff;
    S.g.cs = A.arena.cs + voff;
    S.g.sn = A.arena.sn + voff;
    S.g.parent = A.arena.parent + voff;
    S.g.tk = A.arena.tk + voff;
    S.g.heap_key = A.arena.heap_key + voff;
    S.g.heap_id = A.arena.heap_id + voff;

    pdmpc_vehicle_out* __restrict__ O = A.out + slot;

    // ---- prologue 1: stage MPA tables (coalesced 16-byte copies; sizes are padded to 16 B by the host)
    {
        const int mask_bytes = Hp * n * nw * 8;
        stage16(l_mask, A.succ_mask, (mask_bytes + 15) / 16, lane);
        stage16(l_mi, A.man_index, (n * n * 2 + 15) / 16, lane);
        stage16(l_pose, A.man_pose, A.n_man * 2, lane);
        if (A.areas_in_lds) stage16(smem + A.lds.area, A.man_area, A.n_man * 3 * PDMPC_VMAX, lane);
    }
    // ---- prologue 2: vehicle record, result record defaults
    if (lane < Hp) {
        l_rx[lane] = V->ref_x[lane];
        l_ry[lane] = V->ref_y[lane];
        l_dtv[lane] = A.dt * V->v_ref[lane];  // options.dt_seconds * iter.v_ref(k)   expand_node.m:70
    }
    {
        // zero the record; y_predicted starts as NaN (ControlResultsInfo.m:40)
        double* od = (double*)O;
        const int nd = (int)(sizeof(pdmpc_vehicle_out) / 8);
        const int y0 = (int)(offsetof(pdmpc_vehicle_out, y_predicted) / 8);
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        for (int i = lane; i < nd; i += PDMPC_WAVE) od[i] = (i >= y0 && i < y0 + PDMPC_HP_MAX * 3) ? qnan : 0.0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // later result stores hit the same bytes from other lanes
    }
    __syncthreads();

    // ---- prologue 3: obstacle soup of every step: [literal polygons + NaN][predecessor areas padded to VMAX]
    const int n_pred = V->n_pred;
    const int pred_cols = n_pred * PDMPC_VMAX;
    {
        int off = 0;
        for (int k = 0; k < Hp; ++k) {
            const int a = V->lit_off[k], b = V->lit_off[k + 1];
            if (lane == 0) l_soff[k] = off;
            stage16(l_soup + off, (const d2*)A.points + a, b - a, lane);
            off += (b - a) + pred_cols;
        }
        if (lane == 0) l_soff[Hp] = off;
        for (int k = 0; k < Hp; ++k) {
            const int a = V->hdv_off[k], b = V->hdv_off[k + 1];
            if (lane == 0) l_hoff[k] = off;
            stage16(l_soup + off, (const d2*)A.points + a, b - a, lane);
            off += (b - a);
        }
        if (lane == 0) l_hoff[Hp] = off;
        // lanelet soup last
        stage16(l_soup + off, (const d2*)A.points + V->ll_off, V->ll_len, lane);
        if (lane == 0) l_path[PDMPC_HP_MAX + 1] = (uint32_t)off;
    }
    __syncthreads();
    const int ll_base = (int)l_path[PDMPC_HP_MAX + 1];
    const int ll_len = V->ll_len;

    // ---- prologue 4: wait for sequential predecessors and append their solved areas (PrioritizedController.m:476-491)
    bool dep_timeout = false;
    if (n_pred > 0) {
        for (int p = 0; p < n_pred; ++p) {
            const int ps = A.pred[V->pred_off + p];
            uint32_t spins = 0;
            while (__hip_atomic_load(A.done_flag + ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != A.epoch) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > A.spin_limit) {
                    dep_timeout = true;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        for (int idx = lane; idx < Hp * pred_cols; idx += PDMPC_WAVE) {
            const int k = idx / pred_cols;
            const int r = idx - k * pred_cols;
            const int p = r / PDMPC_VMAX;
            const int v = r - p * PDMPC_VMAX;
            const int ps = A.pred[V->pred_off + p];
            const pdmpc_vehicle_out* PO = A.out + ps;
            const int cols = PO->shape_cols[k];
            d2 pt;
            pt.x = qnan;
            pt.y = qnan;
            if (v < cols) {
                pt.x = PO->shapes[k][0][v];
                pt.y = PO->shapes[k][1][v];
            }
            const int lit = V->lit_off[k + 1] - V->lit_off[k];
            l_soup[l_soff[k] + lit + r] = pt;
        }
        __syncthreads();
    }

    // ---- root node (GraphSearch.m:34-46)
    uint32_t nnodes = 1;
    {
        if (lane == 0) {
            const uint16_t tk = (uint16_t)V->trim0;
            S.g.x[0] = V->x0;
            S.g.y[0] = V->y0;
            S.g.yaw[0] = V->yaw0;
            S.g.g[0] = 0.0;
            S.g.h[0] = 0.0;
            S.g.parent[0] = 0;
            S.g.tk[0] = tk;
            S.lx[0] = V->x0;
            S.ly[0] = V->y0;
            S.lyaw[0] = V->yaw0;
            S.lg[0] = 0.0;
            S.lh[0] = 0.0;
            S.lparent[0] = 0;
            S.ltk[0] = tk;
            S.lkey[0] = 0.0;
            S.lid[0] = 1;
        }
        S.heap_len = 1;
    }
    __syncthreads();

    int status = PDMPC_OK;
    int n_popped = 0;
    uint32_t goal = 0;

    // ---- main loop (GraphSearch.m:53-107)
    for (;;) {
        if (S.heap_len == 0) {  // pop on an empty queue returns -1 (mex.cpp:87-93)  GraphSearch.m:57-61
            status = PDMPC_EXHAUSTED;
            break;
        }
        const uint32_t cur = uni_u(heap_id_at(S, 0));  // 1-based node id
        heap_pop(S);
        if (A.trace_cap > 0 && n_popped < A.trace_cap && lane == 0) A.pop_trace[(size_t)slot * A.trace_cap + n_popped] = (int32_t)cur;
        ++n_popped;
        const uint32_t c0 = cur - 1;
        const uint32_t par = uni_u(NODE_RD(S, parent, c0));
        const uint32_t ctk = uni_u((uint32_t)NODE_RD(S, tk, c0));
        const int cTrim = (int)(ctk & 1023u);  // 1-based
        const int cK = (int)(ctk >> 10);

        // ---- eval_edge_exact (GraphSearch.m:111-196)
        bool valid = true;
        if (par) {
            const uint32_t p0 = par - 1;
            const double pX = uni_d(NODE_RD(S, x, p0));
            const double pY = uni_d(NODE_RD(S, y, p0));
            const double c = uni_d(NODE_RD(S, cs, p0));  // cos/sin(pYaw), cached when the parent was expanded
            const double s = uni_d(NODE_RD(S, sn, p0));
            const int pTrim = (int)(uni_u((uint32_t)NODE_RD(S, tk, p0)) & 1023u);
            const int m = uni_i((int)l_mi[(pTrim - 1) * n + (cTrim - 1)]);
            const int ncols = uni_i(l_pose[m].n_cols);
            if (lane < ncols) {
                const d2* ar = area_tab + (size_t)m * 3 * PDMPC_VMAX;
                const d2 a = ar[lane];                                                       // maneuver.area
                const d2 b = ar[((cK == Hp) ? 2 : 1) * PDMPC_VMAX + lane];                   // large offset at k == Hp, else without offset
                d2 sa, sb;
                sa.x = c * a.x - s * a.y + pX;  // GraphSearch.m:158
                sa.y = s * a.x + c * a.y + pY;  // :159
                sb.x = c * b.x - s * b.y + pX;  // :162 / :168
                sb.y = s * b.x + c * b.y + pY;  // :163 / :169
                l_shA[lane] = sa;
                l_shB[lane] = sb;
            }
            __syncthreads();
            const d2* soup_k = l_soup + l_soff[cK - 1];
            const int M_k = uni_i(l_soff[cK] - l_soff[cK - 1]);
            if (A.checker == PDMPC_CHECK_INTERX) {
                // are_constraints_satisfied_interx.m:17-37
                bool hit = interx_wave(l_shA, ncols, soup_k, M_k, lane);
                if (!hit) {
                    const int Hk = uni_i(l_hoff[cK] - l_hoff[cK - 1]);
                    if (Hk > 0) hit = interx_wave(l_shA, ncols, l_soup + l_hoff[cK - 1], Hk, lane);
                }
                if (!hit) hit = interx_wave(l_shB, ncols, l_soup + ll_base, ll_len, lane);
                valid = !hit;
            } else {
                // are_constraints_satisfied_sat.m:15-53
                bool hit = sat_soup_wave(l_shA, ncols, soup_k, M_k, lane);
                if (!hit) hit = sat_boundary_wave(l_shB, ncols, l_soup + ll_base, ll_len, lane);
                valid = !hit;
            }
            __syncthreads();
        }
        if (!valid) continue;  // GraphSearch.m:75-77

        if (cK == Hp) {  // :81-90
            goal = cur;
            break;
        }

        // ---- expand_node.m:1-91
        const double curX = uni_d(NODE_RD(S, x, c0));
        const double curY = uni_d(NODE_RD(S, y, c0));
        const double curYaw = uni_d(NODE_RD(S, yaw, c0));
        const double curG = uni_d(NODE_RD(S, g, c0));
        double sn, cs;
        pdmpc_sincos(curYaw, &sn, &cs);  // expand_node.m:50-51
        if (lane == 0) {
            if (c0 < S.NL) {
                S.lcs[c0] = cs;
                S.lsn[c0] = sn;
            } else {
                S.g.cs[c0] = cs;
                S.g.sn[c0] = sn;
            }
        }
        const int k_exp = cK + 1;                // :13
        const int steps_to_go = Hp - k_exp;      // :37
        const uint64_t* mrow = l_mask + ((size_t)(k_exp - 1) * n + (cTrim - 1)) * nw;
        uint32_t total = 0;
        for (int w = 0; w < nw; ++w) total += (uint32_t)__builtin_popcountll(mrow[w]);
        total = uni_u(total);
        if (nnodes + total > S.max_nodes) {
            status = PDMPC_ARENA_OVERFLOW;
            break;
        }
        for (int w = 0; w < nw; ++w) {
            uint64_t mask = mrow[w];
            {
                const uint32_t lo = uni_u((uint32_t)mask), hi = uni_u((uint32_t)(mask >> 32));
                mask = ((uint64_t)hi << 32) | lo;
            }
            const int cnt = __builtin_popcountll(mask);
            const bool active = (mask >> lane) & 1ull;
            const int rank = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
            double f = 0.0;
            if (active) {
                const int t2 = w * 64 + lane;  // 0-based successor trim
                const int m = (int)l_mi[(cTrim - 1) * n + t2];
                const DevManPose mp = l_pose[m];
                const double expX = cs * mp.dx - sn * mp.dy + curX;  // :53
                const double expY = sn * mp.dx + cs * mp.dy + curY;  // :54
                const double expYaw = curYaw + mp.dyaw;              // :55
                double expG = curG;
                {
                    const double ddx = expX - l_rx[k_exp - 1], ddy = expY - l_ry[k_exp - 1];
                    const double nrm = sqrt(ddx * ddx + ddy * ddy);
                    expG = expG + nrm * nrm;  // :61
                }
                double expH = 0.0, dmax = 0.0;
                for (int it = 1; it <= steps_to_go; ++it) {  // :68-73
                    dmax = dmax + l_dtv[k_exp + it - 1];
                    const double ddx = expX - l_rx[k_exp + it - 1], ddy = expY - l_ry[k_exp + it - 1];
                    const double nrm = sqrt(ddx * ddx + ddy * ddy);
                    const double df = nrm - dmax;
                    const double m0 = (df > 0) ? df : 0.0;
                    expH = expH + m0 * m0;
                }
                f = expG * 1 + expH * 1;  // GraphSearch.m:100-102
                const uint32_t i0 = nnodes + (uint32_t)rank;  // 0-based index of the child (Tree.add_nodes, Tree.m:61)
                const uint16_t tk = (uint16_t)((t2 + 1) | (k_exp << 10));
                S.g.x[i0] = expX;
                S.g.y[i0] = expY;
                S.g.yaw[i0] = expYaw;
                S.g.g[i0] = expG;
                S.g.h[i0] = expH;
                S.g.parent[i0] = cur;
                S.g.tk[i0] = tk;
                if (i0 < S.NL) {
                    S.lx[i0] = expX;
                    S.ly[i0] = expY;
                    S.lyaw[i0] = expYaw;
                    S.lg[i0] = expG;
                    S.lh[i0] = expH;
                    S.lparent[i0] = cur;
                    S.ltk[i0] = tk;
                }
            }
            if (nnodes + (uint32_t)cnt > S.NL) __threadfence_block();
            __syncthreads();
            // pq.push(new_open_nodes, new_open_values): one at a time in ascending trim order (mex.cpp:67-72)
            uint64_t mm = mask;
            uint32_t r = 0;
            while (mm) {
                const int l = __builtin_ctzll(mm);
                mm &= mm - 1;
                const double fk = lane_d(f, l);
                heap_push(S, nnodes + r + 1, fk);
                ++r;
            }
            nnodes += (uint32_t)cnt;
        }
    }

    // ---- results (GraphSearch.m:58-59, 82-89; return_path_to.m; return_path_area.m)
    __syncthreads();
    if (goal) {
        // path_to_root (Tree.m:44-52), reversed
        if (lane == 0) {
            uint32_t nd = goal;
            for (int i = Hp; i >= 0; --i) {
                l_path[i] = nd;
                nd = NODE_RD(S, parent, nd - 1);
            }
        }
        __syncthreads();
        if (lane <= Hp) {
            const uint32_t nd = l_path[lane];
            const uint32_t i0 = nd - 1;
            const uint32_t tk = (uint32_t)NODE_RD(S, tk, i0);
            const double x = NODE_RD(S, x, i0), y = NODE_RD(S, y, i0), yaw = NODE_RD(S, yaw, i0);
            O->tree_path[lane] = (int32_t)nd;
            double* row = O->path_nodes[lane];  // NodeInfo.m:5-13
            row[0] = x;
            row[1] = y;
            row[2] = yaw;
            row[3] = (double)(tk & 1023u);
            row[4] = NODE_RD(S, g, i0);
            row[5] = NODE_RD(S, h, i0);
            row[6] = (double)(tk >> 10);
            row[7] = 1.0;
            if (lane >= 1) {
                O->y_predicted[lane - 1][0] = x;
                O->y_predicted[lane - 1][1] = y;
                O->y_predicted[lane - 1][2] = yaw;
                O->predicted_trims[lane - 1] = (int32_t)(tk & 1023u);
            }
        }
        // shapes along the path: same arithmetic as at pop time (GraphSearch.m:158-160), so the same bits
        for (int idx = lane; idx < Hp * PDMPC_VMAX; idx += PDMPC_WAVE) {
            const int i = idx / PDMPC_VMAX + 1;
            const int v = idx - (i - 1) * PDMPC_VMAX;
            const uint32_t pn = l_path[i - 1] - 1, cn = l_path[i] - 1;
            const int pTrim = (int)((uint32_t)NODE_RD(S, tk, pn) & 1023u);
            const int cTrim = (int)((uint32_t)NODE_RD(S, tk, cn) & 1023u);
            const int m = (int)l_mi[(pTrim - 1) * n + (cTrim - 1)];
            const int ncols = l_pose[m].n_cols;
            if (v == 0) O->shape_cols[i - 1] = ncols;
            if (v < ncols) {
                const double c = NODE_RD(S, cs, pn), s = NODE_RD(S, sn, pn);
                const double pX = NODE_RD(S, x, pn), pY = NODE_RD(S, y, pn);
                const d2 a = area_tab[(size_t)m * 3 * PDMPC_VMAX + v];
                O->shapes[i - 1][0][v] = c * a.x - s * a.y + pX;
                O->shapes[i - 1][1][v] = s * a.x + c * a.y + pY;
            }
        }
    } else if (V->fb_off[0] >= 0) {
        // exhausted: publish the caller-supplied fallback areas so successors of this launch avoid them
        // (PrioritizedController.m:568-616, 678-718)
        for (int idx = lane; idx < Hp * PDMPC_VMAX; idx += PDMPC_WAVE) {
            const int k = idx / PDMPC_VMAX;
            const int v = idx - k * PDMPC_VMAX;
            const int a = V->fb_off[k], b = V->fb_off[k + 1];
            const int cols = (b - a < PDMPC_VMAX) ? (b - a) : PDMPC_VMAX;
            if (v == 0) O->shape_cols[k] = cols;
            if (v < cols) {
                O->shapes[k][0][v] = A.points[2 * (size_t)(a + v)];
                O->shapes[k][1][v] = A.points[2 * (size_t)(a + v) + 1];
            }
        }
    }
    if (lane == 0) {
        O->status = dep_timeout ? PDMPC_ERR_HIP : status;
        O->n_expanded = (int32_t)nnodes;
        O->n_popped = n_popped;
        O->n_hp = Hp;
        A.tree_size[slot] = (int32_t)nnodes;
    }
    // ---- publish: plain stores -> every wave's vmcnt(0) -> barrier -> lane-0 agent release -> flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(A.done_flag + slot, A.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

extern "C" int pdmpc_launch_search(const KernelArgs* args, int count, void* stream) {
    if (count <= 0) return 0;
    hipError_t e = hipFuncSetAttribute((const void*)pdmpc_search_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)args->lds.total);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pdmpc_search_kernel, dim3(count), dim3(PDMPC_WAVE), args->lds.total, (hipStream_t)stream, *args);
    return (int)hipGetLastError();
}
