// sampled_kernel.hip — the reference's sampled optimizer (OptimizerType.MatlabSampled), one wavefront per vehicle.
//
// Restates hlc/optimizer/graph_search/MonteCarloTreeSearch.m:40-249: up to 250 expansions of a tree that is grown by
// random descents from the root (the successor at every step is drawn with a precomputed mt19937ar number, :53,105);
// an edge is checked when its node is created (:170-179), a colliding edge is cut (:183), a node without successors
// is cut from its parent (:108-112); descents that reach the horizon collision-free are candidates, the cheapest one
// (cost = sum of squared distances to the reference points, :143) wins (:197).
//
// The descents are a dependent chain by construction (every descent reads the tree the previous ones left behind and
// the random numbers are consumed in order), so one wavefront runs it: the scalar bookkeeping is executed uniformly
// by all lanes, the edge check — the only data-parallel part — is the same wave-wide InterX / SAT code the optimal
// search uses (edge_checks.hpp).  Tables and the obstacle soups are staged in LDS exactly as in search_kernel.hip;
// the random numbers are generated on the host (api.cpp, mt19937ar + genrand_res53) and read from HBM.
// Floating point: the reference's expression order (3x3 transform times dpose, rows accumulated left to right),
// -ffp-contract=off, sin/cos from pdmpc_math.h: bit-identical to the oracle's restatement.
#include <hip/hip_runtime.h>

#include "../../include/pdmpc_math.h"
#include "pdmpc_device.h"

#define PROF_MEMBERS
#define PROF_STOP(i)

namespace {

#include "wave_primitives.hpp"
#include "search_state.hpp"
#include "edge_checks.hpp"

#define MCTS_EXPANSIONS_MAX 250 /* MonteCarloTreeSearch.m:8 */
#define MCTS_NODE_CAP 288       /* > 1 + 250 + PDMPC_HP_MAX: the last descent may overshoot the limit by Hp - 1 nodes */
#define MCTS_FANOUT 16          /* successors per (trim, step): pdmpc_upload_mpa admits at most 16 per mask word ... */

// the j-th (0-based) set bit of the successor mask of (trim, step), as a 0-based trim index; -1 if there is none
__device__ __forceinline__ int nth_successor(const lds_mask64* row, int nw, int j) {
    for (int w = 0; w < nw; ++w) {
        uint64_t m = row[w];
        const int c = __builtin_popcountll(m);
        if (j < c) {
            for (int q = 0; q < j; ++q) m &= m - 1;
            return w * 64 + __builtin_ctzll(m);
        }
        j -= c;
    }
    return -1;
}
__device__ __forceinline__ int successor_count(const lds_mask64* row, int nw) {
    int c = 0;
    for (int w = 0; w < nw; ++w) c += __builtin_popcountll(row[w]);
    return c;
}

}  // namespace

extern "C" __global__ __launch_bounds__(PDMPC_WAVE) void pdmpc_sampled_kernel(const KernelArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int slot = A.first + blockIdx.x;
    const int Hp = A.Hp, n = A.n_trims, nw = A.n_words;
    const DevVehicle* __restrict__ V = A.veh + slot;
    pdmpc_vehicle_out* __restrict__ O = A.out + slot;
    const double* __restrict__ random_numbers = A.sampled_random + (size_t)slot * A.sampled_n_random;

    // ---- LDS carve (the layout of the optimal search; its open-list region holds the tree arrays here)
    LDS_AS unsigned char* lsm = (LDS_AS unsigned char*)smem;
    lds_mask64* l_mask = (lds_mask64*)(lsm + A.lds.mask);
    lds_i16* l_mi = (lds_i16*)(lsm + A.lds.man_index);
    lds_pose* l_pose = (lds_pose*)(lsm + A.lds.pose);
    lds_f64* l_rx = (lds_f64*)(lsm + A.lds.ref);
    lds_f64* l_ry = l_rx + PDMPC_HP_MAX;
    lds_u32* l_path = (lds_u32*)(lsm + A.lds.path);
    lds_i32* l_soff = (lds_i32*)(l_path + PDMPC_HP_MAX + 2);
    lds_i32* l_hoff = l_soff + PDMPC_HP_MAX + 1;
    lds_d2* l_soup = (lds_d2*)(lsm + A.lds.soup);
    LDS_AS uint16_t* t_child = (LDS_AS uint16_t*)(lsm + A.lds.tree16);  // [MCTS_NODE_CAP][MCTS_FANOUT]: children(:, node)
    LDS_AS uint16_t* t_parent = t_child + MCTS_NODE_CAP * MCTS_FANOUT;   // [MCTS_NODE_CAP]
    LDS_AS uint16_t* t_trim = t_parent + MCTS_NODE_CAP;                  // [MCTS_NODE_CAP]

    CheckCtx C;
    C.l_area = (const lds_d2*)(lsm + A.lds.area);
    C.g_area = (const d2*)A.man_area;
    C.l_soup = l_soup;
    C.l_soff = l_soff;
    C.l_hoff = l_hoff;
    C.areas_in_lds = A.areas_in_lds;
    C.Hp = Hp;
    C.checker = A.checker;
    C.sh = (lds_d2*)(lsm + A.lds.shape);
    C.cand = (lds_u32*)(lsm + A.lds.cand);
    C.tally = (LDS_AS unsigned long long*)(C.sh + 2 * PDMPC_VMAX);
    if (lane == 0) {
        C.tally[0] = 0;
        C.tally[1] = 0;
    }

    // ---- prologue: tables, reference points, record defaults, obstacle soups (no predecessors: one computation level)
    {
        const int mask_bytes = Hp * n * nw * 8;
        stage16(l_mask, A.succ_mask, (mask_bytes + 15) / 16, lane);
        stage16(l_mi, A.man_index, (n * n * 2 + 15) / 16, lane);
        stage16(l_pose, A.man_pose, A.n_man * 2, lane);
        if (A.areas_in_lds) stage16(lsm + A.lds.area, A.man_area, A.n_man * 3 * PDMPC_VMAX, lane);
    }
    if (lane < Hp) {
        l_rx[lane] = V->ref_x[lane];
        l_ry[lane] = V->ref_y[lane];
    }
    {
        double* od = (double*)O;
        const int nd = (int)(sizeof(pdmpc_vehicle_out) / 8);
        const int y0 = (int)(offsetof(pdmpc_vehicle_out, y_predicted) / 8);
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        for (int i = lane; i < nd; i += PDMPC_WAVE) od[i] = (i >= y0 && i < y0 + PDMPC_HP_MAX * 3) ? qnan : 0.0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    {
        int off = 0;
        for (int k = 0; k < Hp; ++k) {
            const int a = V->lit_off[k], b = V->lit_off[k + 1];
            if (lane == 0) l_soff[k] = off;
            stage16(l_soup + off, (const d2*)A.points + a, b - a, lane);
            off += (b - a);
        }
        if (lane == 0) l_soff[Hp] = off;
        for (int k = 0; k < Hp; ++k) {
            const int a = V->hdv_off[k], b = V->hdv_off[k + 1];
            if (lane == 0) l_hoff[k] = off;
            stage16(l_soup + off, (const d2*)A.points + a, b - a, lane);
            off += (b - a);
        }
        if (lane == 0) l_hoff[Hp] = off;
        stage16(l_soup + off, (const d2*)A.points + V->ll_off, V->ll_len, lane);
        C.ll_base = off;
    }
    C.ll_len = uni_i(V->ll_len);
    for (int i = lane; i < MCTS_NODE_CAP * MCTS_FANOUT; i += PDMPC_WAVE) t_child[i] = 0;
    wave_sync();

    // ---- tree with root node (MonteCarloTreeSearch.m:60-74); node ids are 1-based, entry 0 is unused
    const double root_x = V->x0, root_y = V->y0, root_yaw = V->yaw0;
    if (lane == 0) {
        t_trim[1] = (uint16_t)V->trim0;
        t_parent[1] = 0;
    }
    {
        const int rc = successor_count(l_mask + ((size_t)0 * n + (V->trim0 - 1)) * nw, nw);
        if (lane < rc) t_child[1 * MCTS_FANOUT + lane] = 1;  // children(1:size(root_successor_trims, 2), 1) = 1
    }
    wave_sync();
    int n_nodes = 1, n_expansions = 0, n_traversals = 0;
    bool is_finished = false, have_best = false;
    double best_cost = 0.0;
    int best_node = 0;

    while (n_expansions < MCTS_EXPANSIONS_MAX && !is_finished && n_nodes + Hp < MCTS_NODE_CAP) {  // :89
        int node_id = 1;
        double solution_cost = 0.0;
        double px = root_x, py = root_y, pyaw = root_yaw;
        bool is_valid = false;
        int child_position = 0, node_parent = 0;
        for (int i_step = 1; i_step <= Hp; ++i_step) {  // :95
            is_valid = false;
            ++n_traversals;
            // trim_positions = find(children(:, node_id))                                              :100
            const uint32_t cv = lane < MCTS_FANOUT ? (uint32_t)t_child[node_id * MCTS_FANOUT + lane] : 0u;
            const unsigned long long nz = __ballot(cv != 0u);
            const int n_trims = __builtin_popcountll(nz);
            if (n_trims != 0) {
                // child_position = trim_positions(ceil(random_numbers(n_traversals) * n_trims))          :105
                const double r = uni_d(random_numbers[n_traversals - 1]);
                int pick = (int)ceil(r * (double)n_trims) - 1;
                if (pick < 0) pick = 0;  // (r == 0 would index element 0 in the reference: an error there)
                unsigned long long m = nz;
                for (int q = 0; q < pick; ++q) m &= m - 1;
                child_position = __builtin_ctzll(m);
            } else {
                if (node_id != 1) {  // remove edge to node without children                              :108-112
                    const int parent_id = (int)t_parent[node_id];
                    if (lane < MCTS_FANOUT && t_child[parent_id * MCTS_FANOUT + lane] == (uint16_t)node_id) t_child[parent_id * MCTS_FANOUT + lane] = 0;
                    wave_sync();
                    break;
                }
                is_finished = true;  // :114-115
                break;
            }
            // ---- expand node                                                                       :120-138
            const int parent_trim = (int)t_trim[node_id];
            const int goal_trim = nth_successor(l_mask + ((size_t)(i_step - 1) * n + (parent_trim - 1)) * nw, nw, child_position) + 1;
            const int m = (int)l_mi[(parent_trim - 1) * n + (goal_trim - 1)];
            const double dx = l_pose[m].dx, dy = l_pose[m].dy, dyaw = l_pose[m].dyaw;
            const int ncols = l_pose[m].n_cols;
            double s, c;
            pdmpc_sincos(pyaw, &s, &c);
            const double sx0 = px, sy0 = py;  // start_pose
            {
                // node_pose = node_pose + transform * maneuver.dpose: rows accumulated left to right, zeros included
                const double t0 = c * dx + (-s) * dy + 0.0 * dyaw;
                const double t1 = s * dx + c * dy + 0.0 * dyaw;
                const double t2 = 0.0 * dx + 0.0 * dy + 1.0 * dyaw;
                px = px + t0;
                py = py + t1;
                pyaw = pyaw + t2;
            }
            {
                // solution_cost += norm(node_pose(1:2) - reference_trajectory_points(:, i_step))^2          :143
                const double ddx = px - l_rx[i_step - 1], ddy = py - l_ry[i_step - 1];
                const double nrm = sqrt(ddx * ddx + ddy * ddy);
                solution_cost = solution_cost + nrm * nrm;
            }
            const int cval = (int)t_child[node_id * MCTS_FANOUT + child_position];
            if (cval != 1) {  // is_expanded                                                              :145-150
                node_id = cval;
                continue;
            }
            ++n_expansions;  // :152
            node_parent = node_id;
            // shapes = transform(1:2,1:2) * area + start_pose(1:2)                                        :159-166
            if (lane < ncols) {
                const size_t ai = (size_t)m * 3 * PDMPC_VMAX + lane;
                const size_t bi = ai + (size_t)((i_step == Hp) ? 2 : 1) * PDMPC_VMAX;
                d2 a, b;
                if (C.areas_in_lds) {
                    a = C.l_area[ai];
                    b = C.l_area[bi];
                } else {
                    a = C.g_area[ai];
                    b = C.g_area[bi];
                }
                d2 sa, sb;
                sa.x = (c * a.x + (-s) * a.y) + sx0;
                sa.y = (s * a.x + c * a.y) + sy0;
                sb.x = (c * b.x + (-s) * b.y) + sx0;
                sb.y = (s * b.x + c * b.y) + sy0;
                C.sh[lane] = sa;
                C.sh[PDMPC_VMAX + lane] = sb;
            }
            wave_sync();
            {
                // are_constraints_satisfied(iter, iVeh, shapes, shapes_for_boundary_check, i_step, ...)      :170-179
                const int so = uni_i(C.l_soff[i_step - 1]);
                const int M_k = uni_i(C.l_soff[i_step]) - so;
                bool hit;
                if (lane == 0) C.tally[0] += 1;
                if (A.checker == PDMPC_CHECK_INTERX) {
                    const int ho = uni_i(C.l_hoff[i_step - 1]);
                    const int Hk = uni_i(C.l_hoff[i_step]) - ho;
                    if (lane == 0) C.tally[1] += (unsigned long long)(ncols - 1) * (unsigned long long)((M_k > 1 ? M_k - 1 : 0) + (Hk > 1 ? Hk - 1 : 0) + (C.ll_len > 1 ? C.ll_len - 1 : 0));
                    hit = interx_check(C.sh, ncols, C.l_soup, so, M_k, ho, Hk, C.ll_base, C.ll_len, lane);
                } else {
                    hit = sat_soup_wave(C.sh, ncols, C.l_soup + so, M_k, lane);
                    if (!hit) hit = sat_boundary_wave(C.sh + PDMPC_VMAX, ncols, C.l_soup + C.ll_base, C.ll_len, lane);
                }
                wave_sync();
                is_valid = !hit;
            }
            if (!is_valid) {
                if (lane == 0) t_child[node_parent * MCTS_FANOUT + child_position] = 0;  // remove edge     :183
                wave_sync();
                break;
            }
            ++n_nodes;  // add node                                                                         :186-193
            {
                const int cc = (i_step != Hp) ? successor_count(l_mask + ((size_t)i_step * n + (goal_trim - 1)) * nw, nw) : 0;
                if (lane < MCTS_FANOUT) t_child[n_nodes * MCTS_FANOUT + lane] = lane < cc ? (uint16_t)1 : (uint16_t)0;
                if (lane == 0) {
                    t_parent[n_nodes] = (uint16_t)node_parent;
                    t_trim[n_nodes] = (uint16_t)goal_trim;
                    t_child[node_parent * MCTS_FANOUT + child_position] = (uint16_t)n_nodes;
                }
            }
            wave_sync();
            node_id = n_nodes;
        }
        if (is_valid) {  // valid_nodes_at_hp.push(node_id, solution_cost); only the queue's top is ever read         :199-203
            if (!have_best || solution_cost < best_cost) {
                have_best = true;
                best_cost = solution_cost;
                best_node = node_id;
            }
            if (lane == 0) t_child[node_parent * MCTS_FANOUT + child_position] = 0;  // avoid double exploration
            wave_sync();
        }
    }

    // ---- results                                                                                       :207-248
    if (lane == 0) {
        O->n_expanded = n_expansions;
        O->n_popped = n_traversals;
        O->n_hp = Hp;
        A.tree_size[slot] = n_nodes;
    }
    if (have_best) {
        if (lane == 0) {
            int nd = best_node;
            for (int i = Hp; i >= 0; --i) {
                l_path[i] = (uint32_t)nd;
                nd = (int)t_parent[nd];
            }
        }
        wave_sync();
        // poses along the path, recomputed like the reference does (:228-239), then the swept areas with the descent's
        // arithmetic (the reference kept them from the descent that created each node: same inputs, same bits)
        double px = root_x, py = root_y, pyaw = root_yaw;
        for (int i = 0; i <= Hp; ++i) {
            const int nd = (int)l_path[i];
            if (i >= 1) {
                const int t1 = (int)t_trim[l_path[i - 1]], t2 = (int)t_trim[nd];
                const int m = (int)l_mi[(t1 - 1) * n + (t2 - 1)];
                const double dx = l_pose[m].dx, dy = l_pose[m].dy, dyaw = l_pose[m].dyaw;
                const int ncols = l_pose[m].n_cols;
                double s, c;
                pdmpc_sincos(pyaw, &s, &c);
                if (lane < ncols) {
                    const d2 a = C.g_area[(size_t)m * 3 * PDMPC_VMAX + lane];
                    O->shapes[i - 1][0][lane] = (c * a.x + (-s) * a.y) + px;
                    O->shapes[i - 1][1][lane] = (s * a.x + c * a.y) + py;
                }
                if (lane == 0) O->shape_cols[i - 1] = ncols;
                const double t0 = c * dx + (-s) * dy + 0.0 * dyaw;
                const double u1 = s * dx + c * dy + 0.0 * dyaw;
                const double u2 = 0.0 * dx + 0.0 * dy + 1.0 * dyaw;
                px = px + t0;
                py = py + u1;
                pyaw = pyaw + u2;
            }
            if (lane == 0) {
                O->tree_path[i] = nd;
                double* row = O->path_nodes[i];  // NodeInfo order; g/h stay -1 except g of the chosen node (:223-225, 244)
                row[0] = px;
                row[1] = py;
                row[2] = pyaw;
                row[3] = (double)t_trim[nd];
                row[4] = (nd == best_node) ? best_cost : -1.0;
                row[5] = -1.0;
                row[6] = (double)(i + 1);
                row[7] = 1.0;
                if (i >= 1) {
                    O->y_predicted[i - 1][0] = px;
                    O->y_predicted[i - 1][1] = py;
                    O->y_predicted[i - 1][2] = pyaw;
                    O->predicted_trims[i - 1] = (int32_t)t_trim[nd];
                }
            }
        }
        if (lane == 0) O->status = PDMPC_OK;
    } else if (lane == 0) {
        O->status = PDMPC_EXHAUSTED;  // :212-215
    }
    if (lane == 0) {
        atomicAdd(A.work_count + 0, C.tally[0]);
        atomicAdd(A.work_count + 1, C.tally[1]);
    }
    // publish like the optimal search (a later launch of the same step may wait on the flag)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_sync();
    if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(A.done_flag + slot, A.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

extern "C" int pdmpc_launch_sampled(const KernelArgs* args, int count, void* stream) {
    if (count <= 0) return 0;
    hipError_t e = hipFuncSetAttribute((const void*)pdmpc_sampled_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)args->lds.total);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pdmpc_sampled_kernel, dim3(count), dim3(PDMPC_WAVE), args->lds.total, (hipStream_t)stream, *args);
    return (int)hipGetLastError();
}
