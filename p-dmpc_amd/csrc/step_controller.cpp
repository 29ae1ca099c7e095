// step_controller.cpp — the caller's side of the optimizer boundary, natively: one MPC time step of the prioritized
// sequential controller around pdmpc_plan_step.  Host code only (no device code in this file).
//
// What it restates (file:line relative to the reference root), the C++ twin of p-dmpc_amd/pdmpc/controller.py:
//   traffic info per step     HighLevelController.update_controlled_vehicles_traffic_info (hlc/controller/HighLevelController.m:167-270)
//   trim from measurement     MotionPrimitiveAutomaton.trim_from_values (hlc/model/motion_primitive_automaton/MotionPrimitiveAutomaton.m:193-236)
//   occupied areas            hlc/controller/common/get_occupied_areas.m:21-31, utility/translate_global.m:19-22
//   reference trajectory      hlc/controller/common/get_reference_trajectory.m:27-46, sample_reference_trajectory.m:1-99,
//                             get_arc_distance_to_endpoint.m:39-114, projection_2d.m:14-42
//   predicted lanelets        hlc/controller/common/get_predicted_lanelets.m:25-62, get_lanelets_boundary.m:18-68
//   coupling                  Coupler.m:31-32 (full), DistanceCoupler.m:15-50 (distance)
//   priorities -> DAG         ConstantPrioritizer.m:14-20, Prioritizer.m:36-77, ColoringPrioritizer.m:11-131
//   grouping                  PrioritizedController.group (hlc/controller/prioritized/PrioritizedController.m:375-389),
//                             weight/DistanceWeigher.m:12-39, weight/ConstantWeigher.m:15-17, cut/GreedyCutter.m:5-86
//   computation levels        utility/kahn.m:1-24
//   obstacle assembly         PrioritizedController.plan / consider_predecessors / consider_successors (:297-324, 449-566)
//   exhaustion, fallbacks     handle_graph_search_exhaustion / plan_fallback (:568-616, 678-718), check_others_fallback (:623-676),
//                             HighLevelController.handle_others_fallback (HighLevelController.m:449-463)
//   plant                     Simulation.apply (plant/Simulation.m:86-100)
// Every floating-point expression keeps the order of the Python twin (which keeps the reference's), and both call the
// same libm, so the step problems the two build are bit-identical (tests/test_native_controller.py).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <utility>
#include <memory>
#include <vector>

#include "../../include/pdmpc.h"
#include "mt19937ar.hpp"

namespace {

struct Poly {  // 2 x V, MATLAB [x; y]
    std::vector<double> x, y;
    int n() const { return (int)x.size(); }
};

// the non-zero entries of a matrix as lists: by row (idx[off[i] .. off[i + 1]) = the columns of row i) or by column (the rows of
// column j), ascending in both forms
struct Lists {
    std::vector<int32_t> off, idx, fill;  // (fill: scratch of lists_by_column)
    int size(int i) const { return off[i + 1] - off[i]; }
    const int32_t* begin(int i) const { return idx.data() + off[i]; }
    const int32_t* end(int i) const { return idx.data() + off[i + 1]; }
};

struct Plan {  // what the controller keeps of a vehicle's ControlResultsInfo (ControlResultsInfo.m:5-17)
    bool present = false;
    bool needs_fallback = false;
    bool exhausted = false;
    std::vector<Poly> shapes;         // Hp
    std::vector<int32_t> trims;       // Hp
    std::vector<double> yx, yy, yyaw; // Hp
    int32_t n_expanded = 0;
};

struct VehicleDef {
    double x_start, y_start, yaw_start, reference_speed;
    std::vector<double> px, py;             // reference path
    std::vector<int32_t> lanelets_index;    // 1-based lanelet ids along the loop (empty: no lanelets, circle scenario)
    std::vector<int32_t> points_index;      // 1-based index of the last path point of each of those lanelets
    bool is_loop;
    double tile_dx, tile_dy;
};

}  // namespace

struct pdmpc_controller {
    pdmpc_handle* h = nullptr;
    pdmpc_controller_config cfg{};
    int n = 0, Hp = 0;
    std::vector<VehicleDef> veh;
    std::vector<Poly> bl_left, bl_right;  // per lanelet boundary polylines (RoadDataCommonRoad.get_lanelet_boundary)
    std::vector<Poly> static_obstacles;
    std::vector<double> trim_speed, trim_steering;
    // state
    int k = 0;
    std::vector<double> mx, my, myaw, mspeed, msteer;  // measurements
    std::vector<Plan> info_old, infos;
    bool follow_own = false;                // the explorative step applies the plans of the controller's OWN prioritization (instance 0) whatever the choice: the traffic then follows pdmpc_controller_step's closed loop (measurement: the same steps as a recorded replay)
    bool lean_explore = false;              // the explorative step reads back status + final cost of every plan and the chosen plans' records only
    std::vector<int32_t> x_status;
    std::vector<double> x_final_cost;
    double timing[6] = {0, 0, 0, 0, 0, 0};  // pdmpc_controller_last_timing
    double timing_sum[6] = {0, 0, 0, 0, 0, 0};  // ... summed over the steps since the last pdmpc_controller_timing_sum(reset)
    int64_t timing_steps = 0;
    std::vector<double> last_pops;  // per vehicle: nodes its search popped in the last step (the next step's expected work, pdmpc_set_step_weights)
    // per step
    std::vector<int32_t> trims;
    std::vector<Poly> occ_offset, occ_plain;
    std::vector<std::vector<double>> ref_x, ref_y, v_ref;
    std::vector<Poly> bnd_left, bnd_right;
    std::vector<uint8_t> adjacency, directed, directed_seq;  // n x n row-major
    std::vector<int32_t> levels, order, slot_of;
    // the step problem in the C ABI's form (what pdmpc_plan_step takes); the arena keeps the pointed-to data alive
    std::vector<pdmpc_vehicle_in> in;
    std::vector<pdmpc_polygon_set> fb;
    std::vector<int32_t> pred_offset, pred_index;
    // the arrays of the step's polygon sets: chunks that are kept from step to step and handed out front to back (a set's arrays
    // never move; build_step starts over at the first chunk)
    struct Arena {
        struct Chunk {
            std::unique_ptr<double[]> mem;  // (doubles: 8-byte alignment for both kinds of arrays)
            size_t cap = 0;
        };
        std::vector<Chunk> chunks;
        size_t cur = 0, used = 0;
        void reset() { cur = used = 0; }
        void* take(size_t bytes) {
            const size_t need = (bytes + 7) / 8;
            while (cur < chunks.size() && used + need > chunks[cur].cap) {
                ++cur;
                used = 0;
            }
            if (cur == chunks.size()) {
                Chunk ch;
                ch.cap = std::max(need, (size_t)1 << 17);
                ch.mem.reset(new double[ch.cap]);
                chunks.push_back(std::move(ch));
                used = 0;
            }
            void* p = chunks[cur].mem.get() + used;
            used += need;
            return p;
        }
    } arena;
    std::vector<int32_t> sb_off;  // SetBuilder's scratch (one builder at a time)
    std::vector<double> sb_x, sb_y;
    std::vector<pdmpc_vehicle_out> out;
    // sets that do not depend on the prioritization are built once per time step and shared by the prioritizations of an explorative step
    std::vector<pdmpc_polygon_set> fb_of;
    std::vector<uint8_t> fb_done;
    // obstacle sets of a vehicle by who contributes to them (a function of the vehicle and of those lists alone): the prioritizations
    // of an explorative step differ in a few couplings, so most of their vehicles share their sets — one build, one pointer, and
    // pdmpc_pack_step packs a set it has seen under the same pointer once (api.cpp: pack_common)
    struct MemoKey {  // who contributes, as bit masks over the vehicles (up to 512: larger scenarios build every set)
        uint64_t w[16];
    };
    struct Memo {  // a vehicle's sets built so far this step, by key (a handful: searched front to back)
        std::vector<MemoKey> keys;
        std::vector<pdmpc_polygon_set> sets;
        const pdmpc_polygon_set* find(const MemoKey& k) const {
            for (size_t q = 0; q < keys.size(); ++q)
                if (std::memcmp(keys[q].w, k.w, sizeof k.w) == 0) return &sets[q];
            return nullptr;
        }
        void clear() {
            keys.clear();
            sets.clear();
        }
    };
    std::vector<Memo> obst_memo, dyn_memo;
    Lists ls_dir_succ, ls_dir_pred, ls_seq_succ, ls_seq_pred;  // assemble_step's scratch (kept: no allocation per prioritization)
    std::vector<int> kahn_indeg, kahn_cur, kahn_next;
    pdmpc_polygon_set empty_set{};
    bool empty_done = false;
    bool exploring = false;  // an explorative step is being built: its prioritizations share sets through the memos
    // explorative step (PrioritizedExplorativeController): the prioritizations of the current traffic state, flattened
    struct Instance {
        std::vector<uint8_t> directed, directed_seq;
        std::vector<int32_t> levels, order, slot_of;
    };
    std::vector<Instance> inst;
    struct Part {  // an instance's step problem as assemble_step left it (explore_build's scratch, kept from step to step)
        std::vector<pdmpc_vehicle_in> in;
        std::vector<pdmpc_polygon_set> fb;
        std::vector<int32_t> pred_offset, pred_index;
    };
    std::vector<Part> x_parts;
    std::vector<pdmpc_vehicle_in> x_in;
    std::vector<pdmpc_polygon_set> x_fb;
    std::vector<int32_t> x_pred_offset, x_pred_index, x_instance, x_vehicle, x_level, x_slot;  // x_slot[p * n + vehicle] = slot in the flattened batch
    std::vector<pdmpc_vehicle_out> x_out;
    std::vector<int32_t> x_chosen;  // per vehicle: the instance its sub-graph chose
    std::vector<double> x_cost;     // n_perm x n_graphs
    int x_graphs = 0;
    std::string err;
};

namespace {

thread_local std::string g_cerr;

inline uint8_t& at(std::vector<uint8_t>& m, int n, int i, int j) { return m[(size_t)i * n + j]; }
inline uint8_t at(const std::vector<uint8_t>& m, int n, int i, int j) { return m[(size_t)i * n + j]; }

// f(j) for the non-zero entries j of a matrix row, ascending.  Rows of the coupling matrices are mostly zero (a vehicle is coupled with
// the few around it): eight entries per test.
template <class F>
inline void for_each_set(const uint8_t* row, int n, F&& f) {
    int j = 0;
    for (; j + 8 <= n; j += 8) {
        uint64_t w;
        std::memcpy(&w, row + j, 8);
        if (w == 0) continue;
        for (int q = 0; q < 8; ++q)
            if (row[j + q]) f(j + q);
    }
    for (; j < n; ++j)
        if (row[j]) f(j);
}

void lists_by_row(const std::vector<uint8_t>& M, int n, Lists& L) {
    L.off.assign((size_t)n + 1, 0);
    L.idx.clear();
    for (int i = 0; i < n; ++i) {
        for_each_set(M.data() + (size_t)i * n, n, [&](int j) { L.idx.push_back(j); });
        L.off[i + 1] = (int32_t)L.idx.size();
    }
}
void lists_by_column(const std::vector<uint8_t>& M, int n, const Lists& by_row, Lists& L) {
    (void)M;
    L.off.assign((size_t)n + 1, 0);
    for (int32_t j : by_row.idx) ++L.off[j + 1];
    for (int j = 0; j < n; ++j) L.off[j + 1] += L.off[j];
    L.idx.resize(by_row.idx.size());
    L.fill.assign(L.off.begin(), L.off.end() - 1);
    for (int i = 0; i < n; ++i)
        for (const int32_t* q = by_row.begin(i); q != by_row.end(i); ++q) L.idx[L.fill[*q]++] = i;
}

// utility/kahn.m:1-24: computation level (1-based) of every vertex of the DAG A (A[i][j] = 1: i before j)
bool kahn(const std::vector<uint8_t>& A, int n, std::vector<int32_t>& L) {
    // level = 1 + the longest path from a source (what removing all current sources, level by level, assigns); in-degrees are
    // counted once and decremented along the removed vertices' rows
    L.assign(n, 0);
    std::vector<int> indeg(n, 0), cur, next;
    for (int i = 0; i < n; ++i) for_each_set(A.data() + (size_t)i * n, n, [&](int j) { ++indeg[j]; });
    for (int j = 0; j < n; ++j)
        if (indeg[j] == 0) cur.push_back(j);
    int n_done = 0, level = 1;
    while (n_done < n) {
        if (cur.empty()) return false;  // a cycle
        next.clear();
        for (int v : cur) {
            L[v] = level;
            ++n_done;
        }
        for (int v : cur)
            for_each_set(A.data() + (size_t)v * n, n, [&](int j) {
                if (--indeg[j] == 0) next.push_back(j);
            });
        cur.swap(next);
        ++level;
    }
    return true;
}

// kahn over the successor lists of the matrix
bool kahn_lists(const Lists& succ, int n, std::vector<int32_t>& L, std::vector<int>& indeg, std::vector<int>& cur, std::vector<int>& next) {
    L.assign(n, 0);
    indeg.assign(n, 0);
    cur.clear();
    for (int32_t j : succ.idx) ++indeg[j];
    for (int j = 0; j < n; ++j)
        if (indeg[j] == 0) cur.push_back(j);
    int n_done = 0, level = 1;
    while (n_done < n) {
        if (cur.empty()) return false;  // a cycle
        next.clear();
        for (int v : cur) {
            L[v] = level;
            ++n_done;
        }
        for (int v : cur)
            for (const int32_t* q = succ.begin(v); q != succ.end(v); ++q)
                if (--indeg[*q] == 0) next.push_back(*q);
        cur.swap(next);
        ++level;
    }
    return true;
}

// MotionPrimitiveAutomaton.trim_from_values (:193-236): 1-based index of the closest trim
int trim_from_values(const pdmpc_controller& c, double speed, double steering) {
    const int nt = (int)c.trim_speed.size();
    if (steering == 0) {
        int best = -1;
        double bd = 0;
        for (int t = 0; t < nt; ++t) {
            if (c.trim_steering[t] != 0) continue;
            const double d = std::fabs(c.trim_speed[t] - speed);
            if (best < 0 || d < bd) {
                best = t;
                bd = d;
            }
        }
        return best + 1;
    }
    double sp_min = c.trim_speed[0], sp_max = c.trim_speed[0], st_min = c.trim_steering[0], st_max = c.trim_steering[0];
    for (int t = 1; t < nt; ++t) {
        sp_min = std::min(sp_min, c.trim_speed[t]);
        sp_max = std::max(sp_max, c.trim_speed[t]);
        st_min = std::min(st_min, c.trim_steering[t]);
        st_max = std::max(st_max, c.trim_steering[t]);
    }
    const double sp_s = sp_max - sp_min, st_s = st_max - st_min;
    int best = 0;
    double bd = 0;
    for (int t = 0; t < nt; ++t) {
        const double a = (c.trim_speed[t] - sp_min) / sp_s - (speed - sp_min) / sp_s;
        const double b = (c.trim_steering[t] - st_min) / st_s - (steering - st_min) / st_s;
        const double d = std::hypot(a, b);
        if (t == 0 || d < bd) {
            best = t;
            bd = d;
        }
    }
    return best + 1;
}

// get_occupied_areas.m:21-31 -> closed rectangles with and without the offset (translate_global.m:19-22)
void occupied_areas(double x, double y, double yaw, double length, double width, double offset, Poly& with_offset, Poly& plain) {
    static const double sx[5] = {-1, -1, 1, 1, -1}, sy[5] = {-1, 1, 1, -1, -1};
    const double c = std::cos(yaw), s = std::sin(yaw);
    with_offset.x.resize(5);
    with_offset.y.resize(5);
    plain.x.resize(5);
    plain.y.resize(5);
    for (int q = 0; q < 5; ++q) {
        const double xa = sx[q] * (length / 2 + offset), ya = sy[q] * (width / 2 + offset);
        with_offset.x[q] = c * xa + (-s) * ya + x;
        with_offset.y[q] = s * xa + c * ya + y;
        const double xb = sx[q] * (length / 2), yb = sy[q] * (width / 2);
        plain.x[q] = c * xb + (-s) * yb + x;
        plain.y[q] = s * xb + c * yb + y;
    }
}

inline double norm2(double a, double b) { return std::sqrt(a * a + b * b); }

// projection_2d.m:14-42 -> projected point and lambda
void projection_2d(double x1, double y1, double x2, double y2, double x3, double y3, double& xp, double& yp, double& lambda) {
    const double b = std::sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1));
    if (b != 0) {
        const double xn = (x2 - x1) / b, yn = (y2 - y1) / b;
        const double x31 = x3 - x1, y31 = y3 - y1;
        const double dot = xn * x31 + yn * y31;
        xp = x1 + dot * xn;
        yp = y1 + dot * yn;
        lambda = dot / b;
    } else {
        xp = x1;
        yp = y1;
        lambda = 0.0;
    }
}

// get_arc_distance_to_endpoint.m:39-114 (the part the sampler uses): projected point and 1-based idx_next
void arc_projection(double px, double py, const std::vector<double>& cx, const std::vector<double>& cy, double& xp, double& yp, int& idx_next) {
    const int np = (int)cx.size();
    int ic = 0;
    double best = 0;
    auto sq_of = [&](int i) { return (cx[i] - px) * (cx[i] - px) + (cy[i] - py) * (cy[i] - py); };
    for (int i = 0; i < np; ++i) {
        const double d = sq_of(i);
        if (i == 0 || d < best) {
            best = d;
            ic = i;
        }
    }
    int f, s;
    if (ic == 0) {
        f = 0;
        s = 1;
    } else if (ic == np - 1) {
        f = np - 2;
        s = np - 1;
    } else if (sq_of(ic - 1) <= sq_of(ic + 1)) {
        f = ic - 1;
        s = ic;
    } else {
        f = ic;
        s = ic + 1;
    }
    double lam;
    projection_2d(cx[f], cy[f], cx[s], cy[s], px, py, xp, yp, lam);
    const int idx_closest = ic + 1;
    idx_next = idx_closest;
    if ((0 <= lam && lam <= 0.5) || lam >= 1) idx_next = idx_closest < np ? idx_closest + 1 : 1;
    idx_next = std::max(2, idx_next);
}

// sample_reference_trajectory.m:1-99 (indices 1-based)
void sample_reference(int n_samples, const std::vector<double>& rx, const std::vector<double>& ry, double x_cur, double y_cur, const std::vector<double>& step,
                      std::vector<double>& out_x, std::vector<double>& out_y, std::vector<int32_t>& points_index, int& current_point_index) {
    out_x.assign(n_samples, 0.0);
    out_y.assign(n_samples, 0.0);
    points_index.assign(n_samples, 0);
    double cx, cy;
    int point_index;
    arc_projection(x_cur, y_cur, rx, ry, cx, cy, point_index);
    current_point_index = point_index;
    const int n_line = (int)rx.size();
    const bool is_loop = norm2(rx[0] - rx[n_line - 1], ry[0] - ry[n_line - 1]) < 1e-8;
    bool at_end = point_index == n_line;
    int last = point_index - 1;
    if (is_loop && at_end) point_index = 1;
    auto X = [&](int i) { return rx[i - 1]; };
    auto Y = [&](int i) { return ry[i - 1]; };
    for (int i = 0; i < n_samples; ++i) {
        double remaining = norm2(cx - X(point_index), cy - Y(point_index));
        if (remaining > step[i] || point_index == n_line) {
            while (X(point_index) == X(last) && Y(point_index) == Y(last) && last > 1) --last;
            const double dx = X(point_index) - X(last), dy = Y(point_index) - Y(last);
            const double nn = norm2(dx, dy);
            cx = cx + step[i] * (dx / nn);
            cy = cy + step[i] * (dy / nn);
        } else {
            double reflength = remaining;
            while (remaining < step[i]) {
                reflength = remaining;
                cx = X(point_index);
                cy = Y(point_index);
                last = point_index;
                point_index = std::min(point_index + 1, n_line);
                at_end = point_index == n_line;
                if (is_loop && at_end) point_index = 1;
                remaining = remaining + norm2(cx - X(point_index), cy - Y(point_index));
            }
            const double dx = X(point_index) - X(last), dy = Y(point_index) - Y(last);
            const double nn = norm2(dx, dy);
            cx = cx + (step[i] - reflength) * (dx / nn);
            cy = cy + (step[i] - reflength) * (dy / nn);
        }
        out_x[i] = cx;
        out_y[i] = cy;
        points_index[i] = point_index;
    }
}

// get_predicted_lanelets.m:25-62 + get_lanelets_boundary.m:18-68 for vehicle v
void lanelet_boundary(const pdmpc_controller& c, int v, const std::vector<int32_t>& ref_points_index, int current_point_index, Poly& left, Poly& right) {
    const VehicleDef& V = c.veh[v];
    left.x.clear();
    left.y.clear();
    right.x.clear();
    right.y.clear();
    if (V.lanelets_index.empty()) return;
    const int n_total = (int)V.px.size(), n_lan = (int)V.lanelets_index.size();
    int rpi[PDMPC_HP_MAX + 1], seen[PDMPC_HP_MAX + 2], predicted[PDMPC_HP_MAX + 2];
    int n_rpi = 0, n_seen = 0, n_pred = 0;
    for (int32_t p : ref_points_index) rpi[n_rpi++] = p;
    int index_add = rpi[n_rpi - 1] + 4;
    if (index_add > n_total) index_add -= n_total;
    rpi[n_rpi++] = index_add;
    for (int t = 0; t < n_rpi; ++t) {
        const int p = rpi[t];
        int q = 1;
        for (int u = 0; u < n_lan; ++u) q += p > V.points_index[u];
        if (std::find(seen, seen + n_seen, q) == seen + n_seen) seen[n_seen++] = q;  // unique(..., 'stable')
    }
    if (n_seen == 1) {
        int nxt = seen[0] + 1;
        if (nxt > n_lan) nxt = 1;
        seen[n_seen++] = nxt;
    }
    (void)current_point_index;
    for (int t = 0; t < n_seen; ++t) predicted[n_pred++] = V.lanelets_index[std::min(seen[t], n_lan) - 1];
    auto append = [](Poly& dst, const Poly& src, int from, int to) {
        dst.x.insert(dst.x.end(), src.x.begin() + from, src.x.begin() + to);
        dst.y.insert(dst.y.end(), src.y.begin() + from, src.y.begin() + to);
    };
    // up to four points of the predecessor lanelet in front   :39-65
    int pos = (int)(std::find(V.lanelets_index.begin(), V.lanelets_index.end(), predicted[0]) - V.lanelets_index.begin());
    int pred = -1;
    if (pos != 0)
        pred = V.lanelets_index[pos - 1];
    else if (V.is_loop)
        pred = V.lanelets_index.back();
    if (pred >= 0) {
        const Poly& pl = c.bl_left[pred - 1];
        const Poly& pr = c.bl_right[pred - 1];
        const int num_added = std::min(4, std::min(pr.n() - 1, pl.n() - 1));
        append(left, pl, pl.n() - 1 - num_added, pl.n() - 1);
        append(right, pr, pr.n() - 1 - num_added, pr.n() - 1);
    }
    // then the boundaries of the predicted lanelets back to back, each without its last point but the final one   :26-32
    for (int q = 0; q < n_pred; ++q) {
        const Poly& bl = c.bl_left[predicted[q] - 1];
        const Poly& br = c.bl_right[predicted[q] - 1];
        const bool final_one = q + 1 == n_pred;
        append(left, bl, 0, final_one ? bl.n() : bl.n() - 1);
        append(right, br, 0, final_one ? br.n() : br.n() - 1);
    }
    for (int i = 0; i < left.n(); ++i) {
        left.x[i] = left.x[i] + V.tile_dx;
        left.y[i] = left.y[i] + V.tile_dy;
    }
    for (int i = 0; i < right.n(); ++i) {
        right.x[i] = right.x[i] + V.tile_dx;
        right.y[i] = right.y[i] + V.tile_dy;
    }
}

// ColoringPrioritizer.prioritize (:11-27): directed coupling from a colouring of the undirected graph
void coloring_directed(const std::vector<uint8_t>& adjacency, int n, std::vector<uint8_t>& directed) {
    // neighbour lists of the graph without self-loops; the degrees the selection compares are the matrix's column sums (:38-45)
    Lists nb;
    nb.off.assign((size_t)n + 1, 0);
    nb.idx.clear();
    std::vector<int> degree(n, 0), color(n, 0);
    std::vector<long> deg(n, 0);  // column counts of the matrix as given (order_topo, :93)
    for (int i = 0; i < n; ++i) {
        for_each_set(adjacency.data() + (size_t)i * n, n, [&](int j) {
            ++deg[j];
            if (j == i) return;
            nb.idx.push_back(j);
            degree[j] += adjacency[(size_t)i * n + j];
        });
        nb.off[i + 1] = (int32_t)nb.idx.size();
    }
    for (int i = 0; i < n; ++i)
        if (degree[i] == 0) color[i] = 1;  // :45
    // per vertex the distinct colours its neighbours carry, as a bit set (kept up to date as vertices are coloured: the selection
    // below is then a scan of the vertices, not of the matrix — 512 vehicles: 63 ms -> well under 1 ms per step)
    const int cw = (n + 2 + 63) / 64;
    std::vector<int> ncol(n, 0);                   // distinct colours among the coloured neighbours
    std::vector<uint64_t> has((size_t)n * cw, 0);  // bit c of has[i]: a neighbour of i carries colour c
    auto mark = [&](int j, int col) {
        uint64_t& w = has[(size_t)j * cw + (col >> 6)];
        const uint64_t bit = 1ull << (col & 63);
        if (!(w & bit)) {
            w |= bit;
            ++ncol[j];
        }
    };
    // vertex_sdo_ldo (:65-89) scans the uncoloured vertices for the most distinct neighbour colours and, among equals, moves on to
    // a vertex only if its degree is strictly larger than the current pick's: the pick is the first uncoloured vertex with the
    // largest (colours, degree) pair.  key = that pair for an uncoloured vertex, -1 for a coloured one.
    // The largest key is found over blocks of 32 vertices whose maxima are kept up to date (keys of uncoloured vertices only grow;
    // the picked vertex's block is rescanned).
    constexpr int KB = 32;
    const int nblk = (n + KB - 1) / KB;
    std::vector<int64_t> key(n), bmax((size_t)nblk, -1);
    auto key_of = [&](int i) { return color[i] != 0 ? (int64_t)-1 : ((int64_t)ncol[i] << 32) | (int64_t)(uint32_t)degree[i]; };
    int left = 0;
    for (int i = 0; i < n; ++i) left += color[i] == 0;
    for (int i = 0; i < n; ++i)
        if (color[i] != 0)
            for (const int32_t* q = nb.begin(i); q != nb.end(i); ++q) mark(*q, color[i]);
    for (int i = 0; i < n; ++i) {
        key[i] = key_of(i);
        bmax[i / KB] = std::max(bmax[i / KB], key[i]);
    }
    while (left > 0) {
        int blk = 0;
        for (int b = 1; b < nblk; ++b)
            if (bmax[b] > bmax[blk]) blk = b;  // the first block that holds the largest key
        int idx = blk * KB;
        while (key[idx] != bmax[blk]) ++idx;
        int cpick = 1;
        while (has[(size_t)idx * cw + (cpick >> 6)] >> (cpick & 63) & 1) ++cpick;  // the smallest colour no neighbour carries
        color[idx] = cpick;
        --left;
        key[idx] = -1;
        bmax[blk] = -1;
        for (int i = blk * KB; i < std::min(n, blk * KB + KB); ++i) bmax[blk] = std::max(bmax[blk], key[i]);
        for (const int32_t* q = nb.begin(idx); q != nb.end(idx); ++q) {
            const int j = *q;
            mark(j, cpick);
            key[j] = key_of(j);
            bmax[j / KB] = std::max(bmax[j / KB], key[j]);
        }
    }
    // level matrix rows = colours in ascending order; order_topo (:91-131)
    int cmax = 0;
    for (int i = 0; i < n; ++i) cmax = std::max(cmax, color[i]);
    std::vector<int> row_of_colour((size_t)cmax + 1, -1);
    for (int i = 0; i < n; ++i) row_of_colour[color[i]] = 0;
    int nl = 0;
    for (int col = 0; col <= cmax; ++col)
        if (row_of_colour[col] == 0) row_of_colour[col] = nl++;
    std::vector<int> row(n);  // the level-matrix row a vertex stands in
    for (int v = 0; v < n; ++v) row[v] = row_of_colour[color[v]];
    std::vector<int> order;
    std::vector<int> place((size_t)nl, -1);  // position of a row in `order`
    long total = 0;
    for (long d : deg) total += d;
    if (total == 0) {
        for (int g = 0; g < nl; ++g) order.push_back(g);
    } else {
        while (total != 0) {
            int max_idx = 0;
            for (int i = 1; i < n; ++i)
                if (deg[i] > deg[max_idx]) max_idx = i;  // first index of the maximum
            const int lvl = row[max_idx];
            order.push_back(lvl);
            for (int i = 0; i < n; ++i)
                if (row[i] == lvl) deg[i] = 0;
            total = 0;
            for (long d : deg) total += d;
        }
        for (int g = 0; g < nl; ++g)
            if (std::find(order.begin(), order.end(), g) == order.end()) order.push_back(g);
    }
    for (size_t q = 0; q < order.size(); ++q)
        if (place[order[q]] < 0) place[order[q]] = (int)q;  // (find: the first position)
    std::vector<int> level(n, 0);
    for (int v = 0; v < n; ++v) level[v] = place[row[v]] + 1;
    directed.assign((size_t)n * n, 0);
    for (int i = 0; i < n; ++i)
        for_each_set(adjacency.data() + (size_t)i * n, n, [&](int j) {
            if (i != j && !(level[i] > level[j])) at(directed, n, i, j) = 1;  // Prioritizer.m:52-55
        });
}

// PrioritizedController.group (:375-389): weigh + GreedyCutter.cut (cut/GreedyCutter.m:5-86)
// (dir_succ / dir_pred: `directed` as lists by row / by column; uncut = nothing had to be cut: seq is `directed` and L its levels)
bool group(pdmpc_controller& c, const std::vector<uint8_t>& directed, const Lists& dir_succ, const Lists& dir_pred, std::vector<uint8_t>& seq,
           std::vector<int32_t>& L, bool& uncut) {
    const int n = c.n;
    uncut = false;
    if (!kahn_lists(dir_succ, n, L, c.kahn_indeg, c.kahn_cur, c.kahn_next)) return false;
    int depth = 0;
    for (int v : L) depth = std::max(depth, v);
    if (depth <= c.cfg.max_num_CLs) {
        seq = directed;  // every sub-graph of the DAG is at most as deep: the cutter accepts every edge
        uncut = true;
        return true;
    }
    seq.assign((size_t)n * n, 0);
    if (c.cfg.max_num_CLs == 1) return true;
    // weights; [row, col] = find(M): column-major order
    struct Edge { int a, b; double w; };
    std::vector<Edge> edges;
    const double vmax = *std::max_element(c.trim_speed.begin(), c.trim_speed.end());
    const double max_distance = 2 * vmax * c.cfg.dt_seconds * c.Hp;
    for (int b = 0; b < n; ++b)
        for (const int32_t* q = dir_pred.begin(b); q != dir_pred.end(b); ++q) {
            const int a = *q;
            double w = 0.5;  // ConstantWeigher
            if (c.cfg.weight_strategy == PDMPC_WEIGHT_DISTANCE) {
                const double dx = c.mx[a] - c.mx[b], dy = c.my[a] - c.my[b];
                w = 1 - std::sqrt(dx * dx + dy * dy) / max_distance;
            }
            if (w != 0) edges.push_back({a, b, w});  // (find() on the weighted matrix skips exact zeros)
        }
    std::stable_sort(edges.begin(), edges.end(), [](const Edge& p, const Edge& q) { return p.w > q.w; });
    // GreedyCutter.cut (:25-86) accepts an edge if the graph stays acyclic and at most max_num_CLs levels deep.  The levels are
    // longest-path layers (kahn), edges are only ever added, so the layers only grow: instead of a trial copy of the matrix and a
    // kahn pass per edge (128 vehicles: 6 ms per step), the new layers are relaxed from the edge's head through the accepted
    // successors; reaching the edge's tail again is a cycle, a layer beyond the limit a rejection (both undo the relaxation).
    std::vector<int32_t> levels((size_t)n, 1);  // (kahn of the graph without edges)
    std::vector<std::vector<int>> succ(n);
    std::vector<std::pair<int, int32_t>> undo;
    std::vector<int> work;
    for (const Edge& e : edges) {
        if (levels[e.a] < levels[e.b]) {
            at(seq, n, e.a, e.b) = 1;
            succ[e.a].push_back(e.b);
            continue;
        }
        undo.clear();
        work.clear();
        bool ok = levels[e.a] + 1 <= c.cfg.max_num_CLs;
        if (ok) {
            undo.emplace_back(e.b, levels[e.b]);
            levels[e.b] = levels[e.a] + 1;
            work.push_back(e.b);
        }
        while (ok && !work.empty()) {
            const int u = work.back();
            work.pop_back();
            for (int w : succ[u]) {
                if (levels[w] >= levels[u] + 1) continue;
                if (w == e.a || levels[u] + 1 > c.cfg.max_num_CLs) {  // a cycle / too deep
                    ok = false;
                    break;
                }
                undo.emplace_back(w, levels[w]);
                levels[w] = levels[u] + 1;
                work.push_back(w);
            }
        }
        if (ok) {
            at(seq, n, e.a, e.b) = 1;
            succ[e.a].push_back(e.b);
        } else {
            for (auto it = undo.rbegin(); it != undo.rend(); ++it) levels[it->first] = it->second;
        }
    }
    return true;
}

inline double ms_since(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); }

struct SetBuilder {  // builds a pdmpc_polygon_set whose arrays live in the controller's arena; one builder at a time
    pdmpc_controller& c;
    explicit SetBuilder(pdmpc_controller& ctl) : c(ctl) {
        c.sb_off.assign(1, 0);
        c.sb_x.clear();
        c.sb_y.clear();
    }
    void add(const Poly& p) {
        c.sb_x.insert(c.sb_x.end(), p.x.begin(), p.x.end());
        c.sb_y.insert(c.sb_y.end(), p.y.begin(), p.y.end());
        c.sb_off.push_back((int32_t)c.sb_x.size());
    }
    pdmpc_polygon_set finish() {
        pdmpc_polygon_set s;
        s.n_polygons = (int32_t)c.sb_off.size() - 1;
        const size_t np = c.sb_x.size();
        int32_t* off = (int32_t*)c.arena.take(c.sb_off.size() * sizeof(int32_t));
        double* x = (double*)c.arena.take((np + 1) * sizeof(double));  // (one entry more: never an empty array)
        double* y = (double*)c.arena.take((np + 1) * sizeof(double));
        std::memcpy(off, c.sb_off.data(), c.sb_off.size() * sizeof(int32_t));
        if (np) {
            std::memcpy(x, c.sb_x.data(), np * sizeof(double));
            std::memcpy(y, c.sb_y.data(), np * sizeof(double));
        }
        x[np] = y[np] = 0.0;
        s.offset = off;
        s.x = x;
        s.y = y;
        return s;
    }
};

int cfail(pdmpc_controller* c, int code, const std::string& msg) {
    g_cerr = msg;
    if (c) c->err = msg;
    return code;
}

}  // namespace

extern "C" {

const char* pdmpc_controller_last_error(void) { return g_cerr.c_str(); }

int pdmpc_controller_create(pdmpc_handle* handle, const pdmpc_controller_config* cfg, const pdmpc_scenario* sc, pdmpc_controller** out) {
    if (!cfg || !sc || !out) return cfail(nullptr, PDMPC_ERR_INVALID, "null argument");
    if (sc->n_vehicles < 1 || cfg->Hp < 1 || cfg->Hp > PDMPC_HP_MAX || sc->n_trims < 1) return cfail(nullptr, PDMPC_ERR_INVALID, "bad sizes");
    if (handle) {
        // the backend reads Hp entries of every reference and writes one record per vehicle: a handle created for another
        // horizon or a smaller batch must not be driven by this controller
        pdmpc_config hc{};
        int32_t has_mpa = 0;
        if (pdmpc_get_config(handle, &hc, &has_mpa) != PDMPC_OK) return cfail(nullptr, PDMPC_ERR_INVALID, "bad backend handle");
        if (hc.Hp != cfg->Hp) return cfail(nullptr, PDMPC_ERR_INVALID, "the handle was created for another horizon (config.Hp) than the controller");
        if (hc.max_vehicles < sc->n_vehicles) return cfail(nullptr, PDMPC_ERR_CAPACITY, "the handle's max_vehicles is smaller than the scenario");
        if (!has_mpa) return cfail(nullptr, PDMPC_ERR_NO_MPA, "pdmpc_upload_mpa has not been called on the handle");
    }
    pdmpc_controller* c = new pdmpc_controller();
    c->h = handle;
    c->cfg = *cfg;
    c->n = sc->n_vehicles;
    c->Hp = cfg->Hp;
    c->trim_speed.assign(sc->trim_speed, sc->trim_speed + sc->n_trims);
    c->trim_steering.assign(sc->trim_steering, sc->trim_steering + sc->n_trims);
    for (int v = 0; v < c->n; ++v) {
        VehicleDef d;
        d.x_start = sc->x_start[v];
        d.y_start = sc->y_start[v];
        d.yaw_start = sc->yaw_start[v];
        d.reference_speed = sc->reference_speed[v];
        d.px.assign(sc->path_x + sc->path_offset[v], sc->path_x + sc->path_offset[v + 1]);
        d.py.assign(sc->path_y + sc->path_offset[v], sc->path_y + sc->path_offset[v + 1]);
        if (d.px.size() < 2) {
            delete c;
            return cfail(nullptr, PDMPC_ERR_INVALID, "a reference path needs at least two points");
        }
        if (sc->lanelets_offset) {
            d.lanelets_index.assign(sc->lanelets_index + sc->lanelets_offset[v], sc->lanelets_index + sc->lanelets_offset[v + 1]);
            d.points_index.assign(sc->points_index + sc->lanelets_offset[v], sc->points_index + sc->lanelets_offset[v + 1]);
        }
        d.is_loop = sc->is_loop ? sc->is_loop[v] != 0 : true;
        d.tile_dx = sc->tile_dx ? sc->tile_dx[v] : 0.0;
        d.tile_dy = sc->tile_dy ? sc->tile_dy[v] : 0.0;
        c->veh.push_back(std::move(d));
    }
    for (int l = 0; l < sc->n_lanelets; ++l) {
        Poly a, b;
        a.x.assign(sc->left_x + sc->left_offset[l], sc->left_x + sc->left_offset[l + 1]);
        a.y.assign(sc->left_y + sc->left_offset[l], sc->left_y + sc->left_offset[l + 1]);
        b.x.assign(sc->right_x + sc->right_offset[l], sc->right_x + sc->right_offset[l + 1]);
        b.y.assign(sc->right_y + sc->right_offset[l], sc->right_y + sc->right_offset[l + 1]);
        c->bl_left.push_back(std::move(a));
        c->bl_right.push_back(std::move(b));
    }
    for (int p = 0; p < sc->obstacles.n_polygons; ++p) {
        Poly o;
        o.x.assign(sc->obstacles.x + sc->obstacles.offset[p], sc->obstacles.x + sc->obstacles.offset[p + 1]);
        o.y.assign(sc->obstacles.y + sc->obstacles.offset[p], sc->obstacles.y + sc->obstacles.offset[p + 1]);
        c->static_obstacles.push_back(std::move(o));
    }
    // Simulation.setup: initial speed = steering = 0 (Simulation.m:52-65)
    c->mx.resize(c->n);
    c->my.resize(c->n);
    c->myaw.resize(c->n);
    c->mspeed.assign(c->n, 0.0);
    c->msteer.assign(c->n, 0.0);
    for (int v = 0; v < c->n; ++v) {
        c->mx[v] = c->veh[v].x_start;
        c->my[v] = c->veh[v].y_start;
        c->myaw[v] = c->veh[v].yaw_start;
    }
    c->info_old.assign(c->n, Plan());
    c->infos.assign(c->n, Plan());
    *out = c;
    return PDMPC_OK;
}

int pdmpc_controller_destroy(pdmpc_controller* c) {
    delete c;
    return PDMPC_OK;
}

// Everything one launch needs to plan the whole time step (controller.py: build_step_problem): vehicles in level order
// (slot = position), per-slot predecessor slots, per-slot areas to publish on exhaustion.
namespace {
int assemble_step(pdmpc_controller* c, bool seq_given = false);
}

int pdmpc_controller_build_step(pdmpc_controller* c) {
    if (!c) return cfail(nullptr, PDMPC_ERR_INVALID, "null controller");
    const int n = c->n, Hp = c->Hp;
    c->k += 1;
    c->arena.reset();
    if (c->exploring) {
        c->obst_memo.resize((size_t)n);
        c->dyn_memo.resize((size_t)n);
        for (int v = 0; v < n; ++v) {
            c->obst_memo[(size_t)v].clear();
            c->dyn_memo[(size_t)v].clear();
        }
    }
    c->fb_of.assign(n, pdmpc_polygon_set());
    c->fb_done.assign(n, 0);
    c->empty_done = false;
    // ---- traffic info
    c->trims.assign(n, 0);
    // (resized, not re-created: the per-vehicle vectors keep their capacity from step to step; every one of them is rewritten below)
    c->occ_offset.resize(n);
    c->occ_plain.resize(n);
    c->ref_x.resize(n);
    c->ref_y.resize(n);
    c->v_ref.resize(n);
    c->bnd_left.resize(n);
    c->bnd_right.resize(n);
    std::vector<double> step(Hp);
    std::vector<int32_t> pidx;
    for (int v = 0; v < n; ++v) {
        c->trims[v] = trim_from_values(*c, c->mspeed[v], c->msteer[v]);
        occupied_areas(c->mx[v], c->my[v], c->myaw[v], c->cfg.vehicle_length, c->cfg.vehicle_width, c->cfg.offset, c->occ_offset[v], c->occ_plain[v]);
        // get_reference_trajectory.m:27-46
        std::vector<double>& vref = c->v_ref[v];
        vref.assign(Hp, c->veh[v].reference_speed);
        const double v_current = c->trim_speed[c->trims[v] - 1];
        for (int q = 0; q < Hp; ++q) step[q] = (((q == 0 ? v_current : vref[q - 1]) + vref[q]) / 2) * c->cfg.dt_seconds;
        int cpi = 0;
        sample_reference(Hp, c->veh[v].px, c->veh[v].py, c->mx[v], c->my[v], step, c->ref_x[v], c->ref_y[v], pidx, cpi);
        lanelet_boundary(*c, v, pidx, cpi, c->bnd_left[v], c->bnd_right[v]);
    }
    // ---- coupling
    c->adjacency.assign((size_t)n * n, 0);
    if (c->cfg.coupling == PDMPC_COUPLING_FULL) {
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) at(c->adjacency, n, a, b) = a != b;
    } else if (c->cfg.coupling == PDMPC_COUPLING_DISTANCE) {
        const double vmax = *std::max_element(c->trim_speed.begin(), c->trim_speed.end());
        const double max_distance = 2 * vmax * c->cfg.dt_seconds * Hp;
        // (hypot(dx, dy) >= max(|dx|, |dy|), also as rounded: a pair farther apart along one axis alone is not coupled — most pairs of
        // a tiled network.  That test runs over the whole row, the distance itself over the survivors.)
        for (int a = 0; a < n; ++a) {
            uint8_t* row = c->adjacency.data() + (size_t)a * n;
            const double xa = c->mx[a], ya = c->my[a];
            const double *px = c->mx.data(), *py = c->my.data();
            for (int b = a + 1; b < n; ++b) row[b] = (uint8_t)(!(std::fabs(xa - px[b]) > max_distance) & !(std::fabs(ya - py[b]) > max_distance));
            for_each_set(row + a + 1, n - a - 1, [&](int q) {
                const int b = a + 1 + q;
                row[b] = std::hypot(xa - px[b], ya - py[b]) <= max_distance;
                at(c->adjacency, n, b, a) = row[b];
            });
        }
    }
    // ---- priorities -> directed coupling
    if (c->cfg.priority_strategy == PDMPC_PRIORITY_COLORING) {
        coloring_directed(c->adjacency, n, c->directed);
    } else {  // constant priorities = vehicle index (ConstantPrioritizer.m:14-20): keep i -> j iff i <= j
        c->directed.assign((size_t)n * n, 0);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                if (at(c->adjacency, n, i, j) && !(j < i)) at(c->directed, n, i, j) = 1;
    }
    return assemble_step(c);
}

namespace {
// c->directed -> sequential couplings, levels, slot order and the per-slot inputs of pdmpc_plan_step (the arena is the caller's
// to clear: the explorative step keeps several problems alive side by side)
int assemble_step(pdmpc_controller* c, bool seq_given) {
    const int n = c->n, Hp = c->Hp;
    // (seq_given: c->directed_seq is the caller's -- the explorative step swaps single couplings of the base prioritization)
    // who a vehicle is coupled with, as lists: the loops below visit a vehicle's few couplings, not rows and columns of the matrices
    Lists &dir_succ = c->ls_dir_succ, &dir_pred = c->ls_dir_pred, &seq_succ_own = c->ls_seq_succ, &seq_pred_own = c->ls_seq_pred;
    lists_by_row(c->directed, n, dir_succ);
    lists_by_column(c->directed, n, dir_succ, dir_pred);
    bool uncut = false;
    if (!seq_given && !group(*c, c->directed, dir_succ, dir_pred, c->directed_seq, c->levels, uncut)) return cfail(c, PDMPC_ERR_INVALID, "coupling graph has a cycle");
    if (!uncut) {  // (uncut: the sequential coupling is `directed` itself, levels and lists included)
        lists_by_row(c->directed_seq, n, seq_succ_own);
        lists_by_column(c->directed_seq, n, seq_succ_own, seq_pred_own);
        if (!kahn_lists(seq_succ_own, n, c->levels, c->kahn_indeg, c->kahn_cur, c->kahn_next)) return cfail(c, PDMPC_ERR_INVALID, "coupling graph has a cycle");
    }
    const Lists& seq_pred = uncut ? dir_pred : seq_pred_own;
    // slot order: by level, vehicles of a level in index order (a counting sort over the levels 1 .. n)
    c->order.resize(n);
    c->slot_of.assign(n, 0);
    {
        std::vector<int>& first = c->kahn_cur;  // (scratch) first[l] = slot of level l's first vehicle
        first.assign((size_t)n + 2, 0);
        for (int i = 0; i < n; ++i) ++first[(size_t)c->levels[i] + 1];
        for (int l = 1; l <= n + 1; ++l) first[l] += first[l - 1];
        for (int i = 0; i < n; ++i) {
            const int s = first[(size_t)c->levels[i]]++;
            c->order[s] = i;
            c->slot_of[i] = s;
        }
    }
    // ---- per slot inputs
    c->in.assign(n, pdmpc_vehicle_in());
    c->fb.assign(n, pdmpc_polygon_set());
    c->pred_offset.assign(n + 1, 0);
    c->pred_index.clear();
    for (int s = 0; s < n; ++s) {
        const int i = c->order[s];
        pdmpc_vehicle_in& I = c->in[s];
        std::memset(&I, 0, sizeof I);
        I.x0 = c->mx[i];
        I.y0 = c->my[i];
        I.yaw0 = c->myaw[i];
        I.trim0 = c->trims[i];
        I.ref_x = c->ref_x[i].data();
        I.ref_y = c->ref_y[i].data();
        I.v_ref = c->v_ref[i].data();
        I.n_left = c->bnd_left[i].n();
        I.n_right = c->bnd_right[i].n();
        I.left_x = c->bnd_left[i].x.data();
        I.left_y = c->bnd_left[i].y.data();
        I.right_x = c->bnd_right[i].x.data();
        I.right_y = c->bnd_right[i].y.data();
        auto add_shifted = [](SetBuilder& b, const std::vector<Poly>& shapes) {  // del_first_rpt_last without the temporary
            for (size_t q = 1; q < shapes.size(); ++q) b.add(shapes[q]);
            b.add(shapes.back());
        };
        // who contributes (in the order the sets are built in): consider_predecessors (:449-506) — sequential predecessors are handed
        // over on the device; the others contribute their previous plan shifted by one step (parallel_coupling_previous_trajectory,
        // :409-447) —, then consider_successors (:508-566)
        const bool memo = c->exploring && n <= 512;  // (one prioritization: every set is built once anyway)
        pdmpc_controller::MemoKey ok, dk;
        if (memo) {
            std::memset(&ok, 0, sizeof ok);
            std::memset(&dk, 0, sizeof dk);
        }
        int ol[512], dpl[512], dsl[512], no = 0, ndp = 0, nds = 0;  // (the contributors in the order the sets are built in)
        std::vector<int> big;  // (n > 512: the lists on the heap)
        int *olp = ol, *dplp = dpl, *dslp = dsl;
        if (n > 512) {
            big.resize((size_t)3 * n);
            olp = big.data();
            dplp = big.data() + n;
            dslp = big.data() + 2 * n;
        }
        for (const int32_t* q = dir_pred.begin(i); q != dir_pred.end(i); ++q) {
            const int j = *q;
            if (at(c->directed_seq, n, j, i)) continue;
            if (c->info_old[j].present && c->k > 1) {
                dplp[ndp++] = j;
                if (memo) dk.w[j >> 6] |= 1ull << (j & 63);
            }
        }
        for (const int32_t* q = dir_succ.begin(i); q != dir_succ.end(i); ++q) {
            const int j = *q;
            if (c->cfg.constraint_from_successor == PDMPC_SUCCESSOR_AREA_OF_STANDSTILL) {
                if (std::fabs(c->mspeed[j]) < 0.01) {  // :536-540
                    olp[no++] = j;
                    if (memo) ok.w[j >> 6] |= 1ull << (j & 63);
                }
            } else if (c->cfg.constraint_from_successor == PDMPC_SUCCESSOR_AREA_OF_PREVIOUS_TRAJECTORY) {
                if (c->info_old[j].present) {
                    dslp[nds++] = j;
                    if (memo) dk.w[8 + (j >> 6)] |= 1ull << (j & 63);
                }
            }
        }
        auto build_obst = [&]() {
            SetBuilder obst(*c);
            for (const Poly& o : c->static_obstacles) obst.add(o);
            for (int q = 0; q < no; ++q) obst.add(c->occ_offset[olp[q]]);
            return obst.finish();
        };
        auto build_dyn = [&]() {
            SetBuilder dyn(*c);
            for (int q = 0; q < ndp; ++q) add_shifted(dyn, c->info_old[dplp[q]].shapes);
            for (int q = 0; q < nds; ++q) add_shifted(dyn, c->info_old[dslp[q]].shapes);
            return dyn.finish();
        };
        if (memo) {
            auto& om = c->obst_memo[(size_t)i];
            if (const pdmpc_polygon_set* hit = om.find(ok)) {
                I.obstacles = *hit;
            } else {
                I.obstacles = build_obst();
                om.keys.push_back(ok);
                om.sets.push_back(I.obstacles);
            }
            auto& dm = c->dyn_memo[(size_t)i];
            if (const pdmpc_polygon_set* hit = dm.find(dk)) {
                I.dynamic_obstacles = *hit;
            } else {
                I.dynamic_obstacles = build_dyn();
                dm.keys.push_back(dk);
                dm.sets.push_back(I.dynamic_obstacles);
            }
        } else {
            I.obstacles = build_obst();
            I.dynamic_obstacles = build_dyn();
        }
        if (!c->empty_done) {
            SetBuilder none(*c);
            c->empty_set = none.finish();
            c->empty_done = true;
        }
        I.hdv_reachable_sets = c->empty_set;
        // sequential predecessors as slots
        for (const int32_t* q = seq_pred.begin(i); q != seq_pred.end(i); ++q) c->pred_index.push_back(c->slot_of[*q]);
        c->pred_offset[s + 1] = (int32_t)c->pred_index.size();
        // what the vehicle publishes if its search is exhausted: its standstill rectangle (:602-611) or the previous plan
        // shifted by one step (:678-718)
        if (!c->fb_done[i]) {  // (a function of the vehicle alone: shared by the prioritizations of an explorative step)
            SetBuilder fbs(*c);
            const bool standstill = c->trim_speed[c->trims[i] - 1] == 0;
            if (standstill && c->cfg.constraint_from_successor != PDMPC_SUCCESSOR_NONE) {
                for (int q = 0; q < Hp; ++q) fbs.add(c->occ_plain[i]);
            } else if (c->info_old[i].present) {
                add_shifted(fbs, c->info_old[i].shapes);
            }
            c->fb_of[i] = fbs.finish();
            c->fb_done[i] = 1;
        }
        c->fb[s] = c->fb_of[i];
    }
    c->pred_index.push_back(0);
    return PDMPC_OK;
}
}  // namespace

// records of the step in slot order -> plans, exhaustion handling, fallbacks of coupled vehicles, plant update
int pdmpc_controller_apply(pdmpc_controller* c, const pdmpc_vehicle_out* recs) {
    if (!c || !recs) return cfail(c, PDMPC_ERR_INVALID, "null argument");
    const int n = c->n, Hp = c->Hp;
    auto fallback_plan = [&](int i, Plan& p) -> bool {  // plan_fallback (:678-718): the previous plan shifted by one step
        const Plan& old = c->info_old[i];
        if (!old.present) return false;
        p.present = true;
        const size_t m = old.shapes.size();
        p.shapes.resize(m);
        p.trims.resize(m);
        p.yx.resize(m);
        p.yy.resize(m);
        p.yyaw.resize(m);
        for (size_t q = 0; q < m; ++q) {
            const size_t from = std::min(q + 1, m - 1);
            p.shapes[q] = old.shapes[from];
            p.trims[q] = old.trims[from];
            p.yx[q] = old.yx[from];
            p.yy[q] = old.yy[from];
            p.yyaw[q] = old.yyaw[from];
        }
        return true;
    };
    // The step's plans are built in c->infos — nothing else reads it — and swapped with c->info_old at the end: an error status or a
    // fallback in the first step leaves the controller's plans as they were, and the vectors of a plan keep their capacity from
    // step to step (they are overwritten entry by entry, not re-created).
    std::vector<Plan>& infos = c->infos;
    infos.resize((size_t)n);
    if (c->last_pops.size() != (size_t)n) c->last_pops.assign((size_t)n, 0.0);
    for (int s = 0; s < n; ++s) {
        const int i = c->order[s];
        const pdmpc_vehicle_out& r = recs[s];
        if (r.status != PDMPC_OK && r.status != PDMPC_EXHAUSTED) return cfail(c, PDMPC_ERR_HIP, "a result record carries an error status: not a planning result");
        Plan& p = infos[i];
        p.present = p.needs_fallback = p.exhausted = false;
        p.n_expanded = r.n_expanded;
        c->last_pops[(size_t)i] = (double)r.n_popped;
        if (r.status == PDMPC_OK) {
            p.present = true;
            p.shapes.resize((size_t)Hp);
            p.trims.resize((size_t)Hp);
            p.yx.resize((size_t)Hp);
            p.yy.resize((size_t)Hp);
            p.yyaw.resize((size_t)Hp);
            for (int q = 0; q < Hp; ++q) {
                Poly& sh = p.shapes[q];
                sh.x.assign(r.shapes[q][0], r.shapes[q][0] + r.shape_cols[q]);
                sh.y.assign(r.shapes[q][1], r.shapes[q][1] + r.shape_cols[q]);
                p.trims[q] = r.predicted_trims[q];
                p.yx[q] = r.y_predicted[q][0];
                p.yy[q] = r.y_predicted[q][1];
                p.yyaw[q] = r.y_predicted[q][2];
            }
        } else {  // PrioritizedController.m:344-352
            p.exhausted = true;
            const bool standstill = c->trim_speed[c->trims[i] - 1] == 0;
            if (standstill && c->cfg.constraint_from_successor != PDMPC_SUCCESSOR_NONE) {  // handle_graph_search_exhaustion (:568-616)
                p.present = true;
                p.shapes.assign((size_t)Hp, c->occ_plain[i]);
                p.trims.assign((size_t)Hp, c->trims[i]);
                p.yx.assign((size_t)Hp, c->mx[i]);
                p.yy.assign((size_t)Hp, c->my[i]);
                p.yyaw.assign((size_t)Hp, c->myaw[i]);
            } else {
                if (!fallback_plan(i, p)) return cfail(c, PDMPC_ERR_INVALID, "a vehicle needs a fallback in its first step");
                p.needs_fallback = true;
            }
        }
    }
    // handle_others_fallback / check_others_fallback
    bool any = false;
    for (int i = 0; i < n; ++i) any = any || infos[i].needs_fallback;
    if (any) {
        std::vector<int> fm((size_t)n * n, 0);
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) {
                int v = at(c->adjacency, n, a, b);
                if (infos[a].needs_fallback && at(c->directed_seq, n, a, b)) v -= 1;
                if (infos[b].needs_fallback && at(c->directed_seq, n, b, a)) v -= 1;
                fm[(size_t)a * n + b] = v;
            }
        std::vector<uint8_t> reached(n, 0);
        for (int f = 0; f < n; ++f) {
            if (!infos[f].needs_fallback) continue;
            std::vector<uint8_t> seen(n, 0);
            std::vector<int> stack{f};
            seen[f] = 1;
            while (!stack.empty()) {
                const int a = stack.back();
                stack.pop_back();
                for (int b = 0; b < n; ++b)
                    if (fm[(size_t)a * n + b] != 0 && !seen[b]) {
                        seen[b] = 1;
                        stack.push_back(b);
                    }
            }
            for (int v = 0; v < n; ++v) reached[v] |= seen[v];
        }
        for (int i = 0; i < n; ++i)
            if (reached[i] && !infos[i].needs_fallback) {
                Plan& p = infos[i];  // (keeps its search's n_expanded and exhausted)
                if (!fallback_plan(i, p)) return cfail(c, PDMPC_ERR_INVALID, "a vehicle needs a fallback in its first step");
                p.needs_fallback = false;  // plan_fallback(is_fallback_while_planning = false)
            }
    }
    std::swap(c->info_old, c->infos);
    // Simulation.apply (Simulation.m:86-100)
    for (int i = 0; i < n; ++i) {
        const Plan& p = c->info_old[i];
        c->mx[i] = p.yx[0];
        c->my[i] = p.yy[0];
        c->myaw[i] = p.yyaw[0];
        c->mspeed[i] = c->trim_speed[p.trims[0] - 1];
        c->msteer[i] = c->trim_steering[p.trims[0] - 1];
    }
    return PDMPC_OK;
}

// One pass of HighLevelController.main_control_loop (:334-373) in simulation: build, plan on the GPU (one launch), apply.
namespace {
// adds the parts of the step's one backend call (pdmpc_last_call_timing) to the controller's timing
void add_call_timing(pdmpc_controller* c) {
    double us[3] = {0, 0, 0};
    if (pdmpc_last_call_timing(c->h, us) == PDMPC_OK)
        for (int i = 0; i < 3; ++i) c->timing[1 + i] = us[i] * 1e-3;
}
}  // namespace

int pdmpc_controller_last_timing(pdmpc_controller* c, double* ms6) {
    if (!c || !ms6) return cfail(c, PDMPC_ERR_INVALID, "null argument");
    for (int i = 0; i < 6; ++i) ms6[i] = c->timing[i];
    return PDMPC_OK;
}

int pdmpc_controller_timing_sum(pdmpc_controller* c, double* ms6, int64_t* n_steps, int32_t reset) {
    if (!c) return cfail(c, PDMPC_ERR_INVALID, "null argument");
    if (ms6)
        for (int i = 0; i < 6; ++i) ms6[i] = c->timing_sum[i];
    if (n_steps) *n_steps = c->timing_steps;
    if (reset) {
        for (int i = 0; i < 6; ++i) c->timing_sum[i] = 0;
        c->timing_steps = 0;
    }
    return PDMPC_OK;
}

int pdmpc_controller_step(pdmpc_controller* c) {
    if (!c || !c->h) return cfail(c, PDMPC_ERR_INVALID, "controller has no backend handle");
    auto t = std::chrono::steady_clock::now();
    int rc = pdmpc_controller_build_step(c);
    if (rc) return rc;
    c->timing[0] = ms_since(t);
    c->timing[4] = 0;
    c->out.resize(c->n);
    if (c->last_pops.size() == (size_t)c->n) {  // the work of the last step as the expected work of this one: heavy searches are dispatched first
        std::vector<double> w((size_t)c->n);
        for (int s = 0; s < c->n; ++s) w[(size_t)s] = c->last_pops[(size_t)c->order[(size_t)s]] + 1.0;
        (void)pdmpc_set_step_weights(c->h, c->n, w.data());
    }
    rc = pdmpc_plan_step(c->h, c->n, c->in.data(), c->pred_offset.data(), c->pred_index.data(), c->fb.data(), c->out.data());
    if (rc) return cfail(c, rc, pdmpc_last_error());
    add_call_timing(c);
    t = std::chrono::steady_clock::now();
    rc = pdmpc_controller_apply(c, c->out.data());
    c->timing[5] = ms_since(t);
    for (int i = 0; i < 6; ++i) c->timing_sum[i] += c->timing[i];
    c->timing_steps += 1;
    return rc;
}

// n_steps closed-loop time steps in one call; ms[i] (may be NULL) receives the wall time of step i: build + pack + launch +
// fetch + apply, everything a caller of the boundary pays per MPC step
int pdmpc_controller_run(pdmpc_controller* c, int32_t n_steps, double* ms) {
    for (int i = 0; i < n_steps; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = pdmpc_controller_step(c);
        if (rc) return rc;
        if (ms) ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return PDMPC_OK;
}

int pdmpc_controller_problem(pdmpc_controller* c, int32_t* n, const pdmpc_vehicle_in** in, const int32_t** pred_offset, const int32_t** pred_index,
                             const pdmpc_polygon_set** fallback, const int32_t** order, const int32_t** levels) {
    if (!c) return cfail(nullptr, PDMPC_ERR_INVALID, "null controller");
    if (n) *n = c->n;
    if (in) *in = c->in.data();
    if (pred_offset) *pred_offset = c->pred_offset.data();
    if (pred_index) *pred_index = c->pred_index.data();
    if (fallback) *fallback = c->fb.data();
    if (order) *order = c->order.data();
    if (levels) *levels = c->levels.data();
    return PDMPC_OK;
}

int pdmpc_controller_state(pdmpc_controller* c, double* x, double* y, double* yaw, double* speed, double* steering, int32_t* needs_fallback, int32_t* time_step) {
    if (!c) return cfail(nullptr, PDMPC_ERR_INVALID, "null controller");
    for (int i = 0; i < c->n; ++i) {
        if (x) x[i] = c->mx[i];
        if (y) y[i] = c->my[i];
        if (yaw) yaw[i] = c->myaw[i];
        if (speed) speed[i] = c->mspeed[i];
        if (steering) steering[i] = c->msteer[i];
        if (needs_fallback) needs_fallback[i] = c->info_old[i].present && c->info_old[i].needs_fallback;
    }
    if (time_step) *time_step = c->k;
    return PDMPC_OK;
}

// PrioritizedExplorativeController.computation_level_permutations (:241-309): n_perm x n_levels table, row-major, row 0 = 1..n;
// rows up to n_levels form a Latin square built "fewest possibilities first" with random choices from
// RandStream("mt19937ar", Seed = seed) / randi (:249, :283-286), a row that meets a dead end is drawn again; further rows
// (the reference stops at n_levels; BASELINE config C5 asks for 64) are Fisher-Yates shuffles from the same stream.
// The twin of pdmpc.explorative.computation_level_permutations.
int pdmpc_exploration_permutations(int32_t n_levels, int32_t n_perm, uint32_t seed, int32_t* out) {
    if (n_levels < 1 || n_perm < 1 || !out) return cfail(nullptr, PDMPC_ERR_INVALID, "bad argument");
    Mt19937ar rng(seed);
    const int n = n_levels;
    std::vector<std::vector<int32_t>> rows;
    rows.emplace_back();
    for (int j = 0; j < n; ++j) rows[0].push_back(j + 1);
    while ((int)rows.size() < std::min(n_perm, n_levels)) {
        std::vector<uint8_t> allowed((size_t)n * n, 1);  // [level][class]
        for (int col = 0; col < n; ++col)
            for (const auto& r : rows) allowed[(size_t)(r[(size_t)col] - 1) * n + col] = 0;
        std::vector<int32_t> perm((size_t)n, 0);
        bool ok = true;
        for (int filled = 0; filled < n && ok; ++filled) {
            int best_col = 0, best_cnt = n + 1;
            for (int col = 0; col < n; ++col) {  // [n_possibilities, i_cell] = min(sum(is_level_allowed, 1)): the first minimum
                int cnt = 0;
                for (int l = 0; l < n; ++l) cnt += allowed[(size_t)l * n + col];
                if (cnt < best_cnt) {
                    best_cnt = cnt;
                    best_col = col;
                }
            }
            if (best_cnt == 0) {
                ok = false;
                break;
            }
            const int pick = rng.randi(best_cnt);  // 1-based position among find(is_level_allowed(:, i_cell))
            int lvl = -1;
            for (int l = 0, seen = 0; l < n; ++l)
                if (allowed[(size_t)l * n + best_col] && ++seen == pick) {
                    lvl = l;
                    break;
                }
            perm[(size_t)best_col] = lvl + 1;
            for (int col = 0; col < n; ++col) allowed[(size_t)lvl * n + col] = 0;
            for (int l = 0; l < n; ++l) allowed[(size_t)l * n + best_col] = 1;
        }
        if (ok) rows.push_back(perm);
    }
    while ((int)rows.size() < n_perm) {
        std::vector<int32_t> perm((size_t)n);
        for (int j = 0; j < n; ++j) perm[(size_t)j] = j + 1;
        for (int i = n - 1; i > 0; --i) std::swap(perm[(size_t)i], perm[(size_t)(rng.randi(i + 1) - 1)]);
        rows.push_back(perm);
    }
    for (int p = 0; p < n_perm; ++p)
        for (int j = 0; j < n; ++j) out[(size_t)p * n + j] = rows[(size_t)p][(size_t)j];
    return PDMPC_OK;
}

// ---- the explorative step (twin of pdmpc.explorative.build_exploration_batch / choose_solution / explore_step)
// PrioritizedExplorativeController.m:25-91: the step's traffic state under n_perm prioritizations, one flattened batch: instance p
// permutes the computation levels of the base prioritization (prepare_permutation :42-58: a vehicle of level L gets the position
// of L in permutation p as its priority), slots ordered by (level, instance, slot).  Advances the time step like build_step.
int pdmpc_controller_explore_build(pdmpc_controller* c, int32_t n_perm, uint32_t seed) {
    if (!c || n_perm < 1) return cfail(c, PDMPC_ERR_INVALID, "bad argument");
    struct Exploring {  // (the memos of the obstacle sets are on while this step's prioritizations are assembled)
        pdmpc_controller* c;
        explicit Exploring(pdmpc_controller* ctl) : c(ctl) { c->exploring = true; }
        ~Exploring() { c->exploring = false; }
    } exploring(c);
    int rc = pdmpc_controller_build_step(c);  // instance 0: the controller's own prioritization
    if (rc) return rc;
    const int n = c->n;
    // base levels: the computation levels of the controller's own prioritization -- kahn of the sequential coupling the step was
    // just built with, whatever the priority strategy (PrioritizedExplorativeController.m prepare_permutation :42-58 permutes
    // kahn(iter.directed_coupling_sequential))
    const std::vector<int32_t> levels0 = c->levels;
    const int n_levels = *std::max_element(levels0.begin(), levels0.end());
    std::vector<int32_t> perms((size_t)n_perm * n_levels);
    rc = pdmpc_exploration_permutations(n_levels, n_perm, seed, perms.data());
    if (rc) return rc;
    // (copies into vectors that are kept from step to step: no allocation once warm)
    using Part = pdmpc_controller::Part;
    std::vector<Part>& parts = c->x_parts;
    if (parts.size() < (size_t)n_perm) parts.resize((size_t)n_perm);
    if (c->inst.size() != (size_t)n_perm) c->inst.resize((size_t)n_perm);
    auto keep = [&](int p) {
        Part& P = parts[(size_t)p];
        P.in = c->in;
        P.fb = c->fb;
        P.pred_offset = c->pred_offset;
        P.pred_index = c->pred_index;
        pdmpc_controller::Instance& I = c->inst[(size_t)p];
        I.directed = c->directed;
        I.directed_seq = c->directed_seq;
        I.levels = c->levels;
        I.order = c->order;
        I.slot_of = c->slot_of;
    };
    keep(0);
    std::vector<int32_t> where;
    for (int p = 1; p < n_perm; ++p) {
        where.assign((size_t)n_levels + 1, 0);
        for (int j = 0; j < n_levels; ++j) where[(size_t)perms[(size_t)p * n_levels + j]] = j + 1;
        // prepare_permutation (:64-77): every coupling i -> j of the base prioritization whose permuted levels invert it is swapped
        // in ALL coupling matrices (swap_entries_all_coupling_matrices): a sequential coupling stays sequential, a parallel one
        // (cut by the grouping, or between vehicles of one level) stays parallel and keeps its direction
        const pdmpc_controller::Instance& I0 = c->inst[0];
        c->directed = I0.directed;
        c->directed_seq = I0.directed_seq;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                if (at(I0.directed, n, i, j) && where[(size_t)levels0[i]] > where[(size_t)levels0[j]]) {
                    at(c->directed, n, i, j) = 0;
                    at(c->directed, n, j, i) = 1;
                    if (at(I0.directed_seq, n, i, j)) {
                        at(c->directed_seq, n, i, j) = 0;
                        at(c->directed_seq, n, j, i) = 1;
                    }
                }
        rc = assemble_step(c, true);
        if (rc) return rc;
        keep(p);
    }
    // flatten: (level, instance, slot)
    struct Key {
        int32_t level, p, s;
    };
    std::vector<Key> flat;
    for (int p = 0; p < n_perm; ++p)
        for (int s = 0; s < n; ++s) flat.push_back(Key{c->inst[(size_t)p].levels[(size_t)c->inst[(size_t)p].order[(size_t)s]], p, s});
    std::stable_sort(flat.begin(), flat.end(), [](const Key& a, const Key& b) { return a.level < b.level; });  // (generated in (p, s) order)
    const int N = n_perm * n;
    std::vector<int32_t> slot_of((size_t)N);  // [p * n + s]
    for (int i = 0; i < N; ++i) slot_of[(size_t)flat[(size_t)i].p * n + flat[(size_t)i].s] = i;
    c->x_in.resize((size_t)N);
    c->x_fb.resize((size_t)N);
    c->x_pred_offset.assign((size_t)N + 1, 0);
    c->x_pred_index.clear();
    c->x_instance.resize((size_t)N);
    c->x_vehicle.resize((size_t)N);
    c->x_level.resize((size_t)N);
    c->x_slot.assign((size_t)N, 0);
    for (int i = 0; i < N; ++i) {
        const Key& k = flat[(size_t)i];
        const Part& P = parts[(size_t)k.p];
        c->x_in[(size_t)i] = P.in[(size_t)k.s];
        c->x_fb[(size_t)i] = P.fb[(size_t)k.s];
        for (int32_t q = P.pred_offset[(size_t)k.s]; q < P.pred_offset[(size_t)k.s + 1]; ++q) c->x_pred_index.push_back(slot_of[(size_t)k.p * n + P.pred_index[(size_t)q]]);
        c->x_pred_offset[(size_t)i + 1] = (int32_t)c->x_pred_index.size();
        c->x_instance[(size_t)i] = k.p;
        c->x_vehicle[(size_t)i] = c->inst[(size_t)k.p].order[(size_t)k.s];
        c->x_level[(size_t)i] = k.level;
        c->x_slot[(size_t)k.p * n + c->x_vehicle[(size_t)i]] = i;
    }
    c->x_pred_index.push_back(0);
    // the controller's own problem again (instance 0)
    c->in = parts[0].in;
    c->fb = parts[0].fb;
    c->pred_offset = parts[0].pred_offset;
    c->pred_index = parts[0].pred_index;
    const pdmpc_controller::Instance& I0 = c->inst[0];
    c->directed = I0.directed;
    c->directed_seq = I0.directed_seq;
    c->levels = I0.levels;
    c->order = I0.order;
    c->slot_of = I0.slot_of;
    return PDMPC_OK;
}

int pdmpc_controller_explore_problem(pdmpc_controller* c, int32_t* n_slots, const pdmpc_vehicle_in** in, const int32_t** pred_offset, const int32_t** pred_index,
                                     const pdmpc_polygon_set** fallback, const int32_t** instance, const int32_t** vehicle, const int32_t** level) {
    if (!c || c->x_in.empty()) return cfail(c, PDMPC_ERR_INVALID, "no exploration batch has been built");
    if (n_slots) *n_slots = (int32_t)c->x_in.size();
    if (in) *in = c->x_in.data();
    if (pred_offset) *pred_offset = c->x_pred_offset.data();
    if (pred_index) *pred_index = c->x_pred_index.data();
    if (fallback) *fallback = c->x_fb.data();
    if (instance) *instance = c->x_instance.data();
    if (vehicle) *vehicle = c->x_vehicle.data();
    if (level) *level = c->x_level.data();
    return PDMPC_OK;
}

// compute_solution_cost / choose_solution (:94-176): per weakly connected sub-graph of the coupling graph the instance with the
// smallest sum of the cost-to-come of the vehicles' final nodes after round(., 8); a vehicle whose search was exhausted makes its
// instance infinitely expensive.  chosen[v] = instance of vehicle v's sub-graph; cost (may be NULL): n_perm x n_graphs, graphs
// ordered by their smallest vehicle.  The chosen instances' couplings become the controller's (what apply's fallback handling sees).
namespace {
int explore_choose_on(pdmpc_controller* c, const int32_t* status, const double* final_cost, int32_t* chosen, int32_t* n_graphs, double* cost);
}
int pdmpc_controller_explore_choose(pdmpc_controller* c, const pdmpc_vehicle_out* recs, int32_t* chosen, int32_t* n_graphs, double* cost) {
    if (!c || !recs || c->inst.empty()) return cfail(c, PDMPC_ERR_INVALID, "no exploration batch has been built");
    const int N = (int)c->inst.size() * c->n;
    std::vector<int32_t> st((size_t)N);
    std::vector<double> fc((size_t)N);
    for (int s = 0; s < N; ++s) {
        st[(size_t)s] = recs[s].status;
        fc[(size_t)s] = recs[s].path_nodes[c->Hp][4];
    }
    return explore_choose_on(c, st.data(), fc.data(), chosen, n_graphs, cost);
}
namespace {
// (status and cost-to-come of the final node per slot of the batch: all the choice looks at)
int explore_choose_on(pdmpc_controller* c, const int32_t* status, const double* final_cost, int32_t* chosen, int32_t* n_graphs, double* cost) {
    const int n = c->n, K = (int)c->inst.size();
    std::vector<int> label((size_t)n);
    for (int i = 0; i < n; ++i) label[(size_t)i] = i;
    auto find = [&](int a) {
        while (label[(size_t)a] != a) a = label[(size_t)a] = label[(size_t)label[(size_t)a]];
        return a;
    };
    const std::vector<uint8_t>& seq0 = c->inst[0].directed_seq;  // conncomp(directed_coupling_sequential) of the base prioritization (:94-112)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            if (at(seq0, n, i, j) || at(seq0, n, j, i)) {
                const int a = find(i), b = find(j);
                if (a != b) label[(size_t)std::max(a, b)] = std::min(a, b);
            }
    std::vector<int> graph_of((size_t)n), roots;
    for (int i = 0; i < n; ++i)
        if (find(i) == i) roots.push_back(i);  // ascending: the graphs ordered by their smallest vehicle
    for (int i = 0; i < n; ++i) graph_of[(size_t)i] = (int)(std::lower_bound(roots.begin(), roots.end(), find(i)) - roots.begin());
    const int G = (int)roots.size();
    c->x_graphs = G;
    c->x_cost.assign((size_t)K * G, 0.0);
    const int N = K * n;
    for (int s = 0; s < N; ++s) {  // (slot order: the order the twin adds in)
        if (status[s] != PDMPC_OK && status[s] != PDMPC_EXHAUSTED) return cfail(c, PDMPC_ERR_HIP, "a result record carries an error status: not a planning result");
        const double v = status[s] == PDMPC_OK ? final_cost[s] : std::numeric_limits<double>::infinity();
        c->x_cost[(size_t)c->x_instance[(size_t)s] * G + graph_of[(size_t)c->x_vehicle[(size_t)s]]] += v;
    }
    for (double& v : c->x_cost) v = std::nearbyint(v * 1e8) / 1e8;
    std::vector<int> best((size_t)G, 0);
    for (int g = 0; g < G; ++g)
        for (int p = 1; p < K; ++p)
            if (c->x_cost[(size_t)p * G + g] < c->x_cost[(size_t)best[(size_t)g] * G + g]) best[(size_t)g] = p;  // [~, chosen] = min(.): the first minimum
    c->x_chosen.resize((size_t)n);
    for (int i = 0; i < n; ++i) c->x_chosen[(size_t)i] = best[(size_t)graph_of[(size_t)i]];
    // obj.iter = obj.iter_array_tmp{chosen_solution} (:157-158): every vehicle goes on with the couplings of its sub-graph's choice
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const pdmpc_controller::Instance& I = c->inst[(size_t)c->x_chosen[(size_t)i]];
            at(c->directed, n, i, j) = at(I.directed, n, i, j);
            at(c->directed_seq, n, i, j) = at(I.directed_seq, n, i, j);
        }
    if (chosen) std::copy(c->x_chosen.begin(), c->x_chosen.end(), chosen);
    if (n_graphs) *n_graphs = G;
    if (cost) std::copy(c->x_cost.begin(), c->x_cost.end(), cost);
    return PDMPC_OK;
}
}  // namespace

// One explorative time step: build the batch, plan all prioritizations with ONE launch, choose per sub-graph, apply the chosen plans.
int pdmpc_controller_explore_step(pdmpc_controller* c, int32_t n_perm) {
    if (!c || !c->h) return cfail(c, PDMPC_ERR_INVALID, "controller has no backend handle");
    auto t = std::chrono::steady_clock::now();
    int rc = pdmpc_controller_explore_build(c, n_perm, (uint32_t)(c->k + 1));  // RandStream("mt19937ar", Seed = obj.k) (:249)
    if (rc) return rc;
    c->timing[0] = ms_since(t);
    const int N = (int)c->x_in.size();
    c->x_out.resize((size_t)N);
    if (c->last_pops.size() == (size_t)c->n) {
        std::vector<double> w((size_t)N);
        for (int i = 0; i < N; ++i) w[(size_t)i] = c->last_pops[(size_t)c->x_vehicle[(size_t)i]] + 1.0;
        (void)pdmpc_set_step_weights(c->h, N, w.data());
    }
    if (c->lean_explore) {
        // the closed loop keeps the chosen plans only (obj.iter = obj.iter_array_tmp{chosen_solution}, :157-158): status and final
        // cost of every plan come back for the choice, the chosen vehicles' records afterwards — not 2.9 KB for each of the N plans
        c->x_out.clear();
        c->x_status.resize((size_t)N);
        c->x_final_cost.resize((size_t)N);
        rc = pdmpc_plan_step_lean(c->h, N, c->x_in.data(), c->x_pred_offset.data(), c->x_pred_index.data(), c->x_fb.data(), c->x_status.data(), c->x_final_cost.data());
        if (rc) return cfail(c, rc, pdmpc_last_error());
        add_call_timing(c);
        t = std::chrono::steady_clock::now();
        rc = explore_choose_on(c, c->x_status.data(), c->x_final_cost.data(), nullptr, nullptr, nullptr);
        if (rc) return rc;
        std::vector<int32_t> want((size_t)c->n);
        for (int s = 0; s < c->n; ++s) {
            const int v = c->order[(size_t)s];
            want[(size_t)s] = c->x_slot[(size_t)(c->follow_own ? 0 : c->x_chosen[(size_t)v]) * c->n + v];
        }
        c->out.resize((size_t)c->n);
        rc = pdmpc_fetch_records_at(c->h, c->n, want.data(), c->out.data());
        if (rc) return cfail(c, rc, pdmpc_last_error());
    } else {
        rc = pdmpc_plan_step(c->h, N, c->x_in.data(), c->x_pred_offset.data(), c->x_pred_index.data(), c->x_fb.data(), c->x_out.data());
        if (rc) return cfail(c, rc, pdmpc_last_error());
        add_call_timing(c);
        t = std::chrono::steady_clock::now();
        rc = pdmpc_controller_explore_choose(c, c->x_out.data(), nullptr, nullptr, nullptr);
        if (rc) return rc;
        c->out.resize((size_t)c->n);
        for (int s = 0; s < c->n; ++s) {
            const int v = c->order[(size_t)s];
            c->out[(size_t)s] = c->x_out[(size_t)c->x_slot[(size_t)(c->follow_own ? 0 : c->x_chosen[(size_t)v]) * c->n + v]];
        }
    }
    if (c->follow_own) {  // (the couplings of instance 0 again: what apply's fallback handling sees)
        c->directed = c->inst[0].directed;
        c->directed_seq = c->inst[0].directed_seq;
    }
    c->timing[4] = ms_since(t);
    t = std::chrono::steady_clock::now();
    rc = pdmpc_controller_apply(c, c->out.data());
    c->timing[5] = ms_since(t);
    for (int i = 0; i < 6; ++i) c->timing_sum[i] += c->timing[i];
    c->timing_steps += 1;
    return rc;
}

int pdmpc_controller_explore_follow_own(pdmpc_controller* c, int32_t on) {
    if (!c) return cfail(c, PDMPC_ERR_INVALID, "null argument");
    c->follow_own = on != 0;
    return PDMPC_OK;
}

int pdmpc_controller_explore_run(pdmpc_controller* c, int32_t n_perm, int32_t n_steps, double* ms) {
    if (!c) return cfail(c, PDMPC_ERR_INVALID, "null argument");
    struct Lean {  // (nobody looks at the plans that were not chosen: pdmpc_controller_explore_step keeps them all, this loop does not)
        pdmpc_controller* c;
        bool was;
        ~Lean() { c->lean_explore = was; }
    } lean{c, c->lean_explore};
    c->lean_explore = true;
    for (int i = 0; i < n_steps; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = pdmpc_controller_explore_step(c, n_perm);
        if (rc) return rc;
        if (ms) ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return PDMPC_OK;
}

int pdmpc_controller_explore_result(pdmpc_controller* c, int32_t* chosen, int32_t* n_graphs, const double** cost, const pdmpc_vehicle_out** records) {
    if (!c || c->x_chosen.empty()) return cfail(c, PDMPC_ERR_INVALID, "no explorative step has been chosen");
    if (chosen) std::copy(c->x_chosen.begin(), c->x_chosen.end(), chosen);
    if (n_graphs) *n_graphs = c->x_graphs;
    if (cost) *cost = c->x_cost.data();
    if (records) *records = c->x_out.empty() ? nullptr : c->x_out.data();
    return PDMPC_OK;
}

const pdmpc_vehicle_out* pdmpc_controller_records(pdmpc_controller* c) { return c && !c->out.empty() ? c->out.data() : nullptr; }

}  // extern "C"
