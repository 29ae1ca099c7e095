// pdmpc_oracle.cpp — CPU restatement of p-dmpc's graph-search optimizer.  TEST INFRASTRUCTURE ONLY.
//
// This file is the parity oracle for the HIP backend.  Only tests/, __graft_entry__.smoke() and
// bench.py's `cpu_baseline` leg may load it; nothing under p-dmpc_amd/ links, imports or calls it.
//
// It follows the reference's MATLAB sources line by line (citations are relative to the reference
// repository root) and keeps the reference's own priority-queue type and comparator
// (hlc/optimizer/graph_search/priority_queue/priority_queue_interface_mex.cpp:19-31), so libstdc++'s
// heap tie order is inherited, not imitated.
//
// Pinning status.  The reference path is MATLAB; it cannot run here and its only native file needs
// mex.hpp, so there is no reference build under oracle/_ref.  The oracle is pinned against every
// known-answer vector the reference's tests hold for this path
// (tests/unittests/hlc/intersect_unittest.m:8-54, see tests/test_oracle_golden.py).  For MATLAB
// built-ins (cos, sin, norm, vecnorm, BLAS-backed products) **parity is unpinned**: this file
// evaluates them as single IEEE double operations in source order (no FMA contraction) and uses
// include/pdmpc_math.h for sin/cos so CPU and GPU agree bit for bit.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off -shared -fPIC).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <queue>
#include <thread>
#include <tuple>
#include <vector>

#include "../include/pdmpc.h"
#include "../include/pdmpc_math.h"

// Arithmetic variants for tools/tolerance_study.py (never used by the parity tests): MATLAB's cos / sin / norm / matrix
// products are closed source, so the oracle's single-IEEE-operation reading of them is one of several plausible ones.
//   ORACLE_LIBM_SINCOS   glibc's sin / cos instead of include/pdmpc_math.h
//   ORACLE_HYPOT_NORM    hypot(dx, dy) (scaled, correctly rounded in glibc) instead of sqrt(dx*dx + dy*dy) for norm / vecnorm
//   (FMA contraction of the dot products: the same file compiled with -ffp-contract=fast -mfma)
#ifdef ORACLE_LIBM_SINCOS
#define ORACLE_SINCOS(x, s, c) (*(s) = std::sin(x), *(c) = std::cos(x))
#else
#define ORACLE_SINCOS(x, s, c) pdmpc_sincos((x), (s), (c))
#endif
#ifdef ORACLE_HYPOT_NORM
#define ORACLE_NORM2(a, b) std::hypot((a), (b))
#else
#define ORACLE_NORM2(a, b) std::sqrt((a) * (a) + (b) * (b))
#endif

namespace {

const double kNaN = std::numeric_limits<double>::quiet_NaN();

// 2 x n matrix, MATLAB [x; y]
struct Poly {
    std::vector<double> x, y;
    size_t n() const { return x.size(); }
    bool empty() const { return x.empty(); }
    void push(double px, double py) {
        x.push_back(px);
        y.push_back(py);
    }
};

Poly make_poly(const double* x, const double* y, int n) {
    Poly p;
    p.x.assign(x, x + n);
    p.y.assign(y, y + n);
    return p;
}

Poly poly_from_set(const pdmpc_polygon_set& s, int i) {
    int a = s.offset[i], b = s.offset[i + 1];
    return make_poly(s.x + a, s.y + a, b - a);
}

// MATLAB min/max over a vector ignore NaN unless all entries are NaN.
inline double mmin(double a, double b) {
    if (std::isnan(a)) return b;
    if (std::isnan(b)) return a;
    return b < a ? b : a;
}
inline double mmax(double a, double b) {
    if (std::isnan(a)) return b;
    if (std::isnan(b)) return a;
    return b > a ? b : a;
}

// intersect_sat.m:17-42 (local function intersect_a_b)
bool intersect_a_b(const Poly& s1, const Poly& s2) {
    const size_t n1 = s1.n(), n2 = s2.n();
    bool any_d1 = false, any_d2 = false;
    for (size_t e = 0; e < n1; ++e) {
        // edge_vector = diff([shape1, shape1(:,1)], 1, 2)                       :19
        const size_t e2 = (e + 1 == n1) ? 0 : e + 1;
        const double ex = s1.x[e2] - s1.x[e];
        const double ey = s1.y[e2] - s1.y[e];
        // axis = [-edge_vector(2,:); edge_vector(1,:)]                           :21
        const double ax = -ey, ay = ex;
        // normed_axis = axis ./ vecnorm(axis)                                    :23
        const double nrm = ORACLE_NORM2(ax, ay);
        const double nx = ax / nrm, ny = ay / nrm;
        // dotprod1 = normed_axis' * shape1; min/max over columns                 :26-29
        double min1 = kNaN, max1 = kNaN, min2 = kNaN, max2 = kNaN;
        for (size_t v = 0; v < n1; ++v) {
            const double d = nx * s1.x[v] + ny * s1.y[v];
            min1 = v ? mmin(min1, d) : d;
            max1 = v ? mmax(max1, d) : d;
        }
        for (size_t v = 0; v < n2; ++v) {
            const double d = nx * s2.x[v] + ny * s2.y[v];
            min2 = v ? mmin(min2, d) : d;
            max2 = v ? mmax(max2, d) : d;
        }
        const double d1 = min1 - max2;  // :33
        const double d2 = min2 - max1;  // :34
        if (d1 > 0) any_d1 = true;      // NaN > 0 is false: a zero-length edge never separates
        if (d2 > 0) any_d2 = true;
    }
    // if any(d1 > 0) || any(d2 > 0) collide = false                               :36-40
    return !(any_d1 || any_d2);
}

// intersect_sat.m:1-15
bool intersect_sat(const Poly& s1, const Poly& s2) {
    bool collide = true;
    if (!intersect_a_b(s1, s2))
        collide = false;
    else if (!intersect_a_b(s2, s1))
        collide = false;
    return collide;
}

// hlc/optimizer/common/intersect_lanelet_boundary.m:1-56
bool intersect_lanelet_boundary(const Poly& shape, const Poly& left, const Poly& right) {
    // n_point = length(boundary_points): for a 2 x P matrix with P >= 2 that is P; [] gives 0   :8-9
    const size_t n_left = left.n(), n_right = right.n();
    double max_x = shape.x[0], min_x = shape.x[0], max_y = shape.y[0], min_y = shape.y[0];
    for (size_t v = 1; v < shape.n(); ++v) {  // :11-14
        max_x = mmax(max_x, shape.x[v]);
        min_x = mmin(min_x, shape.x[v]);
        max_y = mmax(max_y, shape.y[v]);
        min_y = mmin(min_y, shape.y[v]);
    }
    for (int side = 0; side < 2; ++side) {  // left loop :16-34, right loop :36-54
        const Poly& b = side == 0 ? left : right;
        const size_t np = side == 0 ? n_left : n_right;
        for (size_t n = 0; n + 1 < np; ++n) {
            const double x1 = b.x[n], x2 = b.x[n + 1], y1 = b.y[n], y2 = b.y[n + 1];
            // if all(max_x < seg_x) || all(min_x > seg_x) || all(max_y < seg_y) || all(min_y > seg_y) continue   :20,40
            if ((max_x < x1 && max_x < x2) || (min_x > x1 && min_x > x2) || (max_y < y1 && max_y < y2) ||
                (min_y > y1 && min_y > y2))
                continue;
            Poly seg;
            seg.push(x1, y1);
            seg.push(x2, y2);
            if (intersect_sat(shape, seg)) return true;  // :24,44
        }
    }
    return false;
}

// graph_search/intersect_lanelets.m:1-22 — lanelet rows are [rx ry lx ly cx cy] (LaneletInfo.m:5-10)
bool intersect_lanelets(const Poly& shape, const double* lanelet, int n_rows) {
    const int n_seg = n_rows - 1;
    for (int i = 0; i < n_seg; ++i) {
        Poly r, l;
        r.push(lanelet[i * 6 + 0], lanelet[i * 6 + 1]);
        r.push(lanelet[(i + 1) * 6 + 0], lanelet[(i + 1) * 6 + 1]);
        if (intersect_sat(shape, r)) return true;  // right bound :9
        l.push(lanelet[i * 6 + 2], lanelet[i * 6 + 3]);
        l.push(lanelet[(i + 1) * 6 + 2], lanelet[(i + 1) * 6 + 3]);
        if (intersect_sat(shape, l)) return true;  // left bound :15
    }
    return false;
}

// graph_search/InterX.m:48-103 with isReturnPoints = false
bool interx(const Poly& L1, const Poly& L2) {
    if (L2.empty() || L1.empty()) return false;  // :48-52
    const size_t n1 = L1.n(), n2 = L2.n();
    if (n1 < 2 || n2 < 2) return false;  // diff() of a single column is empty -> C1 & C2 empty
    const std::vector<double>&x1 = L1.x, &y1 = L1.y, &x2 = L2.x, &y2 = L2.y;
    // dx1 = diff(x1) ...; S1 = dx1 .* y1(1:end-1) - dy1 .* x1(1:end-1)           :65-70
    std::vector<double> dx1(n1 - 1), dy1(n1 - 1), S1(n1 - 1), dx2(n2 - 1), dy2(n2 - 1), S2(n2 - 1);
    for (size_t i = 0; i + 1 < n1; ++i) {
        dx1[i] = x1[i + 1] - x1[i];
        dy1[i] = y1[i + 1] - y1[i];
        S1[i] = dx1[i] * y1[i] - dy1[i] * x1[i];
    }
    for (size_t j = 0; j + 1 < n2; ++j) {
        dx2[j] = x2[j + 1] - x2[j];
        dy2[j] = y2[j + 1] - y2[j];
        S2[j] = dx2[j] * y2[j] - dy2[j] * x2[j];
    }
    for (size_t i = 0; i + 1 < n1; ++i) {
        for (size_t j = 0; j + 1 < n2; ++j) {
            // C1 = D(dx1 * y2 - dy1 * x2, S1) < 0, D(x,y) = (x(:,1:end-1) - y) .* (x(:,2:end) - y)   :72,108-110
            const double a0 = dx1[i] * y2[j] - dy1[i] * x2[j];
            const double a1 = dx1[i] * y2[j + 1] - dy1[i] * x2[j + 1];
            const bool c1 = (a0 - S1[i]) * (a1 - S1[i]) < 0;
            // C2 = (D((y1 * dx2 - x1 * dy2)', S2') < 0)'                                              :73
            const double b0 = y1[i] * dx2[j] - x1[i] * dy2[j];
            const double b1 = y1[i + 1] * dx2[j] - x1[i + 1] * dy2[j];
            const bool c2 = (b0 - S2[j]) * (b1 - S2[j]) < 0;
            if (c1 && c2) return true;  // [i, j] = find(C1 & C2) non-empty        :76,99
        }
    }
    return false;
}

// graph_search/vectorize_all_obstacles.m:1-66
struct VectorizedObstacles {
    std::vector<Poly> vehicle_obstacles;  // {1 x Hp}
    std::vector<Poly> hdv_obstacles;      // {1 x Hp}
    Poly lanelet_boundary;
};

void append_with_nan(Poly& dst, const Poly& src) {  // cellfun(@(c)[c, [nan; nan]], ...)
    dst.x.insert(dst.x.end(), src.x.begin(), src.x.end());
    dst.y.insert(dst.y.end(), src.y.begin(), src.y.end());
    dst.push(kNaN, kNaN);
}

VectorizedObstacles vectorize_all_obstacles(const pdmpc_vehicle_in& it, int Hp) {
    VectorizedObstacles v;
    v.vehicle_obstacles.resize(Hp);
    v.hdv_obstacles.resize(Hp);
    // lanelet_boundary = [left, NaN, right, NaN]                                   :27-30
    append_with_nan(v.lanelet_boundary, make_poly(it.left_x, it.left_y, it.n_left));
    append_with_nan(v.lanelet_boundary, make_poly(it.right_x, it.right_y, it.n_right));
    const int n_dyn = it.dynamic_obstacles.n_polygons / Hp;
    const int n_hdv = it.hdv_reachable_sets.n_polygons / Hp;
    for (int k = 0; k < Hp; ++k) {  // :36-62
        for (int i = 0; i < it.obstacles.n_polygons; ++i)
            append_with_nan(v.vehicle_obstacles[k], poly_from_set(it.obstacles, i));
        for (int i = 0; i < n_dyn; ++i)
            append_with_nan(v.vehicle_obstacles[k], poly_from_set(it.dynamic_obstacles, i * Hp + k));
        for (int i = 0; i < n_hdv; ++i)
            append_with_nan(v.hdv_obstacles[k], poly_from_set(it.hdv_reachable_sets, i * Hp + k));
    }
    return v;
}

// MPA tables in the layout of include/pdmpc.h
struct Mpa {
    int n = 0, Hp = 0;
    std::vector<uint8_t> transition;  // [k][i][j]
    std::vector<int32_t> mindex;      // [i][j]
    std::vector<pdmpc_maneuver> man;
    const pdmpc_maneuver& maneuver(int t1, int t2) const { return man[mindex[(t1 - 1) * n + (t2 - 1)]]; }
    bool allowed(int t1, int t2, int k_exp) const {
        return transition[((size_t)(k_exp - 1) * n + (t1 - 1)) * n + (t2 - 1)] != 0;
    }
};

Mpa load_mpa(const pdmpc_mpa* m) {
    Mpa a;
    a.n = m->n_trims;
    a.Hp = m->Hp;
    a.transition.assign(m->transition, m->transition + (size_t)m->Hp * m->n_trims * m->n_trims);
    a.mindex.assign(m->maneuver_index, m->maneuver_index + (size_t)m->n_trims * m->n_trims);
    a.man.assign(m->maneuvers, m->maneuvers + m->n_maneuvers);
    return a;
}

// graph_search/Tree.m:3-13 for a single vehicle (nVeh == 1 in every prioritized run)
struct Tree {
    std::vector<uint32_t> parent;
    std::vector<double> x, y, yaw, g, h;
    std::vector<int> trim, k;
    size_t size() const { return parent.size(); }
};

// priority_queue_interface_mex.cpp:19-31, verbatim type and comparator
typedef std::tuple<size_t, double> queue_entry;
struct queue_entry_comparator {
    inline bool operator()(const queue_entry& a, const queue_entry& b) { return (std::get<1>(a) > std::get<1>(b)); }
};
typedef std::priority_queue<queue_entry, std::vector<queue_entry>, queue_entry_comparator> prio_q;

struct SearchTrace {
    std::vector<int32_t> pops;
    Tree tree;
};

Poly rotate_translate(const double area[2][PDMPC_VMAX], int ncols, double c, double s, double px, double py) {
    Poly p;
    for (int v = 0; v < ncols; ++v) {
        // shape_x = c * area(1,:) - s * area(2,:) + pX;  shape_y = s * area(1,:) + c * area(2,:) + pY     GraphSearch.m:158-159
        const double sx = c * area[0][v] - s * area[1][v] + px;
        const double sy = s * area[0][v] + c * area[1][v] + py;
        p.push(sx, sy);
    }
    return p;
}

// GraphSearch.do_graph_search (GraphSearch.m:23-107) + eval_edge_exact (:111-196) + expand_node.m
void graph_search(const pdmpc_config& opt, const Mpa& mpa, const pdmpc_vehicle_in& it, pdmpc_vehicle_out& info,
                  SearchTrace* trace) {
    const int Hp = opt.Hp;
    const size_t max_nodes = opt.max_nodes > 0 ? (size_t)((opt.max_nodes + 1) & ~1) : (size_t)32768;
    std::memset(&info, 0, sizeof info);
    info.n_hp = Hp;
    for (int k = 0; k < PDMPC_HP_MAX; ++k)  // ControlResultsInfo.m:40: y_predicted = nan(3, Hp, nVeh)
        for (int c = 0; c < 3; ++c) info.y_predicted[k][c] = kNaN;

    Tree tree;  // Tree(x, y, yaw, trim, k, g, h)                                     GraphSearch.m:34-41
    tree.parent.push_back(0);
    tree.x.push_back(it.x0);
    tree.y.push_back(it.y0);
    tree.yaw.push_back(it.yaw0);
    tree.trim.push_back(it.trim0);
    tree.k.push_back(0);
    tree.g.push_back(0);
    tree.h.push_back(0);
    std::vector<Poly> shapes_tmp(1);  // shapes_tmp(:, id)

    prio_q pq;
    pq.push(queue_entry(1, 0.0));  // pq.push(1, 0)                                    :45-46

    VectorizedObstacles vo;  // set_up_constraints                                     :48-49
    if (opt.checker == PDMPC_CHECK_INTERX) vo = vectorize_all_obstacles(it, Hp);
    const Poly left = make_poly(it.left_x, it.left_y, it.n_left);
    const Poly right = make_poly(it.right_x, it.right_y, it.n_right);
    const int n_dyn = it.dynamic_obstacles.n_polygons / Hp;

    int n_popped = 0;
    while (true) {
        // cur_node_id = pq.pop(): empty queue returns -1                               :55, mex.cpp:87-93
        if (pq.empty()) {
            info.n_expanded = (int32_t)tree.size();  // :58
            info.status = PDMPC_EXHAUSTED;           // :59
            break;
        }
        const size_t cur = std::get<0>(pq.top());
        pq.pop();
        ++n_popped;
        if (trace) trace->pops.push_back((int32_t)cur);

        // ---- eval_edge_exact                                                         :111-196
        bool is_valid = true;
        Poly shape;
        const uint32_t par = tree.parent[cur - 1];
        if (par) {  // root has no parent: valid, empty shape                            :137-139
            const double pX = tree.x[par - 1], pY = tree.y[par - 1], pYaw = tree.yaw[par - 1];
            const int t1 = tree.trim[par - 1], t2 = tree.trim[cur - 1];
            const int cK = tree.k[cur - 1];
            const pdmpc_maneuver& m = mpa.maneuver(t1, t2);
            double c, s;
            ORACLE_SINCOS(pYaw, &s, &c);  // c = cos(pYaw); s = sin(pYaw)                   :155-156
            shape = rotate_translate(m.area, m.n_cols, c, s, pX, pY);  // :158-160
            Poly shape_wo = rotate_translate(m.area_without_offset, m.n_cols, c, s, pX, pY);  // :162-164
            Poly shape_bc = (cK == Hp) ? rotate_translate(m.area_large_offset, m.n_cols, c, s, pX, pY)  // :166-170
                                       : shape_wo;                                                      // :171-174
            const int i_step = cK;  // :176
            if (opt.checker == PDMPC_CHECK_SAT) {
                // are_constraints_satisfied_sat.m:13-66
                for (int i = 0; i < it.obstacles.n_polygons && is_valid; ++i)  // :15-22
                    if (intersect_sat(shape, poly_from_set(it.obstacles, i))) is_valid = false;
                for (int i = 0; i < n_dyn && is_valid; ++i)  // :24-35
                    if (intersect_sat(shape, poly_from_set(it.dynamic_obstacles, i * Hp + (i_step - 1)))) is_valid = false;
                // :37-44 other vehicles of the same node: i_vehicle == 1, loop empty
                if (is_valid && intersect_lanelet_boundary(shape_bc, left, right)) is_valid = false;  // :46-53
                // :55-66 guarded by ~any(hdv_adjacency): loop body unreachable, no-op
            } else {
                // are_constraints_satisfied_interx.m:13-37
                if (interx(shape, vo.vehicle_obstacles[i_step - 1])) is_valid = false;  // :17-21
                if (is_valid) {
                    const Poly& hdv = vo.hdv_obstacles[i_step - 1];
                    bool all_nan = true;  // ~all(all(isnan(hdv_obstacles{i_step}))); all([]) is true    :23
                    for (size_t q = 0; q < hdv.n(); ++q)
                        if (!(std::isnan(hdv.x[q]) && std::isnan(hdv.y[q]))) all_nan = false;
                    if (!all_nan && interx(shape, hdv)) is_valid = false;  // :25-31
                }
                if (is_valid && interx(shape_bc, vo.lanelet_boundary)) is_valid = false;  // :34-37
            }
        }
        if (!is_valid) continue;  // :75-77
        if (shapes_tmp.size() < cur) shapes_tmp.resize(cur);
        shapes_tmp[cur - 1] = shape;  // :79

        if (tree.k[cur - 1] == Hp) {  // :81-90
            std::vector<size_t> path;  // fliplr(path_to_root(tree, cur))               Tree.m:44-52
            for (size_t nd = cur;; nd = tree.parent[nd - 1]) {
                path.push_back(nd);
                if (nd == 1) break;
            }
            std::reverse(path.begin(), path.end());
            for (size_t i = 0; i < path.size(); ++i) {
                const size_t nd = path[i];
                info.tree_path[i] = (int32_t)nd;
                double* row = info.path_nodes[i];  // NodeInfo.m:5-13
                row[0] = tree.x[nd - 1];
                row[1] = tree.y[nd - 1];
                row[2] = tree.yaw[nd - 1];
                row[3] = tree.trim[nd - 1];
                row[4] = tree.g[nd - 1];
                row[5] = tree.h[nd - 1];
                row[6] = tree.k[nd - 1];
                row[7] = 1;
                if (i >= 1) {
                    info.y_predicted[i - 1][0] = tree.x[nd - 1];  // return_path_to.m:14-23
                    info.y_predicted[i - 1][1] = tree.y[nd - 1];
                    info.y_predicted[i - 1][2] = tree.yaw[nd - 1];
                    info.predicted_trims[i - 1] = tree.trim[nd - 1];  // GraphSearch.m:86
                    const Poly& sh = shapes_tmp[nd - 1];              // return_path_area.m:4-7
                    info.shape_cols[i - 1] = (int32_t)sh.n();
                    for (size_t v = 0; v < sh.n(); ++v) {
                        info.shapes[i - 1][0][v] = sh.x[v];
                        info.shapes[i - 1][1][v] = sh.y[v];
                    }
                }
            }
            info.status = PDMPC_OK;
            info.n_expanded = (int32_t)tree.size();  // :89
            break;
        }

        // ---- expand_node.m:1-91 (single vehicle: cartprod over one set is the set itself, :20-26)
        const double curX = tree.x[cur - 1], curY = tree.y[cur - 1], curYaw = tree.yaw[cur - 1];
        const int curTrim = tree.trim[cur - 1];
        const int curK = tree.k[cur - 1];
        const double curG = tree.g[cur - 1];
        const int k_exp = curK + 1;                 // :13
        const int time_steps_to_go = Hp - k_exp;    // :37
        std::vector<int> succ;                      // find(transition_matrix_single(curTrim, :, k_exp)) ascending   :18
        for (int j = 1; j <= mpa.n; ++j)
            if (mpa.allowed(curTrim, j, k_exp)) succ.push_back(j);
        if (tree.size() + succ.size() > max_nodes) {
            info.status = PDMPC_ARENA_OVERFLOW;  // backend capacity guard; the reference tree is unbounded
            info.n_expanded = (int32_t)tree.size();
            break;
        }
        std::vector<queue_entry> new_open;
        for (size_t ic = 0; ic < succ.size(); ++ic) {
            const int t2 = succ[ic];
            const pdmpc_maneuver& m = mpa.maneuver(curTrim, t2);
            double c, s;
            ORACLE_SINCOS(curYaw, &s, &c);                         // :50-51
            const double expX = c * m.dx - s * m.dy + curX;       // :53
            const double expY = s * m.dx + c * m.dy + curY;       // :54
            const double expYaw = curYaw + m.dyaw;                // :55
            double expG = curG;                                   // :34
            {
                // expG += norm([expX - ref(k_exp,1); expY - ref(k_exp,2)])^2               :61
                const double ddx = expX - it.ref_x[k_exp - 1], ddy = expY - it.ref_y[k_exp - 1];
                const double nrm = ORACLE_NORM2(ddx, ddy);
                expG = expG + nrm * nrm;
            }
            double expH = 0;          // :35
            double d_traveled_max = 0;  // :66
            for (int i_t = 1; i_t <= time_steps_to_go; ++i_t) {  // :68-73
                d_traveled_max = d_traveled_max + opt.dt_seconds * it.v_ref[k_exp + i_t - 1];
                const double ddx = expX - it.ref_x[k_exp + i_t - 1], ddy = expY - it.ref_y[k_exp + i_t - 1];
                const double nrm = ORACLE_NORM2(ddx, ddy);
                const double diff = nrm - d_traveled_max;
                const double m0 = (diff > 0) ? diff : 0.0;  // max(0, diff): NaN -> 0, -0 handled below
                expH = expH + m0 * m0;
            }
            // add_nodes                                                                   Tree.m:54-70
            tree.parent.push_back((uint32_t)cur);
            tree.x.push_back(expX);
            tree.y.push_back(expY);
            tree.yaw.push_back(expYaw);
            tree.trim.push_back(t2);
            tree.k.push_back(k_exp);
            tree.g.push_back(expG);
            tree.h.push_back(expH);
            // new_open_values = g * 1 + h * 1                                              GraphSearch.m:100-102
            new_open.push_back(queue_entry(tree.size(), expG * 1 + expH * 1));
        }
        for (const queue_entry& e : new_open) pq.push(e);  // PUSH loop, mex.cpp:67-72
    }
    info.n_popped = n_popped;
    if (trace) trace->tree = std::move(tree);
}


// ---------------------------------------------------------------------------------------------------------------------
// The sampled optimizer: graph_search/MonteCarloTreeSearch.m (OptimizerType.MatlabSampled, OptimizerInterface.m:29-31).
//
// Random numbers: RandStream('mt19937ar', Seed = time_step + vehicle_index) (:32), rand(stream, 1, Hp * n_expansions_max)
// (:53).  mt19937ar is Matsumoto & Nishimura's reference generator; MATLAB's rand draws 53-bit doubles from two 32-bit
// outputs (genrand_res53).  Restated here from the published algorithm (parity unpinned against MATLAB itself; pinned
// against numpy's RandomState, which implements the same two routines, in tests/test_oracle_golden.py).
struct Mt19937 {
    uint32_t mt[624];
    int mti;
    explicit Mt19937(uint32_t seed) {  // init_genrand
        mt[0] = seed;
        for (mti = 1; mti < 624; ++mti) mt[mti] = 1812433253u * (mt[mti - 1] ^ (mt[mti - 1] >> 30)) + (uint32_t)mti;
    }
    uint32_t next_u32() {  // genrand_int32
        if (mti >= 624) {
            for (int kk = 0; kk < 624; ++kk) {
                const uint32_t y = (mt[kk] & 0x80000000u) | (mt[(kk + 1) % 624] & 0x7fffffffu);
                mt[kk] = mt[(kk + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            mti = 0;
        }
        uint32_t y = mt[mti++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    double next_double() {  // genrand_res53
        const uint32_t a = next_u32() >> 5, b = next_u32() >> 6;
        return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
    }
};

// are_constraints_satisfied_sat.m:13-66 / are_constraints_satisfied_interx.m:13-37 for one vehicle (the same statements
// graph_search evaluates inline above)
bool constraints_satisfied(const pdmpc_config& opt, const pdmpc_vehicle_in& it, const VectorizedObstacles& vo, const Poly& left, const Poly& right,
                           const Poly& shape, const Poly& shape_bc, int i_step) {
    const int Hp = opt.Hp;
    if (opt.checker == PDMPC_CHECK_SAT) {
        const int n_dyn = it.dynamic_obstacles.n_polygons / Hp;
        for (int i = 0; i < it.obstacles.n_polygons; ++i)
            if (intersect_sat(shape, poly_from_set(it.obstacles, i))) return false;
        for (int i = 0; i < n_dyn; ++i)
            if (intersect_sat(shape, poly_from_set(it.dynamic_obstacles, i * Hp + (i_step - 1)))) return false;
        if (intersect_lanelet_boundary(shape_bc, left, right)) return false;
        return true;
    }
    if (interx(shape, vo.vehicle_obstacles[i_step - 1])) return false;
    const Poly& hdv = vo.hdv_obstacles[i_step - 1];
    bool all_nan = true;
    for (size_t q = 0; q < hdv.n(); ++q)
        if (!(std::isnan(hdv.x[q]) && std::isnan(hdv.y[q]))) all_nan = false;
    if (!all_nan && interx(shape, hdv)) return false;
    if (interx(shape_bc, vo.lanelet_boundary)) return false;
    return true;
}

// transform * maneuver.dpose added to the pose (MonteCarloTreeSearch.m:131-138): a 3x3 matrix-vector product, each row
// accumulated left to right, zeros included
void apply_maneuver(double pose[3], const pdmpc_maneuver& m, double c, double s) {
    const double t0 = c * m.dx + (-s) * m.dy + 0.0 * m.dyaw;
    const double t1 = s * m.dx + c * m.dy + 0.0 * m.dyaw;
    const double t2 = 0.0 * m.dx + 0.0 * m.dy + 1.0 * m.dyaw;
    pose[0] = pose[0] + t0;
    pose[1] = pose[1] + t1;
    pose[2] = pose[2] + t2;
}

// transform(1:2,1:2) * area + start_pose(1:2)                                          MonteCarloTreeSearch.m:159-166
Poly transform_area(const double area[2][PDMPC_VMAX], int ncols, double c, double s, double px, double py) {
    Poly p;
    for (int v = 0; v < ncols; ++v) p.push((c * area[0][v] + (-s) * area[1][v]) + px, (s * area[0][v] + c * area[1][v]) + py);
    return p;
}

// MonteCarloTreeSearch.do_graph_search (:40-249).  `seed` = time_step + vehicle_index (:32).
void monte_carlo_tree_search(const pdmpc_config& opt, const Mpa& mpa, const pdmpc_vehicle_in& it, uint32_t seed, pdmpc_vehicle_out& info) {
    const int Hp = opt.Hp;
    const int n_expansions_max = 250;  // :8 (config/mcts.json, which could override it, does not exist in the reference tree)
    std::memset(&info, 0, sizeof info);
    info.n_hp = Hp;
    for (int k = 0; k < PDMPC_HP_MAX; ++k)
        for (int c = 0; c < 3; ++c) info.y_predicted[k][c] = kNaN;

    Mt19937 rng(seed);
    std::vector<double> random_numbers((size_t)Hp * n_expansions_max);  // :53
    for (double& r : random_numbers) r = rng.next_double();

    auto successor_trims = [&](int trim, int step) {  // mpa.successor_trims{trim, step}: find(transition_matrix_single(trim, :, step))
        std::vector<int> v;
        for (int j = 1; j <= mpa.n; ++j)
            if (mpa.allowed(trim, j, step)) v.push_back(j);
        return v;
    };
    int n_successor_trims_max = 0;  // mpa.maximum_branching_factor()               MotionPrimitiveAutomaton.m:689-692
    for (int k = 1; k <= mpa.Hp; ++k)
        for (int i = 1; i <= mpa.n; ++i) n_successor_trims_max = std::max(n_successor_trims_max, (int)successor_trims(i, k).size());

    // :60-74
    const size_t cap = (size_t)n_expansions_max + 1;  // (the reference's arrays grow when n_nodes passes n_expansions_max)
    std::vector<int> trims(cap + 1, 0);
    std::vector<uint32_t> parents(cap + 1, 0);
    std::vector<std::vector<uint32_t>> children(cap + 1, std::vector<uint32_t>((size_t)n_successor_trims_max, 0));  // children(:, node)
    const double root_pose[3] = {it.x0, it.y0, it.yaw0};
    trims[1] = it.trim0;
    {
        const std::vector<int> rs = successor_trims(it.trim0, 1);
        for (size_t q = 0; q < rs.size(); ++q) children[1][q] = 1;
    }
    int n_nodes = 1;
    bool have_best = false;  // valid_nodes_at_hp: only its top is ever read (:197); with the comparator a.key > b.key the first
    double best_cost = 0;    // entry pushed among those of minimal key stays on top (std::push_heap moves up on strict > only)
    uint32_t best_node = 0;
    std::vector<Poly> shapes_tmp(cap + 1);

    VectorizedObstacles vo;  // :78-79
    if (opt.checker == PDMPC_CHECK_INTERX) vo = vectorize_all_obstacles(it, Hp);
    const Poly left = make_poly(it.left_x, it.left_y, it.n_left);
    const Poly right = make_poly(it.right_x, it.right_y, it.n_right);

    int n_expansions = 0, n_traversals = 0;
    bool is_finished = false;
    while (n_expansions < n_expansions_max && !is_finished) {  // :89
        uint32_t node_id = 1;
        double solution_cost = 0;
        double node_pose[3] = {root_pose[0], root_pose[1], root_pose[2]};
        bool is_valid = false;
        size_t child_position = 0;  // 0-based here
        uint32_t node_parent = 0;
        for (int i_step = 1; i_step <= Hp; ++i_step) {  // :95
            is_valid = false;
            ++n_traversals;
            std::vector<size_t> trim_positions;  // find(children(:, node_id))              :100
            for (size_t q = 0; q < children[node_id].size(); ++q)
                if (children[node_id][q] != 0) trim_positions.push_back(q);
            const size_t n_trims = trim_positions.size();
            if (n_trims != 0) {
                // child_position = trim_positions(ceil(random_numbers(n_traversals) * n_trims))        :105
                const double pick = std::ceil(random_numbers[(size_t)n_traversals - 1] * (double)n_trims);
                child_position = trim_positions[(size_t)pick - 1];
            } else {
                if (node_id != 1) {  // remove edge to node without children                      :108-112
                    const uint32_t parent_id = parents[node_id];
                    for (uint32_t& c : children[parent_id])
                        if (c == node_id) c = 0;
                    break;
                }
                is_finished = true;  // :114-115
                break;
            }
            // expand                                                                        :120-138
            const int parent_trim = trims[node_id];
            const std::vector<int> succ = successor_trims(parent_trim, i_step);
            const int goal_trim = succ[child_position];
            const pdmpc_maneuver& m = mpa.maneuver(parent_trim, goal_trim);
            double c, s;
            ORACLE_SINCOS(node_pose[2], &s, &c);
            const double start_pose[3] = {node_pose[0], node_pose[1], node_pose[2]};
            apply_maneuver(node_pose, m, c, s);
            {  // solution_cost += norm(node_pose(1:2) - reference_trajectory_points(:, i_step))^2       :143
                const double ddx = node_pose[0] - it.ref_x[i_step - 1], ddy = node_pose[1] - it.ref_y[i_step - 1];
                const double nrm = ORACLE_NORM2(ddx, ddy);
                solution_cost = solution_cost + nrm * nrm;
            }
            const bool is_expanded = children[node_id][child_position] != 1;  // :145
            if (is_expanded) {
                node_id = children[node_id][child_position];
                continue;
            }
            ++n_expansions;  // :152
            node_parent = node_id;
            const Poly shape_wo = transform_area(m.area_without_offset, m.n_cols, c, s, start_pose[0], start_pose[1]);
            const Poly shape = transform_area(m.area, m.n_cols, c, s, start_pose[0], start_pose[1]);
            Poly shape_bc;
            std::vector<int> child_successor_trims;
            if (i_step != Hp) {  // :162-168
                shape_bc = shape_wo;
                child_successor_trims = successor_trims(goal_trim, i_step + 1);
            } else {
                shape_bc = transform_area(m.area_large_offset, m.n_cols, c, s, start_pose[0], start_pose[1]);
            }
            is_valid = constraints_satisfied(opt, it, vo, left, right, shape, shape_bc, i_step);  // :170-179
            if (!is_valid) {
                children[node_parent][child_position] = 0;  // :183
                break;
            }
            ++n_nodes;  // :186-193
            if ((size_t)n_nodes >= trims.size()) {
                trims.resize(n_nodes + 1, 0);
                parents.resize(n_nodes + 1, 0);
                children.resize(n_nodes + 1, std::vector<uint32_t>((size_t)n_successor_trims_max, 0));
                shapes_tmp.resize(n_nodes + 1);
            }
            parents[n_nodes] = node_parent;
            trims[n_nodes] = goal_trim;
            for (size_t q = 0; q < child_successor_trims.size(); ++q) children[n_nodes][q] = 1;
            children[node_parent][child_position] = (uint32_t)n_nodes;
            shapes_tmp[n_nodes] = shape;
            node_id = (uint32_t)n_nodes;
        }
        if (is_valid) {  // :199-203
            if (!have_best || solution_cost < best_cost) {
                have_best = true;
                best_cost = solution_cost;
                best_node = node_id;
            }
            children[node_parent][child_position] = 0;  // avoid double exploration
        }
    }
    info.n_expanded = n_expansions;  // :209
    info.n_popped = n_traversals;    // (not a reference output: the number of tree descents steps, for the statistics)
    if (!have_best) {  // :212-215
        info.status = PDMPC_EXHAUSTED;
        return;
    }
    // final path                                                                            :217-248
    std::vector<uint32_t> path;
    for (uint32_t nd = best_node;; nd = parents[nd]) {
        path.push_back(nd);
        if (nd == 1) break;
    }
    std::reverse(path.begin(), path.end());
    double pose[3] = {root_pose[0], root_pose[1], root_pose[2]};
    for (size_t i = 0; i < path.size(); ++i) {
        const uint32_t nd = path[i];
        if (i >= 1) {
            const pdmpc_maneuver& m = mpa.maneuver(trims[path[i - 1]], trims[nd]);
            double c, s;
            ORACLE_SINCOS(pose[2], &s, &c);
            apply_maneuver(pose, m, c, s);
        }
        info.tree_path[i] = (int32_t)nd;
        double* row = info.path_nodes[i];  // NodeInfo order; the reference leaves g/h = -1 except g of the chosen node (:223-225, 244)
        row[0] = pose[0];
        row[1] = pose[1];
        row[2] = pose[2];
        row[3] = trims[nd];
        row[4] = (nd == best_node) ? best_cost : -1.0;
        row[5] = -1.0;
        row[6] = (double)(i + 1);  // tree.k(final_nodes) = 1:length(final_nodes)                 :242
        row[7] = 1;
        if (i >= 1) {
            info.y_predicted[i - 1][0] = pose[0];
            info.y_predicted[i - 1][1] = pose[1];
            info.y_predicted[i - 1][2] = pose[2];
            info.predicted_trims[i - 1] = trims[nd];
            const Poly& sh = shapes_tmp[nd];
            info.shape_cols[i - 1] = (int32_t)sh.n();
            for (size_t v = 0; v < sh.n(); ++v) {
                info.shapes[i - 1][0][v] = sh.x[v];
                info.shapes[i - 1][1][v] = sh.y[v];
            }
        }
    }
    info.status = PDMPC_OK;
}

}  // namespace

extern "C" {

// ---- known-answer entry points (tests/test_oracle_golden.py) ----
int oracle_intersect_sat(const double* x1, const double* y1, int n1, const double* x2, const double* y2, int n2) {
    return intersect_sat(make_poly(x1, y1, n1), make_poly(x2, y2, n2)) ? 1 : 0;
}

int oracle_intersect_lanelet_boundary(const double* sx, const double* sy, int n, const double* lx, const double* ly,
                                      int nl, const double* rx, const double* ry, int nr) {
    return intersect_lanelet_boundary(make_poly(sx, sy, n), make_poly(lx, ly, nl), make_poly(rx, ry, nr)) ? 1 : 0;
}

int oracle_intersect_lanelets(const double* sx, const double* sy, int n, const double* lanelet_rows, int n_rows) {
    return intersect_lanelets(make_poly(sx, sy, n), lanelet_rows, n_rows) ? 1 : 0;
}

int oracle_interx(const double* x1, const double* y1, int n1, const double* x2, const double* y2, int n2) {
    return interx(make_poly(x1, y1, n1), make_poly(x2, y2, n2)) ? 1 : 0;
}

// The reference queue driven by a command script: op[i] == 0 pushes (id[i], key[i]); op[i] == 1 pops
// and appends the popped id (or -1 on empty, mex.cpp:87-93) to out.  Returns the number of pops.
int oracle_pq_script(const int32_t* op, const int32_t* id, const double* key, int n, int32_t* out) {
    prio_q pq;
    int n_out = 0;
    for (int i = 0; i < n; ++i) {
        if (op[i] == 0) {
            pq.push(queue_entry((size_t)id[i], key[i]));
        } else {
            if (pq.empty()) {
                out[n_out++] = -1;
            } else {
                out[n_out++] = (int32_t)std::get<0>(pq.top());
                pq.pop();
            }
        }
    }
    return n_out;
}

void oracle_sincos(const double* x, int n, double* s, double* c) {
    for (int i = 0; i < n; ++i) pdmpc_sincos(x[i], &s[i], &c[i]);
}

// ---- the optimizer ----
struct oracle_trace_out {
    int32_t pop_capacity;
    int32_t n_pops;
    int32_t* pops;
    int32_t tree_capacity;
    int32_t n_nodes;
    double *x, *y, *yaw, *g, *h;
    int32_t *trim, *k, *parent;
};

// Plans n independent vehicles with `n_threads` host threads (1 = the scalar port).  trace may be NULL,
// otherwise trace[i] receives vehicle i's pop sequence and tree (truncated to the given capacities).
// Returns 0, or -1 on invalid arguments.  *elapsed_ms (may be NULL) gets the wall time of the planning
// loop only (steady_clock), which is what bench.py reports as cpu_baseline.
int oracle_plan_batch(const pdmpc_config* cfg, const pdmpc_mpa* mpa_in, int n, const pdmpc_vehicle_in* in,
                      pdmpc_vehicle_out* out, oracle_trace_out* trace, int n_threads, double* elapsed_ms) {
    if (!cfg || !mpa_in || n < 0 || (n > 0 && (!in || !out))) return -1;
    if (cfg->Hp < 1 || cfg->Hp > PDMPC_HP_MAX || mpa_in->Hp < cfg->Hp) return -1;
    const Mpa mpa = load_mpa(mpa_in);
    std::atomic<int> next(0);
    auto worker = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) break;
            SearchTrace tr;
            graph_search(*cfg, mpa, in[i], out[i], trace ? &tr : nullptr);
            if (trace) {
                oracle_trace_out& t = trace[i];
                t.n_pops = (int32_t)tr.pops.size();
                for (int q = 0; q < t.n_pops && q < t.pop_capacity; ++q) t.pops[q] = tr.pops[q];
                t.n_nodes = (int32_t)tr.tree.size();
                for (int q = 0; q < t.n_nodes && q < t.tree_capacity; ++q) {
                    t.x[q] = tr.tree.x[q];
                    t.y[q] = tr.tree.y[q];
                    t.yaw[q] = tr.tree.yaw[q];
                    t.g[q] = tr.tree.g[q];
                    t.h[q] = tr.tree.h[q];
                    t.trim[q] = tr.tree.trim[q];
                    t.k[q] = tr.tree.k[q];
                    t.parent[q] = (int32_t)tr.tree.parent[q];
                }
            }
        }
    };
    const auto t0 = std::chrono::steady_clock::now();
    if (n_threads <= 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(worker);
        for (auto& th : pool) th.join();
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (elapsed_ms) *elapsed_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    return 0;
}


// n doubles of the mt19937ar stream seeded with `seed`, as MATLAB's rand(stream, 1, n) draws them
void oracle_mt19937_doubles(uint32_t seed, int n, double* out) {
    Mt19937 rng(seed);
    for (int i = 0; i < n; ++i) out[i] = rng.next_double();
}

// The sampled optimizer (MonteCarloTreeSearch.m) for n independent vehicles; seeds[i] = time_step + vehicle_index.
int oracle_plan_batch_sampled(const pdmpc_config* cfg, const pdmpc_mpa* mpa_in, int n, const pdmpc_vehicle_in* in, const uint32_t* seeds,
                              pdmpc_vehicle_out* out, int n_threads, double* elapsed_ms) {
    if (!cfg || !mpa_in || n < 0 || (n > 0 && (!in || !out || !seeds))) return -1;
    if (cfg->Hp < 1 || cfg->Hp > PDMPC_HP_MAX || mpa_in->Hp < cfg->Hp) return -1;
    const Mpa mpa = load_mpa(mpa_in);
    std::atomic<int> next(0);
    auto worker = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) break;
            monte_carlo_tree_search(*cfg, mpa, in[i], seeds[i], out[i]);
        }
    };
    const auto t0 = std::chrono::steady_clock::now();
    if (n_threads <= 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(worker);
        for (auto& th : pool) th.join();
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (elapsed_ms) *elapsed_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// One whole time step on the host: the level loop of PrioritizedSequentialController.m:77-94 with the hand-over of
// PrioritizedController.m:476-491 (a vehicle's dynamic obstacles = its own + info.shapes(1, :) of every sequential
// predecessor, read from the records planned in earlier levels) and the published fallback areas of exhausted vehicles
// (:568-616, 678-718).  Arguments as for pdmpc_plan_step (include/pdmpc.h); slots are in level order and level_sizes[l]
// vehicles form level l.  All vehicles of a level plan concurrently on min(level size, n_threads) threads of a pool that
// lives across levels and calls -- the stand-in for ComputationMode.parallel_threads (main.m:43-60), with no thread
// start-up inside the timed region.  *elapsed_ms = wall time of the whole step (hand-over included, as the reference's
// `plan` timer includes it, PrioritizedController.m:288-290); *threads_mean = time-weighted mean of the threads that
// were actually busy (level times x min(level size, n_threads) / step time).
namespace {
class WorkerPool {
public:
    explicit WorkerPool(int n) {
        for (int t = 0; t < n; ++t) threads_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    int size() const { return (int)threads_.size(); }
    // runs job(i) for i in [0, n) on at most `width` workers and waits
    void run(int n, int width, const std::function<void(int)>& job) {
        if (n <= 0) return;
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = &job;
            n_ = n;
            next_ = 0;
            pending_ = n;
            width_ = std::min(width, (int)threads_.size());
            active_ = 0;
            gen_ += 1;
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_ == 0; });
        job_ = nullptr;
    }

private:
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return stop_ || (gen_ != seen && job_ && next_ < n_ && active_ < width_); });
            if (stop_) return;
            seen = gen_;
            active_ += 1;
            while (next_ < n_) {
                const int i = next_++;
                const std::function<void(int)>* job = job_;
                lk.unlock();
                (*job)(i);
                lk.lock();
                if (--pending_ == 0) done_.notify_all();
            }
            active_ -= 1;
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* job_ = nullptr;
    int n_ = 0, next_ = 0, pending_ = 0, width_ = 0, active_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

WorkerPool& pool_of(int n_threads) {
    static std::mutex m;
    static std::unique_ptr<WorkerPool> pool;
    std::lock_guard<std::mutex> g(m);
    if (!pool || pool->size() < n_threads) pool.reset(new WorkerPool(n_threads));
    return *pool;
}
}  // namespace

extern "C" int oracle_plan_step(const pdmpc_config* cfg, const pdmpc_mpa* mpa_in, int n, const pdmpc_vehicle_in* in, const int32_t* pred_offset,
                                const int32_t* pred_index, const pdmpc_polygon_set* fallback, int n_levels, const int32_t* level_sizes,
                                pdmpc_vehicle_out* out, int n_threads, double* elapsed_ms, double* threads_mean) {
    if (!cfg || !mpa_in || n < 0 || (n > 0 && (!in || !out || !pred_offset || !level_sizes))) return -1;
    if (cfg->Hp < 1 || cfg->Hp > PDMPC_HP_MAX || mpa_in->Hp < cfg->Hp) return -1;
    const int Hp = cfg->Hp;
    const Mpa mpa = load_mpa(mpa_in);
    int total = 0;
    for (int l = 0; l < n_levels; ++l) total += level_sizes[l];
    if (total != n) return -1;
    if (n_threads < 1) n_threads = 1;
    WorkerPool* pool = n_threads > 1 ? &pool_of(n_threads) : nullptr;
    struct Scratch {  // the vehicle's dynamic obstacles with its predecessors' areas appended (PrioritizedController.m:476-491)
        std::vector<int32_t> off;
        std::vector<double> x, y;
    };
    std::vector<Scratch> scratch((size_t)n);
    double busy = 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    int first = 0;
    for (int l = 0; l < n_levels; ++l) {
        const int size = level_sizes[l];
        const auto tl0 = std::chrono::steady_clock::now();
        std::function<void(int)> job = [&](int q) {
            const int s = first + q;
            pdmpc_vehicle_in v = in[s];
            const int np = pred_offset[s + 1] - pred_offset[s];
            if (np > 0) {
                Scratch& sc = scratch[(size_t)s];
                const pdmpc_polygon_set& d = in[s].dynamic_obstacles;
                const int nd = d.n_polygons / Hp;
                sc.off.assign(1, 0);
                sc.x.clear();
                sc.y.clear();
                auto push_poly = [&](const double* px, const double* py, int cnt) {
                    for (int c = 0; c < cnt; ++c) {
                        sc.x.push_back(px[c]);
                        sc.y.push_back(py[c]);
                    }
                    sc.off.push_back((int32_t)sc.x.size());
                };
                for (int r = 0; r < nd; ++r)
                    for (int k = 0; k < Hp; ++k) {
                        const int p = r * Hp + k;
                        push_poly(d.x + d.offset[p], d.y + d.offset[p], d.offset[p + 1] - d.offset[p]);
                    }
                int rows = nd;
                for (int e = pred_offset[s]; e < pred_offset[s + 1]; ++e) {
                    const int ps = pred_index[e];
                    const pdmpc_vehicle_out& po = out[ps];
                    if (po.status == PDMPC_OK) {
                        for (int k = 0; k < Hp; ++k) push_poly(po.shapes[k][0], po.shapes[k][1], po.shape_cols[k]);
                        rows += 1;
                    } else if (fallback && fallback[ps].n_polygons == Hp) {
                        const pdmpc_polygon_set& fb = fallback[ps];
                        for (int k = 0; k < Hp; ++k) push_poly(fb.x + fb.offset[k], fb.y + fb.offset[k], fb.offset[k + 1] - fb.offset[k]);
                        rows += 1;
                    }
                }
                static const double zero = 0.0;
                v.dynamic_obstacles.n_polygons = rows * Hp;
                v.dynamic_obstacles.offset = sc.off.data();
                v.dynamic_obstacles.x = sc.x.empty() ? &zero : sc.x.data();
                v.dynamic_obstacles.y = sc.y.empty() ? &zero : sc.y.data();
            }
            graph_search(*cfg, mpa, v, out[s], nullptr);
            // the device publishes the fallback areas of an exhausted vehicle in its record: so does this loop
            if (out[s].status != PDMPC_OK && fallback && fallback[s].n_polygons == Hp) {
                const pdmpc_polygon_set& fb = fallback[s];
                for (int k = 0; k < Hp; ++k) {
                    const int cnt = std::min(fb.offset[k + 1] - fb.offset[k], (int32_t)PDMPC_VMAX);
                    out[s].shape_cols[k] = cnt;
                    for (int c = 0; c < cnt; ++c) {
                        out[s].shapes[k][0][c] = fb.x[fb.offset[k] + c];
                        out[s].shapes[k][1][c] = fb.y[fb.offset[k] + c];
                    }
                }
            }
        };
        const int width = std::min(size, n_threads);
        if (pool && width > 1)
            pool->run(size, width, job);
        else
            for (int q = 0; q < size; ++q) job(q);
        const auto tl1 = std::chrono::steady_clock::now();
        busy += std::chrono::duration<double, std::milli>(tl1 - tl0).count() * (double)std::max(width, 1);
        first += size;
    }
    const auto t1 = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (elapsed_ms) *elapsed_ms = ms;
    if (threads_mean) *threads_mean = ms > 0.0 ? busy / ms : 1.0;
    return 0;
}
