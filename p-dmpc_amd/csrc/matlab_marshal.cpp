// matlab_marshal.cpp — the MATLAB-shaped entry points declared in include/pdmpc_matlab.h.  Host code only, no device
// code and no mex.hpp: column-major doubles and cell arrays in linear order come in, the plain structs of include/pdmpc.h go
// to pdmpc_upload_mpa / pdmpc_plan_batch / pdmpc_plan_step.  p-dmpc_amd/matlab/pdmpc_mex.cpp only turns matlab::data arrays
// into the descriptors used here, so everything that can go wrong with an index (cell order of the n_d x Hp obstacle cell,
// the n x n x Hp transition matrix, the 2 x V polygons) is in this file and under tests/test_matlab_marshal.py.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pdmpc_matlab.h"

namespace {

thread_local std::string g_ml_err;

int ml_fail(int code, const std::string& msg) {
    g_ml_err = msg;
    return code;
}

// owns the flattened copy a pdmpc_polygon_set points into
struct PolyStore {
    std::vector<int32_t> off{0};
    std::vector<double> x, y;
    bool add(const pdmpc_ml_matrix& m) {  // 2 x V column-major: [x0 y0 x1 y1 ...]
        const int64_t cnt = (int64_t)m.rows * m.cols;
        if (cnt != 0 && (m.rows != 2 || !m.data)) return false;
        for (int v = 0; v < (cnt ? m.cols : 0); ++v) {
            x.push_back(m.data[2 * (size_t)v]);
            y.push_back(m.data[2 * (size_t)v + 1]);
        }
        off.push_back((int32_t)x.size());
        return true;
    }
    pdmpc_polygon_set view() const {
        static const double zero = 0.0;
        return {(int32_t)off.size() - 1, off.data(), x.empty() ? &zero : x.data(), y.empty() ? &zero : y.data()};
    }
};

// R x C cell (linear order i + k * R) -> polygons in the ABI's order i * C + k (row-major: obstacle-major, step-minor)
bool add_cell_rows(PolyStore& dst, const pdmpc_ml_matrix* cells, int rows, int cols) {
    for (int i = 0; i < rows; ++i)
        for (int k = 0; k < cols; ++k)
            if (!dst.add(cells[(size_t)i + (size_t)k * rows])) return false;
    return true;
}

struct VehicleStore {
    std::vector<double> ref_x, ref_y, v_ref, lx, ly, rx, ry;
    PolyStore stat, dyn, hdv, fb;
};

}  // namespace

struct pdmpc_ml_mpa {
    std::vector<uint8_t> transition;
    std::vector<int32_t> index;
    std::vector<pdmpc_maneuver> maneuvers;
    pdmpc_mpa view{};
};

struct pdmpc_ml_step {
    int32_t n = 0, Hp = 0;
    std::vector<VehicleStore> store;          // per slot
    std::vector<pdmpc_vehicle_in> in;         // per slot
    std::vector<pdmpc_polygon_set> fallback;  // per slot
    std::vector<int32_t> pred_offset, pred_index, order, levels;
    bool any_fallback = false;
};

extern "C" {

const char* pdmpc_ml_last_error(void) { return g_ml_err.c_str(); }

int pdmpc_ml_mpa_create(const double* T, int32_t n, int32_t Hp, const pdmpc_ml_maneuver* man, pdmpc_ml_mpa** out) {
    if (!T || !man || !out || n < 1 || Hp < 1) return ml_fail(PDMPC_ERR_INVALID, "pdmpc_ml_mpa_create: bad argument");
    pdmpc_ml_mpa* m = new pdmpc_ml_mpa();
    m->transition.assign((size_t)Hp * n * n, 0);
    m->index.assign((size_t)n * n, -1);
    // transition[k][i][j] = transition_matrix_single(i + 1, j + 1, k + 1), MATLAB element (i, j, k) at i + j n + k n n
    for (int k = 0; k < Hp; ++k)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) m->transition[((size_t)k * n + i) * n + j] = T[(size_t)i + (size_t)j * n + (size_t)k * n * n] != 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const pdmpc_ml_maneuver& c = man[(size_t)i + (size_t)j * n];  // cell (i, j)
            if (!c.present) continue;
            pdmpc_maneuver q{};
            q.dx = c.dx;
            q.dy = c.dy;
            q.dyaw = c.dyaw;
            const pdmpc_ml_matrix* src[3] = {&c.area, &c.area_without_offset, &c.area_large_offset};
            double(*dst[3])[PDMPC_VMAX] = {q.area, q.area_without_offset, q.area_large_offset};
            for (int a = 0; a < 3; ++a) {
                if (src[a]->rows != 2 || src[a]->cols < 2 || src[a]->cols > PDMPC_VMAX || !src[a]->data || (a > 0 && src[a]->cols != q.n_cols)) {
                    delete m;
                    return ml_fail(PDMPC_ERR_INVALID, "pdmpc_ml_mpa_create: a maneuver area must be 2 x V with 2 <= V <= PDMPC_VMAX, the same V for all three areas");
                }
                q.n_cols = src[a]->cols;
                for (int v = 0; v < q.n_cols; ++v) {
                    dst[a][0][v] = src[a]->data[2 * (size_t)v];
                    dst[a][1][v] = src[a]->data[2 * (size_t)v + 1];
                }
            }
            m->index[(size_t)i * n + j] = (int32_t)m->maneuvers.size();
            m->maneuvers.push_back(q);
        }
    m->view = pdmpc_mpa{n, Hp, m->transition.data(), m->index.data(), (int32_t)m->maneuvers.size(), m->maneuvers.data()};
    *out = m;
    return PDMPC_OK;
}

const pdmpc_mpa* pdmpc_ml_mpa_view(const pdmpc_ml_mpa* m) { return m ? &m->view : nullptr; }

void pdmpc_ml_mpa_destroy(pdmpc_ml_mpa* m) { delete m; }

int pdmpc_ml_upload_mpa(pdmpc_handle* h, const double* T, int32_t n, int32_t Hp, const pdmpc_ml_maneuver* man) {
    pdmpc_ml_mpa* m = nullptr;
    int rc = pdmpc_ml_mpa_create(T, n, Hp, man, &m);
    if (rc) return rc;
    rc = pdmpc_upload_mpa(h, &m->view);
    if (rc) g_ml_err = pdmpc_last_error();
    pdmpc_ml_mpa_destroy(m);
    return rc;
}

int pdmpc_ml_step_create(int32_t Hp, int32_t n, const pdmpc_ml_iter* iters, const double* seq, const pdmpc_ml_matrix* fallback, pdmpc_ml_step** out) {
    if (!out || n < 0 || (n > 0 && !iters) || Hp < 1 || Hp > PDMPC_HP_MAX) return ml_fail(PDMPC_ERR_INVALID, "pdmpc_ml_step_create: bad argument");
    pdmpc_ml_step* s = new pdmpc_ml_step();
    s->n = n;
    s->Hp = Hp;
    // utility/kahn.m:1-24 on directed_coupling_sequential: level of a vertex = 1 + the highest level among its predecessors
    std::vector<int32_t> level((size_t)n, 0), indeg((size_t)n, 0);
    if (seq)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                if (i != j && seq[(size_t)i + (size_t)j * n] != 0.0) indeg[(size_t)j] += 1;
    {
        std::vector<int32_t> frontier, next;
        for (int v = 0; v < n; ++v)
            if (indeg[(size_t)v] == 0) frontier.push_back(v);
        int done = 0, lvl = 1;
        while (!frontier.empty()) {
            for (int v : frontier) level[(size_t)v] = lvl;
            done += (int)frontier.size();
            next.clear();
            if (seq)
                for (int v : frontier)
                    for (int j = 0; j < n; ++j)
                        if (j != v && seq[(size_t)v + (size_t)j * n] != 0.0 && --indeg[(size_t)j] == 0) next.push_back(j);
            std::sort(next.begin(), next.end());
            frontier.swap(next);
            lvl += 1;
        }
        if (done != n) {
            delete s;
            return ml_fail(PDMPC_ERR_INVALID, "pdmpc_ml_step_create: directed_coupling_sequential has a cycle");
        }
    }
    // slots: level by level, ascending vehicle index within a level (find(levels_of_vehicles == i_level))
    std::vector<int32_t> slot_vehicle((size_t)n), slot_of((size_t)n);
    for (int v = 0; v < n; ++v) slot_vehicle[(size_t)v] = v;
    std::stable_sort(slot_vehicle.begin(), slot_vehicle.end(), [&](int32_t a, int32_t b) { return level[(size_t)a] < level[(size_t)b]; });
    for (int sl = 0; sl < n; ++sl) slot_of[(size_t)slot_vehicle[(size_t)sl]] = sl;
    s->store.resize((size_t)n);
    s->in.resize((size_t)n);
    s->fallback.resize((size_t)n);
    s->pred_offset.assign((size_t)n + 1, 0);
    s->order.resize((size_t)n);
    s->levels.resize((size_t)n);
    for (int v = 0; v < n; ++v) s->levels[(size_t)v] = level[(size_t)v];
    auto bad = [&](const std::string& what) {
        delete s;
        return ml_fail(PDMPC_ERR_INVALID, "pdmpc_ml_step_create: " + what);
    };
    for (int sl = 0; sl < n; ++sl) {
        const int v = slot_vehicle[(size_t)sl];
        s->order[(size_t)sl] = v + 1;
        const pdmpc_ml_iter& it = iters[v];
        VehicleStore& st = s->store[(size_t)sl];
        if (!it.x0 || it.n_x0 < 3) return bad("x0 needs x, y, yaw");
        const pdmpc_ml_matrix& ref = it.reference_trajectory_points;
        if (ref.rows != Hp || ref.cols != 2 || !ref.data) return bad("reference_trajectory_points must be Hp x 2");
        if ((int64_t)it.v_ref.rows * it.v_ref.cols != Hp || !it.v_ref.data) return bad("v_ref must hold Hp elements");
        st.ref_x.assign(ref.data, ref.data + Hp);            // column 1
        st.ref_y.assign(ref.data + Hp, ref.data + 2 * Hp);   // column 2
        st.v_ref.assign(it.v_ref.data, it.v_ref.data + Hp);
        for (int side = 0; side < 2; ++side) {
            const pdmpc_ml_matrix& b = it.lanelet_boundary[side];
            std::vector<double>& bx = side == 0 ? st.lx : st.rx;
            std::vector<double>& by = side == 0 ? st.ly : st.ry;
            if ((int64_t)b.rows * b.cols == 0) continue;
            if (b.rows != 2 || !b.data) return bad("a lanelet boundary must be 2 x P");
            for (int q = 0; q < b.cols; ++q) {
                bx.push_back(b.data[2 * (size_t)q]);
                by.push_back(b.data[2 * (size_t)q + 1]);
            }
        }
        for (int q = 0; q < it.n_obstacles; ++q)
            if (!st.stat.add(it.obstacles[q])) return bad("obstacles must be 2 x V");
        if (it.dyn_rows > 0 && it.dyn_cols != Hp) return bad("dynamic_obstacle_area must be n_d x Hp");
        if (it.hdv_rows > 0 && it.hdv_cols != Hp) return bad("hdv_reachable_sets must be n_h x Hp");
        if (it.dyn_rows > 0 && !add_cell_rows(st.dyn, it.dynamic_obstacle_area, it.dyn_rows, it.dyn_cols)) return bad("dynamic obstacle areas must be 2 x V");
        if (it.hdv_rows > 0 && !add_cell_rows(st.hdv, it.hdv_reachable_sets, it.hdv_rows, it.hdv_cols)) return bad("reachable sets must be 2 x V");
        bool has_fb = false;
        if (fallback) {
            has_fb = true;
            for (int k = 0; k < Hp; ++k) has_fb = has_fb && (int64_t)fallback[(size_t)v + (size_t)k * n].rows * fallback[(size_t)v + (size_t)k * n].cols != 0;
            if (has_fb)
                for (int k = 0; k < Hp; ++k)
                    if (!st.fb.add(fallback[(size_t)v + (size_t)k * n])) return bad("fallback areas must be 2 x V");
        }
        static const double zero = 0.0;
        pdmpc_vehicle_in& in = s->in[(size_t)sl];
        std::memset(&in, 0, sizeof in);
        in.x0 = it.x0[0];
        in.y0 = it.x0[1];
        in.yaw0 = it.x0[2];
        in.trim0 = it.trim_index;
        in.n_left = (int32_t)st.lx.size();
        in.n_right = (int32_t)st.rx.size();
        in.ref_x = st.ref_x.data();
        in.ref_y = st.ref_y.data();
        in.v_ref = st.v_ref.data();
        in.left_x = st.lx.empty() ? &zero : st.lx.data();
        in.left_y = st.ly.empty() ? &zero : st.ly.data();
        in.right_x = st.rx.empty() ? &zero : st.rx.data();
        in.right_y = st.ry.empty() ? &zero : st.ry.data();
        in.obstacles = st.stat.view();
        in.dynamic_obstacles = st.dyn.view();
        in.hdv_reachable_sets = st.hdv.view();
        s->fallback[(size_t)sl] = has_fb ? st.fb.view() : pdmpc_polygon_set{0, nullptr, nullptr, nullptr};
        s->any_fallback = s->any_fallback || has_fb;
        // sequential predecessors: find(directed_coupling_sequential(:, v))', ascending, as slots
        if (seq)
            for (int i = 0; i < n; ++i)
                if (i != v && seq[(size_t)i + (size_t)v * n] != 0.0) s->pred_index.push_back(slot_of[(size_t)i]);
        s->pred_offset[(size_t)sl + 1] = (int32_t)s->pred_index.size();
    }
    s->pred_index.push_back(0);  // (never read: keeps data() non-null for an uncoupled step)
    // the views were taken while the stores could still move: take them again now that the vector is final
    for (int sl = 0; sl < n; ++sl) {
        VehicleStore& st = s->store[(size_t)sl];
        pdmpc_vehicle_in& in = s->in[(size_t)sl];
        static const double zero = 0.0;
        in.ref_x = st.ref_x.data();
        in.ref_y = st.ref_y.data();
        in.v_ref = st.v_ref.data();
        in.left_x = st.lx.empty() ? &zero : st.lx.data();
        in.left_y = st.ly.empty() ? &zero : st.ly.data();
        in.right_x = st.rx.empty() ? &zero : st.rx.data();
        in.right_y = st.ry.empty() ? &zero : st.ry.data();
        in.obstacles = st.stat.view();
        in.dynamic_obstacles = st.dyn.view();
        in.hdv_reachable_sets = st.hdv.view();
        if (s->fallback[(size_t)sl].n_polygons) s->fallback[(size_t)sl] = st.fb.view();
    }
    *out = s;
    return PDMPC_OK;
}

int pdmpc_ml_step_problem(const pdmpc_ml_step* s, int32_t* n, const pdmpc_vehicle_in** in, const int32_t** pred_offset, const int32_t** pred_index,
                          const pdmpc_polygon_set** fallback, const int32_t** order, const int32_t** levels) {
    if (!s) return ml_fail(PDMPC_ERR_INVALID, "null step");
    if (n) *n = s->n;
    if (in) *in = s->in.data();
    if (pred_offset) *pred_offset = s->pred_offset.data();
    if (pred_index) *pred_index = s->pred_index.data();
    if (fallback) *fallback = s->fallback.data();
    if (order) *order = s->order.data();
    if (levels) *levels = s->levels.data();
    return PDMPC_OK;
}

void pdmpc_ml_step_destroy(pdmpc_ml_step* s) { delete s; }

int pdmpc_ml_plan_step(pdmpc_handle* h, const pdmpc_ml_step* s, pdmpc_vehicle_out* out) {
    if (!h || !s || (s->n > 0 && !out)) return ml_fail(PDMPC_ERR_INVALID, "pdmpc_ml_plan_step: null argument");
    std::vector<pdmpc_vehicle_out> slots((size_t)std::max(s->n, 1));
    const int rc = pdmpc_plan_step(h, s->n, s->in.data(), s->pred_offset.data(), s->pred_index.data(), s->any_fallback ? s->fallback.data() : nullptr, slots.data());
    if (rc) {
        g_ml_err = pdmpc_last_error();
        return rc;
    }
    for (int sl = 0; sl < s->n; ++sl) out[(size_t)s->order[(size_t)sl] - 1] = slots[(size_t)sl];
    return PDMPC_OK;
}

int pdmpc_ml_plan_step_weighted(pdmpc_handle* h, const pdmpc_ml_step* s, const double* weights, pdmpc_vehicle_out* out) {
    if (!h || !s) return ml_fail(PDMPC_ERR_INVALID, "pdmpc_ml_plan_step_weighted: null argument");
    if (weights && s->n > 0) {  // (per vehicle -> per slot of the step as it is handed to pdmpc_plan_step)
        std::vector<double> w((size_t)s->n);
        for (int sl = 0; sl < s->n; ++sl) w[(size_t)sl] = weights[(size_t)s->order[(size_t)sl] - 1];
        const int rc = pdmpc_set_step_weights(h, s->n, w.data());
        if (rc) {
            g_ml_err = pdmpc_last_error();
            return rc;
        }
    }
    return pdmpc_ml_plan_step(h, s, out);
}

int pdmpc_ml_group_plan_step(pdmpc_group* g, const pdmpc_ml_step* s, const double* weights, int32_t mode, pdmpc_vehicle_out* out) {
    if (!g || !s || (s->n > 0 && !out)) return ml_fail(PDMPC_ERR_INVALID, "pdmpc_ml_group_plan_step: null argument");
    std::vector<pdmpc_vehicle_out> slots((size_t)std::max(s->n, 1));
    std::vector<double> w;
    if (weights) {  // (per vehicle -> per slot)
        w.resize((size_t)s->n);
        for (int sl = 0; sl < s->n; ++sl) w[(size_t)sl] = weights[(size_t)s->order[(size_t)sl] - 1];
    }
    const int rc = pdmpc_group_plan_step(g, s->n, s->in.data(), s->pred_offset.data(), s->pred_index.data(), s->any_fallback ? s->fallback.data() : nullptr,
                                         weights ? w.data() : nullptr, mode, slots.data());
    if (rc) {
        g_ml_err = pdmpc_last_error();
        return rc;
    }
    for (int sl = 0; sl < s->n; ++sl) out[(size_t)s->order[(size_t)sl] - 1] = slots[(size_t)sl];
    return PDMPC_OK;
}

int pdmpc_ml_plan_level(pdmpc_handle* h, int32_t Hp, int32_t n, const pdmpc_ml_iter* iters, pdmpc_vehicle_out* out) {
    pdmpc_ml_step* s = nullptr;
    int rc = pdmpc_ml_step_create(Hp, n, iters, nullptr, nullptr, &s);
    if (rc) return rc;
    rc = pdmpc_plan_batch(h, n, s->in.data(), out);  // (no couplings: slot order = vehicle order)
    if (rc) g_ml_err = pdmpc_last_error();
    pdmpc_ml_step_destroy(s);
    return rc;
}

void pdmpc_ml_record_arrays(const pdmpc_vehicle_out* r, int32_t Hp, double* trims, double* cols, double* y, double* shapes, double* nodes, double* path) {
    if (!r || Hp < 1 || Hp > PDMPC_HP_MAX) return;
    for (int k = 0; k < Hp; ++k) {
        if (trims) trims[k] = r->predicted_trims[k];
        if (cols) cols[k] = r->shape_cols[k];
        if (y)
            for (int c = 0; c < 3; ++c) y[(size_t)k + (size_t)c * Hp] = r->y_predicted[k][c];  // Hp x 3
        if (shapes)
            for (int row = 0; row < 2; ++row)
                for (int v = 0; v < PDMPC_VMAX; ++v) shapes[(size_t)k + (size_t)row * Hp + (size_t)v * Hp * 2] = r->shapes[k][row][v];  // Hp x 2 x VMAX
    }
    for (int k = 0; k <= Hp; ++k) {
        if (path) path[k] = r->tree_path[k];
        if (nodes)
            for (int c = 0; c < 8; ++c) nodes[(size_t)k + (size_t)c * (Hp + 1)] = r->path_nodes[k][c];  // (Hp + 1) x 8
    }
}

}  // extern "C"
