// pdmpc_mex.cpp — MATLAB C++ MEX gateway to libpdmpc_hip.so (source only; needs mex.hpp from a MATLAB install).
//
// Build inside MATLAB, next to compile_priority_queue.m's recipe:
//     mex -R2018a pdmpc_mex.cpp -I<repo>/include -L<repo>/p-dmpc_amd/csrc -lpdmpc_hip
// Commands:
//     h    = pdmpc_mex('create', Hp, checker, dt_seconds [, max_vehicles])
//            pdmpc_mex('upload_mpa', h, transition_matrix_single, maneuvers)
//     out  = pdmpc_mex('plan', h, iter_struct)                     one run_optimizer call (GraphSearchHip.m)
//     out  = pdmpc_mex('plan_sampled', h, iter_struct, seed)       the sampled optimizer (MonteCarloTreeSearchHip.m)
//     outs = pdmpc_mex('plan_level', h, iter_structs)              one computation level: n vehicles, one launch
//     outs = pdmpc_mex('plan_step', h, iter_structs, directed_coupling_sequential, fallback_areas [, weights])
//                                                                  ALL levels of a time step in one launch
//                                                                  (PrioritizedSequentialHipController.m)
//            pdmpc_mex('destroy', h)
// iter_struct(s): struct (array) with fields x0, trim_index, reference_trajectory_points (Hp x 2), v_ref, obstacles (cell),
// dynamic_obstacle_area (n_d x Hp cell), lanelet_boundary (1 x 2 cell), hdv_reachable_sets (n_h x Hp cell).
//
// This file holds NO index arithmetic: it turns matlab::data arrays into the (pointer, rows, cols) descriptors of
// include/pdmpc_matlab.h; the marshalling itself (cell order, column-major matrices, kahn levels, slot order) is
// csrc/matlab_marshal.cpp, compiled into libpdmpc_hip.so and unit-tested without MATLAB (tests/test_matlab_marshal.py).
// It replaces the command protocol of priority_queue_interface_mex.cpp:33-40: the queue now lives inside the kernel.
#include <deque>
#include <string>
#include <vector>

#include "mex.hpp"
#include "mexAdapter.hpp"
#include "pdmpc_matlab.h"

using matlab::data::Array;
using matlab::data::CellArray;
using matlab::data::StructArray;
using matlab::data::TypedArray;
using matlab::mex::ArgumentList;

namespace {

// descriptors point into the MATLAB arrays themselves (TypedArray<double> storage is contiguous, column-major); the arrays are
// kept alive in `pins` for the duration of the call
struct Pins {
    std::deque<TypedArray<double>> arrays;
    std::deque<std::vector<pdmpc_ml_matrix>> cells;
    pdmpc_ml_matrix matrix(const Array& a) {
        if (a.isEmpty()) return {nullptr, 0, 0};
        arrays.emplace_back(TypedArray<double>(a));
        const TypedArray<double>& t = arrays.back();
        const auto d = t.getDimensions();
        return {&*t.begin(), (int32_t)d[0], (int32_t)(t.getNumberOfElements() / d[0])};
    }
    const pdmpc_ml_matrix* cell(const Array& a, int32_t& rows, int32_t& cols) {  // linear (column-major) cell order, as MATLAB stores it
        const CellArray c = a;
        const auto d = c.getDimensions();
        rows = (int32_t)d[0];
        cols = d.size() > 1 ? (int32_t)d[1] : 1;
        cells.emplace_back();
        std::vector<pdmpc_ml_matrix>& v = cells.back();
        for (const Array e : c) v.push_back(matrix(e));
        return v.data();
    }
};

pdmpc_ml_iter iter_from_struct(const StructArray& s, size_t i, Pins& pins) {
    pdmpc_ml_iter it{};
    const pdmpc_ml_matrix x0 = pins.matrix(s[i]["x0"]);
    it.x0 = x0.data;
    it.n_x0 = x0.rows * x0.cols;
    it.trim_index = (int32_t)TypedArray<double>(s[i]["trim_index"])[0];
    it.reference_trajectory_points = pins.matrix(s[i]["reference_trajectory_points"]);
    it.v_ref = pins.matrix(s[i]["v_ref"]);
    int32_t r = 0, c = 0;
    it.obstacles = pins.cell(s[i]["obstacles"], r, c);
    it.n_obstacles = r * c;
    it.dynamic_obstacle_area = pins.cell(s[i]["dynamic_obstacle_area"], it.dyn_rows, it.dyn_cols);
    const pdmpc_ml_matrix* lb = pins.cell(s[i]["lanelet_boundary"], r, c);
    for (int side = 0; side < 2 && side < r * c; ++side) it.lanelet_boundary[side] = lb[side];
    it.hdv_reachable_sets = pins.cell(s[i]["hdv_reachable_sets"], it.hdv_rows, it.hdv_cols);
    return it;
}

}  // namespace

class MexFunction : public matlab::mex::Function {
    matlab::data::ArrayFactory f;

    void fail(const std::string& what, const char* detail) {
        getEngine()->feval(u"error", 0, std::vector<Array>({f.createScalar(what + ": " + detail)}));
    }

    // pdmpc_vehicle_out[n] -> 1 x n struct array in MATLAB's layout (pdmpc_ml_record_arrays)
    StructArray records(const std::vector<pdmpc_vehicle_out>& out, size_t Hp) {
        StructArray s = f.createStructArray({1, out.size()}, {"status", "n_expanded", "n_popped", "predicted_trims", "shape_cols", "y_predicted", "shapes", "path_nodes", "tree_path"});
        for (size_t i = 0; i < out.size(); ++i) {
            TypedArray<double> trims = f.createArray<double>({1, Hp}), cols = f.createArray<double>({1, Hp}), path = f.createArray<double>({1, Hp + 1});
            TypedArray<double> y = f.createArray<double>({Hp, 3}), nodes = f.createArray<double>({Hp + 1, 8}), shapes = f.createArray<double>({Hp, 2, PDMPC_VMAX});
            pdmpc_ml_record_arrays(&out[i], (int32_t)Hp, &*trims.begin(), &*cols.begin(), &*y.begin(), &*shapes.begin(), &*nodes.begin(), &*path.begin());
            s[i]["status"] = f.createScalar<double>(out[i].status);
            s[i]["n_expanded"] = f.createScalar<double>(out[i].n_expanded);
            s[i]["n_popped"] = f.createScalar<double>(out[i].n_popped);
            s[i]["predicted_trims"] = trims;
            s[i]["shape_cols"] = cols;
            s[i]["y_predicted"] = y;
            s[i]["shapes"] = shapes;
            s[i]["path_nodes"] = nodes;
            s[i]["tree_path"] = path;
        }
        return s;
    }

public:
    void operator()(ArgumentList outputs, ArgumentList inputs) {
        const std::string cmd = matlab::data::CharArray(inputs[0]).toAscii();
        if (cmd == "create") {
            pdmpc_config cfg{};
            cfg.Hp = (int32_t)inputs[1][0];
            cfg.checker = (int32_t)inputs[2][0];
            cfg.dt_seconds = inputs[3][0];
            cfg.max_vehicles = inputs.size() > 4 ? (int32_t)inputs[4][0] : 1;  // GraphSearchHip: 1; the step controller: options.amount
            pdmpc_handle* h = nullptr;
            if (pdmpc_create(&cfg, &h) != PDMPC_OK) fail("pdmpc_create", pdmpc_last_error());
            outputs[0] = f.createScalar<uint64_t>((uint64_t)h);
            return;
        }
        // ---- several GPUs (include/pdmpc.h: pdmpc_group_*): 'group_create', Hp, checker, dt, max_vehicles, n_gpus -> group;
        //      'group_destroy', 'group_upload_mpa' and 'group_plan_step' (iters, coupling, fallback areas [, weights [, mode]]) take it
        //      where the single-GPU commands take the handle
        if (cmd == "group_create") {
            pdmpc_config cfg{};
            cfg.Hp = (int32_t)inputs[1][0];
            cfg.checker = (int32_t)inputs[2][0];
            cfg.dt_seconds = inputs[3][0];
            cfg.max_vehicles = (int32_t)inputs[4][0];
            const int32_t n_gpus = (int32_t)inputs[5][0];
            pdmpc_group* g = nullptr;
            if (pdmpc_group_create(&cfg, n_gpus, nullptr, &g) != PDMPC_OK) fail("pdmpc_group_create", pdmpc_last_error());
            outputs[0] = f.createScalar<uint64_t>((uint64_t)g);
            return;
        }
        pdmpc_group* group = nullptr;
        pdmpc_handle* h = nullptr;
        if (cmd.rfind("group_", 0) == 0) {
            group = (pdmpc_group*)(uint64_t)inputs[1][0];
            if (cmd == "group_destroy") {
                pdmpc_group_destroy(group);
                return;
            }
            if (pdmpc_group_handle(group, 0, &h) != PDMPC_OK) fail("pdmpc_group_handle", pdmpc_last_error());
        } else {
            h = (pdmpc_handle*)(uint64_t)inputs[1][0];
        }
        if (cmd == "destroy") {
            pdmpc_destroy(h);
            return;
        }
        pdmpc_config hc{};
        if (pdmpc_get_config(h, &hc, nullptr) != PDMPC_OK) fail("pdmpc_get_config", pdmpc_last_error());
        const size_t Hp = (size_t)hc.Hp;
        if (cmd == "upload_mpa" || cmd == "group_upload_mpa") {
            Pins pins;
            const pdmpc_ml_matrix T = pins.matrix(inputs[2]);  // n x n x Hp, column-major
            const CellArray man = inputs[3];                   // n x n cell of structs (generate_maneuver.m:25-34)
            const int32_t n = T.rows;
            std::vector<pdmpc_ml_maneuver> cells;
            for (const Array e : man) {  // linear order
                pdmpc_ml_maneuver m{};
                if (!e.isEmpty()) {
                    const StructArray s = e;
                    m.present = 1;
                    m.dx = TypedArray<double>(s[0]["dx"])[0];
                    m.dy = TypedArray<double>(s[0]["dy"])[0];
                    m.dyaw = TypedArray<double>(s[0]["dyaw"])[0];
                    m.area = pins.matrix(s[0]["area"]);
                    m.area_without_offset = pins.matrix(s[0]["area_without_offset"]);
                    m.area_large_offset = pins.matrix(s[0]["area_large_offset"]);
                }
                cells.push_back(m);
            }
            int32_t n_dev = 1;
            if (group && pdmpc_group_size(group, &n_dev) != PDMPC_OK) fail("pdmpc_group_size", pdmpc_last_error());
            for (int32_t r = 0; r < n_dev; ++r) {  // (every device of a group holds the tables)
                if (group && pdmpc_group_handle(group, r, &h) != PDMPC_OK) fail("pdmpc_group_handle", pdmpc_last_error());
                if (pdmpc_ml_upload_mpa(h, T.data, n, T.cols / n, cells.data()) != PDMPC_OK) fail("pdmpc_ml_upload_mpa", pdmpc_ml_last_error());
            }
            return;
        }
        if (cmd == "plan" || cmd == "plan_sampled" || cmd == "plan_level" || cmd == "plan_step" || cmd == "group_plan_step") {
            Pins pins;
            const StructArray iters = inputs[2];
            const size_t n = iters.getNumberOfElements();
            std::vector<pdmpc_ml_iter> its;
            for (size_t i = 0; i < n; ++i) its.push_back(iter_from_struct(iters, i, pins));
            std::vector<pdmpc_vehicle_out> out(n);
            if (cmd == "plan_step" || cmd == "group_plan_step") {
                const pdmpc_ml_matrix seq = pins.matrix(inputs[3]);  // n x n directed_coupling_sequential
                // pdmpc_ml_step_create indexes seq[i + j * n] and fallback[v + k * n]: anything smaller reads outside MATLAB's arrays,
                // and an empty coupling matrix would silently plan the step without couplings
                if (seq.data == nullptr || (size_t)seq.rows != n || (size_t)seq.cols != n)
                    fail("plan_step", "directed_coupling_sequential must be an n x n matrix (n = number of iteration structs)");
                int32_t r = 0, c = 0;
                const pdmpc_ml_matrix* fb = inputs.size() > 4 && !inputs[4].isEmpty() ? pins.cell(inputs[4], r, c) : nullptr;  // n x Hp cell
                if (fb != nullptr && ((size_t)r != n || (size_t)c != Hp)) fail("plan_step", "fallback areas must be an n x Hp cell");
                pdmpc_ml_step* step = nullptr;
                if (pdmpc_ml_step_create((int32_t)Hp, (int32_t)n, its.data(), seq.data, fb, &step) != PDMPC_OK) fail("pdmpc_ml_step_create", pdmpc_ml_last_error());
                int rc;
                if (group) {
                    // weights: 1 x n expected work per vehicle (e.g. the n_expanded of its last plan), [] = equal; mode: 0 auto, 1 whole components, 2 levels
                    const pdmpc_ml_matrix w = inputs.size() > 5 && !inputs[5].isEmpty() ? pins.matrix(inputs[5]) : pdmpc_ml_matrix{nullptr, 0, 0};
                    if (w.data != nullptr && (size_t)w.rows * (size_t)w.cols != n) fail("group_plan_step", "weights must hold one value per vehicle");
                    const int32_t mode = inputs.size() > 6 ? (int32_t)inputs[6][0] : PDMPC_SHARD_AUTO;
                    rc = pdmpc_ml_group_plan_step(group, step, w.data, mode, out.data());
                } else {
                    // optional 5th argument: 1 x n expected work per vehicle (e.g. the n_popped of its last plan), [] = none: the launch
                    // fills its slots by priority (pdmpc_ml_plan_step_weighted -> pdmpc_set_step_weights)
                    const pdmpc_ml_matrix w = inputs.size() > 5 && !inputs[5].isEmpty() ? pins.matrix(inputs[5]) : pdmpc_ml_matrix{nullptr, 0, 0};
                    if (w.data != nullptr && (size_t)w.rows * (size_t)w.cols != n) fail("plan_step", "weights must hold one value per vehicle");
                    rc = pdmpc_ml_plan_step_weighted(h, step, w.data, out.data());
                }
                pdmpc_ml_step_destroy(step);
                if (rc != PDMPC_OK) fail(group ? "pdmpc_ml_group_plan_step" : "pdmpc_ml_plan_step", pdmpc_ml_last_error());
            } else if (cmd == "plan_sampled") {
                pdmpc_ml_step* step = nullptr;
                if (pdmpc_ml_step_create((int32_t)Hp, (int32_t)n, its.data(), nullptr, nullptr, &step) != PDMPC_OK) fail("pdmpc_ml_step_create", pdmpc_ml_last_error());
                const pdmpc_vehicle_in* in = nullptr;
                pdmpc_ml_step_problem(step, nullptr, &in, nullptr, nullptr, nullptr, nullptr, nullptr);
                const uint32_t seed = (uint32_t)(double)inputs[3][0];  // MonteCarloTreeSearch.m:32
                const int rc = pdmpc_plan_batch_sampled(h, 1, in, &seed, out.data());
                pdmpc_ml_step_destroy(step);
                if (rc != PDMPC_OK) fail("pdmpc_plan_batch_sampled", pdmpc_last_error());
            } else {
                if (pdmpc_ml_plan_level(h, (int32_t)Hp, (int32_t)n, its.data(), out.data()) != PDMPC_OK) fail("pdmpc_ml_plan_level", pdmpc_ml_last_error());
            }
            outputs[0] = records(out, Hp);
            return;
        }
        fail("unknown command " + cmd, "");
    }
};
