// pdmpc_mex.cpp — MATLAB C++ MEX gateway to libpdmpc_hip.so (source only; needs mex.hpp from a MATLAB install).
//
// Build inside MATLAB, next to compile_priority_queue.m's recipe:
//     mex -R2018a pdmpc_mex.cpp -I<repo>/include -L<repo>/p-dmpc_amd/csrc -lpdmpc_hip
// Commands mirror the life cycle of include/pdmpc.h:
//     h   = pdmpc_mex('create', Hp, checker, dt_seconds)
//           pdmpc_mex('upload_mpa', h, transition_matrix_single, maneuvers)
//     out = pdmpc_mex('plan_sampled', ... same arguments ..., seed)      the sampled optimizer (MonteCarloTreeSearch.m)
//     out = pdmpc_mex('plan', h, x0, trim, ref_points(Hp x 2), v_ref, obstacles, dynamic_obstacle_area,
//                     lanelet_boundary(1 x 2 cell), hdv_reachable_sets)
//           pdmpc_mex('destroy', h)
// It replaces the command protocol of priority_queue_interface_mex.cpp:33-40: the queue now lives inside the kernel.
#include <cstring>
#include <string>
#include <vector>

#include "mex.hpp"
#include "mexAdapter.hpp"
#include "pdmpc.h"

using matlab::data::Array;
using matlab::data::CellArray;
using matlab::data::StructArray;
using matlab::data::TypedArray;
using matlab::mex::ArgumentList;

namespace {

struct PolySet {  // owns the flattened copy a pdmpc_polygon_set points into
    std::vector<int32_t> off{0};
    std::vector<double> x, y;
    void add(const TypedArray<double>& m) {  // 2 x V
        const size_t V = m.getDimensions()[1];
        for (size_t v = 0; v < V; ++v) {
            x.push_back(m[0][v]);
            y.push_back(m[1][v]);
        }
        off.push_back((int32_t)x.size());
    }
    pdmpc_polygon_set view() const {
        static const double zero = 0.0;
        return {(int32_t)off.size() - 1, off.data(), x.empty() ? &zero : x.data(), y.empty() ? &zero : y.data()};
    }
};

// cell (rows x Hp) -> polygons in row-major order i*Hp + (k-1), the ABI's indexing
void flatten_rows(const CellArray& c, PolySet& out) {
    const auto dims = c.getDimensions();
    for (size_t i = 0; i < dims[0]; ++i)
        for (size_t k = 0; k < dims[1]; ++k) out.add(c[i][k]);
}

}  // namespace

class MexFunction : public matlab::mex::Function {
    matlab::data::ArrayFactory f;

    void fail(const std::string& what) {
        getEngine()->feval(u"error", 0, std::vector<Array>({f.createScalar(what + ": " + pdmpc_last_error())}));
    }

public:
    void operator()(ArgumentList outputs, ArgumentList inputs) {
        const std::string cmd = matlab::data::CharArray(inputs[0]).toAscii();
        if (cmd == "create") {
            pdmpc_config cfg{};
            cfg.Hp = (int32_t)inputs[1][0];
            cfg.checker = (int32_t)inputs[2][0];
            cfg.dt_seconds = inputs[3][0];
            cfg.max_vehicles = 1;
            pdmpc_handle* h = nullptr;
            if (pdmpc_create(&cfg, &h) != PDMPC_OK) fail("pdmpc_create");
            outputs[0] = f.createScalar<uint64_t>((uint64_t)h);
            return;
        }
        pdmpc_handle* h = (pdmpc_handle*)(uint64_t)inputs[1][0];
        if (cmd == "destroy") {
            pdmpc_destroy(h);
            return;
        }
        if (cmd == "upload_mpa") {
            const TypedArray<double> T = inputs[2];  // n x n x Hp, column-major
            const CellArray man = inputs[3];         // n x n cell of structs (generate_maneuver.m:25-34)
            const auto d = T.getDimensions();
            const int n = (int)d[0], Hp = (int)d[2];
            std::vector<uint8_t> trans((size_t)Hp * n * n);
            std::vector<int32_t> index((size_t)n * n, -1);
            std::vector<pdmpc_maneuver> mans;
            for (int k = 0; k < Hp; ++k)
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < n; ++j) trans[((size_t)k * n + i) * n + j] = T[i][j][k] != 0;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    const Array cell = man[i][j];
                    if (cell.isEmpty()) continue;
                    const StructArray s = cell;
                    pdmpc_maneuver m{};
                    m.dx = TypedArray<double>(s[0]["dx"])[0];
                    m.dy = TypedArray<double>(s[0]["dy"])[0];
                    m.dyaw = TypedArray<double>(s[0]["dyaw"])[0];
                    const char* names[3] = {"area", "area_without_offset", "area_large_offset"};
                    double(*dst[3])[PDMPC_VMAX] = {m.area, m.area_without_offset, m.area_large_offset};
                    for (int a = 0; a < 3; ++a) {
                        const TypedArray<double> A = s[0][names[a]];
                        m.n_cols = (int32_t)A.getDimensions()[1];
                        for (int v = 0; v < m.n_cols; ++v) {
                            dst[a][0][v] = A[0][v];
                            dst[a][1][v] = A[1][v];
                        }
                    }
                    index[(size_t)i * n + j] = (int32_t)mans.size();
                    mans.push_back(m);
                }
            pdmpc_mpa mpa{n, Hp, trans.data(), index.data(), (int32_t)mans.size(), mans.data()};
            if (pdmpc_upload_mpa(h, &mpa) != PDMPC_OK) fail("pdmpc_upload_mpa");
            return;
        }
        if (cmd == "plan" || cmd == "plan_sampled") {  // plan_sampled: one more trailing argument, the seed (time_step + vehicle_index)
            const TypedArray<double> x0 = inputs[2];
            const TypedArray<double> ref = inputs[4];  // Hp x 2
            const TypedArray<double> vref = inputs[5];
            const size_t Hp = vref.getNumberOfElements();
            std::vector<double> rx(Hp), ry(Hp), vr(Hp);
            for (size_t k = 0; k < Hp; ++k) {
                rx[k] = ref[k][0];
                ry[k] = ref[k][1];
                vr[k] = vref[k];
            }
            PolySet stat, dyn, hdv, lb[2];
            const CellArray obstacles = inputs[6];
            for (auto e : obstacles) stat.add(e);
            flatten_rows(inputs[7], dyn);
            const CellArray boundary = inputs[8];
            for (int s = 0; s < 2; ++s)
                if (!Array(boundary[0][s]).isEmpty()) lb[s].add(boundary[0][s]);
            flatten_rows(inputs[9], hdv);
            pdmpc_vehicle_in in{};
            in.x0 = x0[0];
            in.y0 = x0[1];
            in.yaw0 = x0[2];
            in.trim0 = (int32_t)inputs[3][0];
            in.ref_x = rx.data();
            in.ref_y = ry.data();
            in.v_ref = vr.data();
            in.n_left = (int32_t)lb[0].x.size();
            in.n_right = (int32_t)lb[1].x.size();
            in.left_x = lb[0].x.data();
            in.left_y = lb[0].y.data();
            in.right_x = lb[1].x.data();
            in.right_y = lb[1].y.data();
            in.obstacles = stat.view();
            in.dynamic_obstacles = dyn.view();
            in.hdv_reachable_sets = hdv.view();
            pdmpc_vehicle_out out{};
            if (cmd == "plan") {
                if (pdmpc_plan_batch(h, 1, &in, &out) != PDMPC_OK) fail("pdmpc_plan_batch");
            } else {
                const uint32_t seed = (uint32_t)(double)inputs[10][0];  // MonteCarloTreeSearch.m:32
                if (pdmpc_plan_batch_sampled(h, 1, &in, &seed, &out) != PDMPC_OK) fail("pdmpc_plan_batch_sampled");
            }
            StructArray s = f.createStructArray({1, 1}, {"status", "n_expanded", "predicted_trims", "shape_cols", "y_predicted", "shapes", "path_nodes"});
            s[0]["status"] = f.createScalar<double>(out.status);
            s[0]["n_expanded"] = f.createScalar<double>(out.n_expanded);
            TypedArray<double> trims = f.createArray<double>({1, Hp}), cols = f.createArray<double>({1, Hp});
            TypedArray<double> y = f.createArray<double>({Hp, 3}), nodes = f.createArray<double>({Hp + 1, 8});
            TypedArray<double> shapes = f.createArray<double>({Hp, 2, PDMPC_VMAX});
            for (size_t k = 0; k < Hp; ++k) {
                trims[0][k] = out.predicted_trims[k];
                cols[0][k] = out.shape_cols[k];
                for (int c = 0; c < 3; ++c) y[k][c] = out.y_predicted[k][c];
                for (int r = 0; r < 2; ++r)
                    for (int v = 0; v < PDMPC_VMAX; ++v) shapes[k][r][v] = out.shapes[k][r][v];
            }
            for (size_t k = 0; k <= Hp; ++k)
                for (int c = 0; c < 8; ++c) nodes[k][c] = out.path_nodes[k][c];
            s[0]["predicted_trims"] = trims;
            s[0]["shape_cols"] = cols;
            s[0]["y_predicted"] = y;
            s[0]["shapes"] = shapes;
            s[0]["path_nodes"] = nodes;
            outputs[0] = s;
            return;
        }
        fail("unknown command " + cmd);
    }
};
