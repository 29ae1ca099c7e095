/* pdmpc_matlab.h — the MATLAB-shaped entry points of libpdmpc_hip.so: everything p-dmpc_amd/matlab/pdmpc_mex.cpp does
 * besides touching matlab::data lives behind these calls, so the marshalling can be compiled and tested without MATLAB.
 *
 * Conventions of this header (MATLAB's, not C's):
 *  - numeric arrays are double, COLUMN-major: element (r, c) of an R x C matrix at data[r + c * R], element (i, j, k) of an
 *    n x n x Hp array at data[i + j * n + k * n * n];
 *  - a cell array arrives as an array of matrix descriptors in MATLAB's linear (column-major) cell order: cell (i, k) of an
 *    R x C cell at cells[i + k * R]; an empty cell has rows * cols == 0;
 *  - polygons are 2 x V matrices [x; y] (generate_maneuver.m:46, vectorize_all_obstacles.m:68-75): x_v at data[2 v], y_v at
 *    data[2 v + 1];
 *  - vehicle indices and computation levels handed back are 1-based, as kahn.m and the controllers use them.
 *
 * What each call replaces in the reference:
 *   pdmpc_ml_upload_mpa      mpa.transition_matrix_single / mpa.maneuvers as GraphSearch reads them
 *                            (hlc/optimizer/graph_search/GraphSearch.m:40-46, expand_node.m:18-33)
 *   pdmpc_ml_plan_level      one computation level: the inner loop of PrioritizedSequentialController.controller
 *                            (hlc/controller/prioritized/PrioritizedSequentialController.m:86-88), n run_optimizer calls at once
 *   pdmpc_ml_plan_step       the whole double loop (:77-94) in one call: kahn.m levels from directed_coupling_sequential, the
 *                            predecessors' solved areas handed over on the device (PrioritizedController.m:476-491)
 *   pdmpc_ml_group_plan_step the same step over several GPUs: the per-level exchange of solved areas between the vehicles
 *                            (hlc/communication/PredictionsCommunication.m:34-63) as an RCCL all-gather between the devices
 *   pdmpc_ml_record_arrays   the fields of ControlResultsInfo in MATLAB's layout (OptimizerInterface.m:63-101)
 */
#ifndef PDMPC_MATLAB_H
#define PDMPC_MATLAB_H

#include "pdmpc.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    const double* data; /* column-major; may be NULL when rows * cols == 0 */
    int32_t rows, cols;
} pdmpc_ml_matrix;

/* mpa.maneuvers{i, j} (generate_maneuver.m:25-34); present == 0 for an empty cell */
typedef struct {
    int32_t present;
    int32_t _pad;
    double dx, dy, dyaw;
    pdmpc_ml_matrix area, area_without_offset, area_large_offset; /* 2 x V each */
} pdmpc_ml_maneuver;

/* The one-vehicle IterationData slice as PrioritizedController.plan builds it (PrioritizedController.m:297-341) */
typedef struct {
    const double* x0;                            /* iter.x0(1, :): x, y, yaw (, speed ...) */
    int32_t n_x0;                                /* >= 3 */
    int32_t trim_index;                          /* iter.trim_indices, 1-based */
    pdmpc_ml_matrix reference_trajectory_points; /* squeeze(iter.reference_trajectory_points(1, :, :)): Hp x 2 */
    pdmpc_ml_matrix v_ref;                       /* iter.v_ref(1, :): Hp elements */
    int32_t n_obstacles;                         /* numel(iter.obstacles) */
    const pdmpc_ml_matrix* obstacles;            /* cells of 2 x V */
    int32_t dyn_rows, dyn_cols;                  /* size(iter.dynamic_obstacle_area): n_d x Hp */
    const pdmpc_ml_matrix* dynamic_obstacle_area;
    pdmpc_ml_matrix lanelet_boundary[2];         /* iter.predicted_lanelet_boundary{1, 1} (left), {1, 2} (right): 2 x P or empty */
    int32_t hdv_rows, hdv_cols;                  /* size(iter.hdv_reachable_sets(adjacent, :)): n_h x Hp */
    const pdmpc_ml_matrix* hdv_reachable_sets;
} pdmpc_ml_iter;

/* ---- MPA ---- */
typedef struct pdmpc_ml_mpa pdmpc_ml_mpa;
/* transition: n x n x Hp (transition_matrix_single); maneuvers: n x n cell in linear order */
int pdmpc_ml_mpa_create(const double* transition, int32_t n_trims, int32_t Hp, const pdmpc_ml_maneuver* maneuvers, pdmpc_ml_mpa** out);
const pdmpc_mpa* pdmpc_ml_mpa_view(const pdmpc_ml_mpa* m); /* the tables in the form pdmpc_upload_mpa takes */
void pdmpc_ml_mpa_destroy(pdmpc_ml_mpa* m);
int pdmpc_ml_upload_mpa(pdmpc_handle* handle, const double* transition, int32_t n_trims, int32_t Hp, const pdmpc_ml_maneuver* maneuvers);

/* ---- a time step (or one level of it) ---- */
typedef struct pdmpc_ml_step pdmpc_ml_step;
/* iters[v]: vehicle v's slice WITHOUT the areas of its sequential predecessors (they are handed over on the device).
 * directed_coupling_sequential: n x n, (i, j) ~= 0 <=> vehicle i plans before vehicle j and j reads i's prediction in this
 *   step (iter.directed_coupling_sequential); NULL: no couplings (one level).
 * fallback: n x Hp cell (linear order) of the areas vehicle v publishes when its search is exhausted
 *   (PrioritizedController.m:568-616, 678-718: the standstill rectangle or the previous plan shifted); NULL or empty cells: none. */
int pdmpc_ml_step_create(int32_t Hp, int32_t n, const pdmpc_ml_iter* iters, const double* directed_coupling_sequential, const pdmpc_ml_matrix* fallback,
                         pdmpc_ml_step** out);
/* the problem exactly as pdmpc_plan_step receives it: slots in level order (kahn.m; ascending vehicle index within a level, as
 * find(levels == i) yields them); order[s] = vehicle (1-based) in slot s, levels[v] = computation level (1-based) of vehicle v + 1 */
int pdmpc_ml_step_problem(const pdmpc_ml_step* s, int32_t* n, const pdmpc_vehicle_in** in, const int32_t** pred_offset, const int32_t** pred_index,
                          const pdmpc_polygon_set** fallback, const int32_t** order, const int32_t** levels);
void pdmpc_ml_step_destroy(pdmpc_ml_step* s);
/* plans the step with ONE launch (pdmpc_plan_step); out[v] = record of VEHICLE v (not of slot v) */
int pdmpc_ml_plan_step(pdmpc_handle* handle, const pdmpc_ml_step* s, pdmpc_vehicle_out* out);
/* ... with the expected work per VEHICLE (MATLAB's vehicle order; e.g. n_popped of each vehicle's last plan; NULL: none): the launch
 * fills its slots by priority (pdmpc_set_step_weights).  Same records. */
int pdmpc_ml_plan_step_weighted(pdmpc_handle* handle, const pdmpc_ml_step* step, const double* weights, pdmpc_vehicle_out* out);
/* the same step over the GPUs of a group (pdmpc_group_*, include/pdmpc.h): weights[v] (may be NULL) = expected work of VEHICLE v,
 * mode = PDMPC_SHARD_*; out[v] = record of vehicle v.  What PredictionsCommunication.m:34-63 does between the vehicles' processes
 * happens between the devices: an RCCL all-gather of the solved areas. */
int pdmpc_ml_group_plan_step(pdmpc_group* group, const pdmpc_ml_step* s, const double* weights, int32_t mode, pdmpc_vehicle_out* out);
/* convenience: n uncoupled vehicles (one computation level) in one launch */
int pdmpc_ml_plan_level(pdmpc_handle* handle, int32_t Hp, int32_t n, const pdmpc_ml_iter* iters, pdmpc_vehicle_out* out);

/* ---- results in MATLAB's layout ----
 * predicted_trims, shape_cols: 1 x Hp; y_predicted: Hp x 3; shapes: Hp x 2 x PDMPC_VMAX; path_nodes: (Hp + 1) x 8 (rows in
 * NodeInfo order, NodeInfo.m:4-13); tree_path: 1 x (Hp + 1).  Any pointer may be NULL. */
void pdmpc_ml_record_arrays(const pdmpc_vehicle_out* rec, int32_t Hp, double* predicted_trims, double* shape_cols, double* y_predicted, double* shapes,
                            double* path_nodes, double* tree_path);

const char* pdmpc_ml_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PDMPC_MATLAB_H */
