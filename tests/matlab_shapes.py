"""MATLAB-shaped fixtures for the entry points of include/pdmpc_matlab.h (test infrastructure).

Turns the Python objects the tests use (VehicleIter, MotionPrimitiveAutomaton) into what p-dmpc_amd/matlab/pdmpc_mex.cpp
hands to libpdmpc_hip.so: Fortran-ordered double arrays, cells as arrays of matrix descriptors in MATLAB's linear
(column-major) cell order, n x n x Hp transition matrix, n x n coupling matrix.  Written from the header's comments only.
"""
import ctypes as C

import numpy as np

from pdmpc import abi, backend

_dp = C.POINTER(C.c_double)


class MlMatrix(C.Structure):
    _fields_ = [("data", _dp), ("rows", C.c_int32), ("cols", C.c_int32)]


class MlManeuver(C.Structure):
    _fields_ = [("present", C.c_int32), ("_pad", C.c_int32), ("dx", C.c_double), ("dy", C.c_double), ("dyaw", C.c_double),
                ("area", MlMatrix), ("area_without_offset", MlMatrix), ("area_large_offset", MlMatrix)]


class MlIter(C.Structure):
    _fields_ = [("x0", _dp), ("n_x0", C.c_int32), ("trim_index", C.c_int32), ("reference_trajectory_points", MlMatrix), ("v_ref", MlMatrix),
                ("n_obstacles", C.c_int32), ("obstacles", C.POINTER(MlMatrix)), ("dyn_rows", C.c_int32), ("dyn_cols", C.c_int32),
                ("dynamic_obstacle_area", C.POINTER(MlMatrix)), ("lanelet_boundary", MlMatrix * 2), ("hdv_rows", C.c_int32), ("hdv_cols", C.c_int32),
                ("hdv_reachable_sets", C.POINTER(MlMatrix))]


def lib():
    L = backend.load_library()
    if not getattr(L, "_ml_declared", False):
        H = C.c_void_p
        L.pdmpc_ml_mpa_create.argtypes = [_dp, C.c_int32, C.c_int32, C.POINTER(MlManeuver), C.POINTER(C.c_void_p)]
        L.pdmpc_ml_mpa_view.argtypes = [C.c_void_p]
        L.pdmpc_ml_mpa_view.restype = C.POINTER(abi.Mpa)
        L.pdmpc_ml_mpa_destroy.argtypes = [C.c_void_p]
        L.pdmpc_ml_mpa_destroy.restype = None
        L.pdmpc_ml_upload_mpa.argtypes = [H, _dp, C.c_int32, C.c_int32, C.POINTER(MlManeuver)]
        L.pdmpc_ml_step_create.argtypes = [C.c_int32, C.c_int32, C.POINTER(MlIter), _dp, C.POINTER(MlMatrix), C.POINTER(C.c_void_p)]
        L.pdmpc_ml_step_problem.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.POINTER(abi.VehicleIn)), C.POINTER(abi.c_int32_p), C.POINTER(abi.c_int32_p),
                                            C.POINTER(C.POINTER(abi.PolygonSet)), C.POINTER(abi.c_int32_p), C.POINTER(abi.c_int32_p)]
        L.pdmpc_ml_step_destroy.argtypes = [C.c_void_p]
        L.pdmpc_ml_step_destroy.restype = None
        L.pdmpc_ml_plan_step.argtypes = [H, C.c_void_p, C.POINTER(abi.VehicleOut)]
        L.pdmpc_ml_plan_level.argtypes = [H, C.c_int32, C.c_int32, C.POINTER(MlIter), C.POINTER(abi.VehicleOut)]
        L.pdmpc_ml_record_arrays.argtypes = [C.c_void_p, C.c_int32] + [_dp] * 6
        L.pdmpc_ml_record_arrays.restype = None
        L.pdmpc_ml_last_error.restype = C.c_char_p
        L._ml_declared = True
    return L


class Keep:
    """Keeps the Fortran-ordered arrays alive that the descriptors point into."""

    def __init__(self):
        self.refs = []

    def matrix(self, a):
        """2-D array -> descriptor of its column-major copy (what mxGetDoubles of the same MATLAB matrix returns)."""
        if a is None or np.size(a) == 0:
            return MlMatrix(None, 0, 0)
        f = np.asfortranarray(np.asarray(a, dtype=np.float64))
        if f.ndim == 1:
            f = np.asfortranarray(f.reshape(1, -1))
        self.refs.append(f)
        return MlMatrix(f.ctypes.data_as(_dp), f.shape[0], f.shape[1])

    def cell(self, rows, n_rows, n_cols):
        """rows[i][k] (a Python list of lists = the R x C cell) -> descriptors in MATLAB's linear order i + k * R."""
        arr = (MlMatrix * max(n_rows * n_cols, 1))()
        for i in range(n_rows):
            for k in range(n_cols):
                arr[i + k * n_rows] = self.matrix(rows[i][k])
        self.refs.append(arr)
        return arr


def ml_iter(it, Hp, keep):
    s = MlIter()
    x0 = np.ascontiguousarray(np.asarray(it.x0, dtype=np.float64))
    keep.refs.append(x0)
    s.x0, s.n_x0 = x0.ctypes.data_as(_dp), len(x0)
    s.trim_index = int(it.trim_index)
    s.reference_trajectory_points = keep.matrix(np.asarray(it.reference_trajectory_points, dtype=np.float64).reshape(Hp, 2))
    s.v_ref = keep.matrix(np.asarray(it.v_ref, dtype=np.float64).reshape(1, Hp))
    s.n_obstacles = len(it.obstacles)
    s.obstacles = keep.cell([[o] for o in it.obstacles], len(it.obstacles), 1)
    nd = len(it.dynamic_obstacle_area)
    s.dyn_rows, s.dyn_cols = nd, Hp
    s.dynamic_obstacle_area = keep.cell(it.dynamic_obstacle_area, nd, Hp)
    for side in range(2):
        s.lanelet_boundary[side] = keep.matrix(it.predicted_lanelet_boundary[side])
    nh = len(it.hdv_reachable_sets)
    s.hdv_rows, s.hdv_cols = nh, Hp
    s.hdv_reachable_sets = keep.cell(it.hdv_reachable_sets, nh, Hp)
    return s


def ml_mpa_args(mpa, keep):
    """(transition n x n x Hp column-major, n, Hp, maneuver cell n x n in linear order)"""
    n, Hp = int(mpa.n_trims), int(mpa.Hp)
    T = np.asfortranarray(np.asarray(mpa.transition_matrix_single, dtype=np.float64))
    assert T.shape == (n, n, Hp)
    keep.refs.append(T)
    man = (MlManeuver * (n * n))()
    for i in range(n):
        for j in range(n):
            m = mpa.maneuvers[i][j]
            c = man[i + j * n]
            if m is None:
                c.present = 0
                continue
            c.present = 1
            c.dx, c.dy, c.dyaw = float(m.dx), float(m.dy), float(m.dyaw)
            c.area = keep.matrix(m.area)
            c.area_without_offset = keep.matrix(m.area_without_offset)
            c.area_large_offset = keep.matrix(m.area_large_offset)
    keep.refs.append(man)
    return T.ctypes.data_as(_dp), n, Hp, man


def vehicle_order_problem(prob):
    """A controller step problem (slots in level order) back in VEHICLE order, as the MATLAB controller holds it: per-vehicle
    iters, the n x n directed_coupling_sequential matrix, per-vehicle fallback areas."""
    n = len(prob["iters"])
    order = prob["order"]  # slot -> vehicle (0-based)
    slot_of = {v: s for s, v in enumerate(order)}
    iters = [prob["iters"][slot_of[v]] for v in range(n)]
    fallback = [prob["fallback"][slot_of[v]] for v in range(n)]
    seq = np.zeros((n, n))
    for s, ps in enumerate(prob["preds"]):
        for p in ps:
            seq[order[p], order[s]] = 1.0
    return iters, seq, fallback


def step_create(prob_vehicle_order, Hp, keep):
    iters, seq, fallback = prob_vehicle_order
    n = len(iters)
    L = lib()
    arr = (MlIter * max(n, 1))()
    for v, it in enumerate(iters):
        arr[v] = ml_iter(it, Hp, keep)
    seq_f = np.asfortranarray(seq, dtype=np.float64)
    fb_rows = [[(f[k] if f is not None and len(f) else None) for k in range(Hp)] for f in fallback]
    fb = keep.cell(fb_rows, n, Hp)
    keep.refs += [arr, seq_f]
    step = C.c_void_p()
    rc = L.pdmpc_ml_step_create(Hp, n, arr, seq_f.ctypes.data_as(_dp), fb, C.byref(step))
    if rc != 0:
        raise RuntimeError("pdmpc_ml_step_create: %d %s" % (rc, L.pdmpc_ml_last_error().decode()))
    return step


def step_problem(step):
    """The marshalled problem as raw ABI views: (n, VehicleIn*, pred_offset, pred_index, PolygonSet*, order (1-based), levels)."""
    L = lib()
    n = C.c_int32()
    vin = C.POINTER(abi.VehicleIn)()
    po, pi, order, levels = abi.c_int32_p(), abi.c_int32_p(), abi.c_int32_p(), abi.c_int32_p()
    fb = C.POINTER(abi.PolygonSet)()
    rc = L.pdmpc_ml_step_problem(step, C.byref(n), C.byref(vin), C.byref(po), C.byref(pi), C.byref(fb), C.byref(order), C.byref(levels))
    assert rc == 0
    return n.value, vin, po, pi, fb, order, levels


def plan_step(handle, step, n, weights=None):
    """pdmpc_ml_plan_step (weights: pdmpc_ml_plan_step_weighted, expected work per VEHICLE) on a backend.Handle -> records in VEHICLE order."""
    L = lib()
    out = abi.out_array(n)
    if weights is None:
        rc = L.pdmpc_ml_plan_step(handle.h, step, abi.out_ptr(out))
    else:
        w = np.ascontiguousarray(weights, dtype=np.float64)
        L.pdmpc_ml_plan_step_weighted.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(abi.VehicleOut)]
        rc = L.pdmpc_ml_plan_step_weighted(handle.h, step, w.ctypes.data_as(C.c_void_p), abi.out_ptr(out))
    if rc != 0:
        raise RuntimeError("pdmpc_ml_plan_step: %d %s" % (rc, L.pdmpc_ml_last_error().decode()))
    return out[:n]


def record_arrays(rec, Hp):
    """pdmpc_ml_record_arrays -> dict of numpy arrays shaped as MATLAB sees them."""
    L = lib()
    out = {
        "predicted_trims": np.zeros((1, Hp), order="F"), "shape_cols": np.zeros((1, Hp), order="F"), "y_predicted": np.zeros((Hp, 3), order="F"),
        "shapes": np.zeros((Hp, 2, abi.VMAX), order="F"), "path_nodes": np.zeros((Hp + 1, 8), order="F"), "tree_path": np.zeros((1, Hp + 1), order="F"),
    }
    r = np.ascontiguousarray(rec).reshape(1)
    L.pdmpc_ml_record_arrays(r.ctypes.data_as(C.c_void_p), Hp, *[out[k].ctypes.data_as(_dp) for k in ("predicted_trims", "shape_cols", "y_predicted", "shapes", "path_nodes", "tree_path")])
    return out
