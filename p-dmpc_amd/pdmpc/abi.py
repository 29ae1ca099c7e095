"""ctypes mirror of include/pdmpc.h and numpy -> struct marshalling.

This is the Python analogue of the MEX shim a MATLAB maintainer would write:
it turns the per-vehicle ``IterationData`` slice and the MPA tables into the plain-pointer structs of
the C ABI.  Nothing here computes anything on the hot path.
"""
import ctypes as C

import numpy as np

HP_MAX = 16
VMAX = 8

OK, EXHAUSTED, ARENA_OVERFLOW = 0, 1, 2
CHECK_SAT, CHECK_INTERX = 0, 1

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class Config(C.Structure):
    _fields_ = [
        ("Hp", C.c_int32),
        ("checker", C.c_int32),
        ("dt_seconds", C.c_double),
        ("device", C.c_int32),
        ("max_nodes", C.c_int32),
        ("max_vehicles", C.c_int32),
        ("trace_pops", C.c_int32),
    ]


class Maneuver(C.Structure):
    _fields_ = [
        ("dx", C.c_double),
        ("dy", C.c_double),
        ("dyaw", C.c_double),
        ("n_cols", C.c_int32),
        ("_pad", C.c_int32),
        ("area", (C.c_double * VMAX) * 2),
        ("area_without_offset", (C.c_double * VMAX) * 2),
        ("area_large_offset", (C.c_double * VMAX) * 2),
    ]


class Mpa(C.Structure):
    _fields_ = [
        ("n_trims", C.c_int32),
        ("Hp", C.c_int32),
        ("transition", c_uint8_p),
        ("maneuver_index", c_int32_p),
        ("n_maneuvers", C.c_int32),
        ("maneuvers", C.POINTER(Maneuver)),
    ]


class PolygonSet(C.Structure):
    _fields_ = [
        ("n_polygons", C.c_int32),
        ("offset", c_int32_p),
        ("x", c_double_p),
        ("y", c_double_p),
    ]


class VehicleIn(C.Structure):
    _fields_ = [
        ("x0", C.c_double),
        ("y0", C.c_double),
        ("yaw0", C.c_double),
        ("trim0", C.c_int32),
        ("n_left", C.c_int32),
        ("n_right", C.c_int32),
        ("_pad", C.c_int32),
        ("ref_x", c_double_p),
        ("ref_y", c_double_p),
        ("v_ref", c_double_p),
        ("left_x", c_double_p),
        ("left_y", c_double_p),
        ("right_x", c_double_p),
        ("right_y", c_double_p),
        ("obstacles", PolygonSet),
        ("dynamic_obstacles", PolygonSet),
        ("hdv_reachable_sets", PolygonSet),
    ]


class VehicleOut(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("n_expanded", C.c_int32),
        ("n_popped", C.c_int32),
        ("n_hp", C.c_int32),
        ("tree_path", C.c_int32 * (HP_MAX + 1)),
        ("predicted_trims", C.c_int32 * HP_MAX),
        ("shape_cols", C.c_int32 * HP_MAX),
        ("_pad", C.c_int32),
        ("y_predicted", (C.c_double * 3) * HP_MAX),
        ("shapes", ((C.c_double * VMAX) * 2) * HP_MAX),
        ("path_nodes", (C.c_double * 8) * (HP_MAX + 1)),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("n_vehicles", C.c_int64),
        ("nodes_popped", C.c_int64),
        ("nodes_generated", C.c_int64),
        ("obstacle_columns", C.c_int64),
        ("algorithmic_bytes", C.c_int64),
        ("kernel_ms", C.c_double),
        ("lds_bytes", C.c_int64),
        ("lds_nodes", C.c_int64),
        ("n_launches", C.c_int64),
        ("queue_fallbacks", C.c_int64),
        ("speculation_arrivals", C.c_int64),
        ("edge_checks", C.c_int64),
        ("segment_pair_tests", C.c_int64),
        ("kernel", C.c_int64),
        ("nodes_processed", C.c_int64),
        ("rounds", C.c_int64),
        ("shared_rounds", C.c_int64),
        ("helper_checked", C.c_int64),
        ("safe_replans", C.c_int64),
        ("bad_status_plans", C.c_int64),
    ]


# numpy view of pdmpc_vehicle_out (same memory layout; used for the RCCL exchange and for fast decoding)
VEHICLE_OUT_DTYPE = np.dtype(
    [
        ("status", "<i4"),
        ("n_expanded", "<i4"),
        ("n_popped", "<i4"),
        ("n_hp", "<i4"),
        ("tree_path", "<i4", (HP_MAX + 1,)),
        ("predicted_trims", "<i4", (HP_MAX,)),
        ("shape_cols", "<i4", (HP_MAX,)),
        ("_pad", "<i4"),
        ("y_predicted", "<f8", (HP_MAX, 3)),
        ("shapes", "<f8", (HP_MAX, 2, VMAX)),
        ("path_nodes", "<f8", (HP_MAX + 1, 8)),
    ]
)
assert VEHICLE_OUT_DTYPE.itemsize == C.sizeof(VehicleOut), (VEHICLE_OUT_DTYPE.itemsize, C.sizeof(VehicleOut))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return a.ctypes.data_as(c_double_p)


class _Keep:
    """Holds the numpy buffers a struct points into, so they outlive the struct."""

    def __init__(self):
        self.refs = []

    def f64(self, a):
        a = _f64(a)
        self.refs.append(a)
        return a


def pack_polygon_set(polys, keep):
    """polys: list of (2, V) arrays -> PolygonSet."""
    n = len(polys)
    offs = np.zeros(n + 1, dtype=np.int32)
    for i, p in enumerate(polys):
        p = np.asarray(p, dtype=np.float64)
        if p.ndim != 2 or p.shape[0] != 2:
            raise ValueError("polygon must be a 2 x V array, got %r" % (p.shape,))
        offs[i + 1] = offs[i] + p.shape[1]
    x = np.zeros(max(int(offs[-1]), 1), dtype=np.float64)
    y = np.zeros_like(x)
    for i, p in enumerate(polys):
        p = np.asarray(p, dtype=np.float64)
        x[offs[i] : offs[i + 1]] = p[0]
        y[offs[i] : offs[i + 1]] = p[1]
    keep.refs += [offs, x, y]
    return PolygonSet(n, offs.ctypes.data_as(c_int32_p), _dp(x), _dp(y))


def pack_mpa(mpa):
    """mpa: pdmpc.mpa.MotionPrimitiveAutomaton -> (Mpa struct, keep-alive object)."""
    keep = _Keep()
    n, Hp = mpa.n_trims, mpa.Hp
    # transition_matrix_single is (n, n, Hp) as in MATLAB; the ABI wants [k][i][j]
    trans = np.ascontiguousarray(np.transpose(mpa.transition_matrix_single, (2, 0, 1)).astype(np.uint8))
    index = -np.ones((n, n), dtype=np.int32)
    mans = []
    for i in range(n):
        for j in range(n):
            m = mpa.maneuvers[i][j]
            if m is None:
                continue
            index[i, j] = len(mans)
            mans.append(m)
    arr = (Maneuver * max(len(mans), 1))()
    for q, m in enumerate(mans):
        s = arr[q]
        s.dx, s.dy, s.dyaw = float(m.dx), float(m.dy), float(m.dyaw)
        ncol = m.area.shape[1]
        if ncol > VMAX:
            raise ValueError("maneuver area has %d columns > PDMPC_VMAX" % ncol)
        s.n_cols = ncol
        for name in ("area", "area_without_offset", "area_large_offset"):
            a = getattr(m, name)
            if a.shape != (2, ncol):
                raise ValueError("all three areas of a maneuver must have the same column count")
            dst = getattr(s, name)
            for r in range(2):
                for v in range(ncol):
                    dst[r][v] = float(a[r, v])
    keep.refs += [trans, index, arr]
    out = Mpa(n, Hp, trans.ctypes.data_as(c_uint8_p), index.ctypes.data_as(c_int32_p), len(mans), arr)
    return out, keep


def pack_vehicle(it, Hp, keep, dst):
    """Fill VehicleIn `dst` from a pdmpc.iteration_data.VehicleIter."""
    dst.x0, dst.y0, dst.yaw0 = float(it.x0[0]), float(it.x0[1]), float(it.x0[2])
    dst.trim0 = int(it.trim_index)
    ref = _f64(it.reference_trajectory_points)
    if ref.shape != (Hp, 2):
        raise ValueError("reference_trajectory_points must be (Hp, 2)")
    rx, ry = keep.f64(ref[:, 0]), keep.f64(ref[:, 1])
    vr = keep.f64(it.v_ref)
    if vr.shape != (Hp,):
        raise ValueError("v_ref must have Hp entries")
    dst.ref_x, dst.ref_y, dst.v_ref = _dp(rx), _dp(ry), _dp(vr)
    left = it.predicted_lanelet_boundary[0]
    right = it.predicted_lanelet_boundary[1]
    for side, name in ((left, "left"), (right, "right")):
        if side is None or np.size(side) == 0:
            n = 0
            sx = sy = keep.f64(np.zeros(1))
        else:
            side = _f64(side)
            n = side.shape[1]
            sx, sy = keep.f64(side[0]), keep.f64(side[1])
        setattr(dst, "n_" + name, n)
        setattr(dst, name + "_x", _dp(sx))
        setattr(dst, name + "_y", _dp(sy))
    dst.obstacles = pack_polygon_set(list(it.obstacles), keep)
    dyn = []
    for row in it.dynamic_obstacle_area:
        if len(row) != Hp:
            raise ValueError("dynamic_obstacle_area rows must have Hp entries")
        dyn += list(row)
    dst.dynamic_obstacles = pack_polygon_set(dyn, keep)
    hdv = []
    for row in it.hdv_reachable_sets:
        if len(row) != Hp:
            raise ValueError("hdv_reachable_sets rows must have Hp entries")
        hdv += list(row)
    dst.hdv_reachable_sets = pack_polygon_set(hdv, keep)


def pack_vehicles(iters, Hp):
    keep = _Keep()
    arr = (VehicleIn * max(len(iters), 1))()
    for i, it in enumerate(iters):
        pack_vehicle(it, Hp, keep, arr[i])
    keep.refs.append(arr)
    return arr, keep


def out_array(n):
    """Result buffer: numpy structured array sharing memory with VehicleOut[n]."""
    return np.zeros(max(n, 1), dtype=VEHICLE_OUT_DTYPE)


def out_ptr(arr):
    return arr.ctypes.data_as(C.POINTER(VehicleOut))
