"""The oracle's OWN binding of include/pdmpc.h: struct definitions and packers written independently of the product's
`pdmpc.abi`, straight from the header's field comments.  TEST INFRASTRUCTURE ONLY.

Why a second packer: if both the HIP backend and the oracle were fed by the same marshalling code, a wrong index there
(say the polygon order of `dynamic_obstacles`, documented in pdmpc.h as `i * Hp + (k - 1)`) would be invisible to every
parity test.  Here the two sides only share the C header.  Style is deliberately different from pdmpc.abi: one flat pool of
doubles per call and explicit index arithmetic instead of per-polygon copies.
"""
import ctypes as ct

import numpy as np

HP_MAX, VMAX = 16, 8
_dptr, _iptr, _bptr = ct.POINTER(ct.c_double), ct.POINTER(ct.c_int32), ct.POINTER(ct.c_uint8)


class OConfig(ct.Structure):  # pdmpc_config
    _fields_ = [("Hp", ct.c_int32), ("checker", ct.c_int32), ("dt_seconds", ct.c_double), ("device", ct.c_int32), ("max_nodes", ct.c_int32),
                ("max_vehicles", ct.c_int32), ("trace_pops", ct.c_int32)]


class OManeuver(ct.Structure):  # pdmpc_maneuver
    _fields_ = [("dx", ct.c_double), ("dy", ct.c_double), ("dyaw", ct.c_double), ("n_cols", ct.c_int32), ("_pad", ct.c_int32),
                ("area", ct.c_double * (2 * VMAX)), ("area_without_offset", ct.c_double * (2 * VMAX)), ("area_large_offset", ct.c_double * (2 * VMAX))]


class OMpa(ct.Structure):  # pdmpc_mpa
    _fields_ = [("n_trims", ct.c_int32), ("Hp", ct.c_int32), ("transition", _bptr), ("maneuver_index", _iptr), ("n_maneuvers", ct.c_int32),
                ("maneuvers", ct.POINTER(OManeuver))]


class OPolygonSet(ct.Structure):  # pdmpc_polygon_set
    _fields_ = [("n_polygons", ct.c_int32), ("offset", _iptr), ("x", _dptr), ("y", _dptr)]


class OVehicleIn(ct.Structure):  # pdmpc_vehicle_in
    _fields_ = [("x0", ct.c_double), ("y0", ct.c_double), ("yaw0", ct.c_double), ("trim0", ct.c_int32), ("n_left", ct.c_int32), ("n_right", ct.c_int32),
                ("_pad", ct.c_int32), ("ref_x", _dptr), ("ref_y", _dptr), ("v_ref", _dptr), ("left_x", _dptr), ("left_y", _dptr), ("right_x", _dptr),
                ("right_y", _dptr), ("obstacles", OPolygonSet), ("dynamic_obstacles", OPolygonSet), ("hdv_reachable_sets", OPolygonSet)]


# pdmpc_vehicle_out as a numpy record (field order and sizes from the header)
OUT_DTYPE = np.dtype([
    ("status", "<i4"), ("n_expanded", "<i4"), ("n_popped", "<i4"), ("n_hp", "<i4"), ("tree_path", "<i4", (HP_MAX + 1,)),
    ("predicted_trims", "<i4", (HP_MAX,)), ("shape_cols", "<i4", (HP_MAX,)), ("_pad", "<i4"), ("y_predicted", "<f8", (HP_MAX, 3)),
    ("shapes", "<f8", (HP_MAX, 2, VMAX)), ("path_nodes", "<f8", (HP_MAX + 1, 8)),
])


class _Pool:
    """One growing pool of doubles and one of int32 per call; structs point into them (the pools keep everything alive)."""

    def __init__(self):
        self.d = []
        self.i = []
        self.fix = []  # (struct, field, pool, start) resolved once the pools are frozen

    def put_d(self, values):
        start = sum(len(v) for v in self.d)
        self.d.append(np.asarray(values, dtype=np.float64).ravel())
        return start

    def put_i(self, values):
        start = sum(len(v) for v in self.i)
        self.i.append(np.asarray(values, dtype=np.int32).ravel())
        return start

    def freeze(self):
        self.dbuf = np.concatenate(self.d + [np.zeros(1)])
        self.ibuf = np.concatenate(self.i + [np.zeros(1, dtype=np.int32)])
        dbase, ibase = self.dbuf.ctypes.data, self.ibuf.ctypes.data
        for struct, field, pool, start in self.fix:
            if pool == "d":
                setattr(struct, field, ct.cast(dbase + 8 * start, _dptr))
            else:
                setattr(struct, field, ct.cast(ibase + 4 * start, _iptr))


def _polygon_set(pool, target, polygons):
    """polygons: sequence of 2 x V arrays, stored back to back; offset[p] = first column of polygon p."""
    cols = [np.asarray(p, dtype=np.float64) for p in polygons]
    for p in cols:
        assert p.ndim == 2 and p.shape[0] == 2
    counts = [p.shape[1] for p in cols]
    offsets = np.concatenate([[0], np.cumsum(counts)]) if cols else np.zeros(1)
    xs = np.concatenate([p[0] for p in cols]) if cols else np.zeros(0)
    ys = np.concatenate([p[1] for p in cols]) if cols else np.zeros(0)
    target.n_polygons = len(cols)
    pool.fix.append((target, "offset", "i", pool.put_i(offsets)))
    pool.fix.append((target, "x", "d", pool.put_d(xs)))
    pool.fix.append((target, "y", "d", pool.put_d(ys)))


def pack_vehicles(iters, Hp):
    """list of VehicleIter-like objects -> (OVehicleIn array, pool).  Field meanings per include/pdmpc.h:99-115."""
    n = len(iters)
    arr = (OVehicleIn * max(n, 1))()
    pool = _Pool()
    for v, it in enumerate(iters):
        s = arr[v]
        s.x0, s.y0, s.yaw0 = (float(it.x0[q]) for q in range(3))  # iter.x0(1, 1:3)
        s.trim0 = int(it.trim_index)
        ref = np.asarray(it.reference_trajectory_points, dtype=np.float64)
        assert ref.shape == (Hp, 2) and len(it.v_ref) == Hp
        pool.fix.append((s, "ref_x", "d", pool.put_d(ref[:, 0])))
        pool.fix.append((s, "ref_y", "d", pool.put_d(ref[:, 1])))
        pool.fix.append((s, "v_ref", "d", pool.put_d(it.v_ref)))
        for side, tag in ((it.predicted_lanelet_boundary[0], "left"), (it.predicted_lanelet_boundary[1], "right")):
            pts = np.zeros((2, 0)) if side is None or np.size(side) == 0 else np.asarray(side, dtype=np.float64)
            setattr(s, "n_" + tag, pts.shape[1])
            pool.fix.append((s, tag + "_x", "d", pool.put_d(pts[0])))
            pool.fix.append((s, tag + "_y", "d", pool.put_d(pts[1])))
        _polygon_set(pool, s.obstacles, it.obstacles)
        # n_d x Hp cell -> polygon index i * Hp + (k - 1): obstacle-major, step-minor
        dyn = [it.dynamic_obstacle_area[i][k] for i in range(len(it.dynamic_obstacle_area)) for k in range(Hp)]
        _polygon_set(pool, s.dynamic_obstacles, dyn)
        hdv = [it.hdv_reachable_sets[i][k] for i in range(len(it.hdv_reachable_sets)) for k in range(Hp)]
        _polygon_set(pool, s.hdv_reachable_sets, hdv)
    pool.freeze()
    pool.arr = arr
    return arr, pool


def pack_mpa(mpa):
    """MotionPrimitiveAutomaton-like object -> (OMpa, keep-alive tuple).  transition[k][i][j] = transition_matrix_single(i+1, j+1, k+1)."""
    n, Hp = int(mpa.n_trims), int(mpa.Hp)
    T = np.asarray(mpa.transition_matrix_single)
    trans = np.zeros((Hp, n, n), dtype=np.uint8)
    for k in range(Hp):
        trans[k] = T[:, :, k] != 0
    index = np.full((n, n), -1, dtype=np.int32)
    mans = []
    for i in range(n):
        for j in range(n):
            if mpa.maneuvers[i][j] is not None:
                index[i, j] = len(mans)
                mans.append(mpa.maneuvers[i][j])
    arr = (OManeuver * max(len(mans), 1))()
    for q, m in enumerate(mans):
        s = arr[q]
        s.dx, s.dy, s.dyaw = float(m.dx), float(m.dy), float(m.dyaw)
        cols = int(np.asarray(m.area).shape[1])
        s.n_cols = cols
        for name in ("area", "area_without_offset", "area_large_offset"):
            a = np.asarray(getattr(m, name), dtype=np.float64)
            dst = getattr(s, name)  # [2][VMAX] row-major: row 0 = x, row 1 = y
            for c in range(cols):
                dst[c] = a[0, c]
                dst[VMAX + c] = a[1, c]
    trans = np.ascontiguousarray(trans)
    out = OMpa(n, Hp, trans.ctypes.data_as(_bptr), index.ctypes.data_as(_iptr), len(mans), arr)
    return out, (trans, index, arr)


def out_array(n):
    return np.zeros(max(n, 1), dtype=OUT_DTYPE)
