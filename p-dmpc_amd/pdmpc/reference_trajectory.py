"""Input producers on the caller's side of the optimizer boundary.

Restates, on the host, the reference functions that build the per-step inputs of the search:
    get_reference_trajectory       hlc/controller/common/get_reference_trajectory.m:1-48
    sample_reference_trajectory    hlc/controller/common/sample_reference_trajectory.m:1-103
    get_arc_distance_to_endpoint   hlc/controller/common/get_arc_distance_to_endpoint.m:1-132
    projection_2d                  hlc/controller/common/projection_2d.m:1-44
    get_occupied_areas             hlc/controller/common/get_occupied_areas.m:1-33
These are pure geometry on a handful of points per vehicle and stay on the CPU; their outputs are the
`reference_trajectory_points`, `v_ref` and `occupied_areas` fields of IterationData.
"""
import math

import numpy as np


def projection_2d(x1, y1, x2, y2, x3, y3):
    """projection_2d.m:14-42 -> (xp, yp, projection_distance, lambda, line_segment_len)."""
    b = math.sqrt((x2 - x1) ** 2 + (y2 - y1) ** 2)
    if b != 0:
        xn = (x2 - x1) / b
        yn = (y2 - y1) / b
        x31 = x3 - x1
        y31 = y3 - y1
        dot = xn * x31 + yn * y31
        dist = xn * y31 - yn * x31
        return x1 + dot * xn, y1 + dot * yn, dist, dot / b, b
    return x1, y1, math.sqrt((x3 - x1) ** 2 + (y3 - y1) ** 2), 0.0, b


def get_arc_distance_to_endpoint(px, py, curve_x, curve_y):
    """Subset of get_arc_distance_to_endpoint.m used by the sampler: (x_projected, y_projected, idx_next) with
    idx_next 1-based (:39-114)."""
    n_points = len(curve_x)
    sq = (curve_x - px) ** 2 + (curve_y - py) ** 2
    ic = int(np.argmin(sq))  # 0-based idx_closest
    if ic == 0:  # :47-54
        f, s = 0, 1
    elif ic == n_points - 1:  # :55-62
        f, s = n_points - 2, n_points - 1
    else:  # :63-86
        if sq[ic - 1] <= sq[ic + 1]:  # min() returns the first of equal values -> left
            f, s = ic - 1, ic
        else:
            f, s = ic, ic + 1
    xp, yp, _, lam, _ = projection_2d(curve_x[f], curve_y[f], curve_x[s], curve_y[s], px, py)
    idx_closest = ic + 1  # 1-based from here on
    idx_next = idx_closest
    if (0 <= lam <= 0.5) or lam >= 1:  # :101-109
        idx_next = idx_closest + 1 if idx_closest < n_points else 1
    idx_next = max(2, idx_next)  # :114
    return xp, yp, idx_next


def _norm2(v):
    return math.sqrt(v[0] * v[0] + v[1] * v[1])


def sample_reference_trajectory(n_samples, reference_path, x_current, y_current, step_distances):
    """sample_reference_trajectory.m:1-99 -> (path (n,2), points_index (n,), current_point_index); indices 1-based."""
    ref = np.asarray(reference_path, dtype=np.float64)
    path = np.zeros((n_samples, 2))
    points_index = np.zeros(n_samples, dtype=np.int64)
    xp, yp, point_index = get_arc_distance_to_endpoint(x_current, y_current, ref[:, 0], ref[:, 1])
    current_point_index = point_index
    n_line_pieces = ref.shape[0]
    cur = np.array([xp, yp])
    is_loop = _norm2(ref[0] - ref[-1]) < 1e-8  # :40
    is_vehicle_at_end = point_index == n_line_pieces
    point_index_last = point_index - 1
    if is_loop and is_vehicle_at_end:  # :46-48
        point_index = 1

    def P(i):  # 1-based row access
        return ref[i - 1]

    for i in range(n_samples):
        remaining = _norm2(cur - P(point_index))  # :51
        if remaining > step_distances[i] or point_index == n_line_pieces:  # :53
            while P(point_index)[0] == P(point_index_last)[0] and P(point_index)[1] == P(point_index_last)[1] and point_index_last > 1:  # :56-58
                point_index_last -= 1
            d = P(point_index) - P(point_index_last)
            cur = cur + step_distances[i] * (d / _norm2(d))  # :64
        else:
            reflength = remaining
            while remaining < step_distances[i]:  # :68-87
                reflength = remaining
                cur = P(point_index).copy()
                point_index_last = point_index
                point_index = min(point_index + 1, n_line_pieces)
                is_vehicle_at_end = point_index == n_line_pieces
                if is_loop and is_vehicle_at_end:
                    point_index = 1
                remaining = remaining + _norm2(cur - P(point_index))
            d = P(point_index) - P(point_index_last)
            cur = cur + (step_distances[i] - reflength) * (d / _norm2(d))  # :89
        path[i] = cur
        points_index[i] = point_index
    return path, points_index, current_point_index


def get_reference_trajectory(mpa, reference_path, reference_speed, x_current, y_current, trim_current, dt_seconds):
    """get_reference_trajectory.m:27-46 -> (path (Hp,2), points_index, v_ref (Hp,), current_point_index)."""
    Hp = mpa.Hp
    v_ref = np.ones(Hp) * reference_speed
    v_current = mpa.trims[trim_current - 1].speed
    v_int = (np.concatenate(([v_current], v_ref[:-1])) + v_ref) / 2
    step_distances = v_int * dt_seconds
    path, points_index, cpi = sample_reference_trajectory(Hp, reference_path, x_current, y_current, step_distances)
    return path, points_index, v_ref, cpi


def translate_global(yaw, x0, y0, xl, yl):
    """utility/translate_global.m:19-22."""
    c, s = math.cos(yaw), math.sin(yaw)
    xl = np.asarray(xl, dtype=np.float64)
    yl = np.asarray(yl, dtype=np.float64)
    return c * xl + (-s) * yl + x0, s * xl + c * yl + y0


def get_occupied_areas(x, y, yaw, length, width, offset):
    """get_occupied_areas.m:21-31 -> (normal_offset (2,5), without_offset (2,5)), closed rectangles."""
    sx = np.array([-1, -1, 1, 1, -1.0])
    sy = np.array([-1, 1, 1, -1, -1.0])
    xa, ya = translate_global(yaw, x, y, sx * (length / 2 + offset), sy * (width / 2 + offset))
    xb, yb = translate_global(yaw, x, y, sx * (length / 2), sy * (width / 2))
    return np.vstack([xa, ya]), np.vstack([xb, yb])
