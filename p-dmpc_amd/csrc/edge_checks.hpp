// Edge validity (GraphSearch.m:111-196): InterX and separating-axis checks of one swept area against the obstacle soups (device code, included by search_kernel.hip inside its anonymous namespace).
#pragma once

// ---------------------------------------------------------------------------------------------------
// are_constraints_satisfied_interx.m:17-37 + InterX.m:63-76,108-110.
// The three soups (vehicle obstacles of step k -> shape A, HDV sets of step k -> shape A, lanelet boundary ->
// boundary-check shape B) are processed together.  hit(i, j) = C1(i, j) & C2(i, j) with
//   C1 = (dx1_i*y2_j - dy1_i*x2_j - S1_i) * (dx1_i*y2_{j+1} - dy1_i*x2_{j+1} - S1_i) < 0
//   C2 = (y1_i*dx2_j - x1_i*dy2_j - S2_j) * (y1_{i+1}*dx2_j - x1_{i+1}*dy2_j - S2_j) < 0     (strict; NaN -> false)
// One lane per obstacle segment j: C2 for every shape segment i gives a 7-bit mask; a lane with a non-zero mask evaluates
// C1 for those pairs right away.  The check ends with the first round of 64 segments that holds a hit (most edges of a
// blocked search collide: they leave after one or two of the two or three rounds their soups make).
// interx_segment_n: one obstacle segment (q0, q1) against the shape edges (pt[i], pt[i + 1]), i < ne <= NP - 1.
template <int NP>
__device__ __forceinline__ bool interx_segment_n(const d2 (&pt)[NP], int ne, d2 q0, d2 q1) {
    const double dx2 = q1.x - q0.x, dy2 = q1.y - q0.y;
    const double S2 = dx2 * q0.y - dy2 * q0.x;
    uint32_t bits = 0;
    double e0 = (pt[0].y * dx2 - pt[0].x * dy2) - S2;
#pragma unroll
    for (int i = 0; i < NP - 1; ++i) {
        const double e1 = (pt[i + 1].y * dx2 - pt[i + 1].x * dy2) - S2;
        if (i < ne && e0 * e1 < 0) bits |= 1u << i;
        e0 = e1;
    }
    bool hit = false;
    if (bits) {
#pragma unroll
        for (int i = 0; i < NP - 1; ++i) {
            if ((bits >> i) & 1u) {
                const d2 p0 = pt[i], p1 = pt[i + 1];
                const double dx1 = p1.x - p0.x, dy1 = p1.y - p0.y;
                const double S1 = dx1 * p0.y - dy1 * p0.x;
                const double a0 = dx1 * q0.y - dy1 * q0.x;
                const double a1 = dx1 * q1.y - dy1 * q1.x;
                hit = hit || ((a0 - S1) * (a1 - S1) < 0);
            }
        }
    }
    return hit;
}

// interx_segment: one obstacle segment against the whole shape pt[0 .. V-1].
__device__ __forceinline__ bool interx_segment(const d2 (&pt)[PDMPC_VMAX], int V, d2 q0, d2 q1) { return interx_segment_n<PDMPC_VMAX>(pt, V - 1, q0, q1); }

// sh2 holds shape A in [0, VMAX) and shape B in [VMAX, 2*VMAX).
__device__ bool interx_check(const lds_d2* sh2, int V, const lds_d2* soup, int so, int M_k, int ho, int Hk, int lo, int Ml, int lane) {
    if (V < 2) return false;
    // The segments of all three soups form one index space: ceil(total / 64) rounds instead of one set of rounds per soup
    // (the soups are short: ~60 obstacle points and ~30 boundary points make two rounds instead of two plus a nearly empty
    // third; a lane picks its soup and the shape that goes with it).
    const int n0 = M_k > 1 ? M_k - 1 : 0, n1 = Hk > 1 ? Hk - 1 : 0, n2 = Ml > 1 ? Ml - 1 : 0;
    const int T = n0 + n1 + n2;
    for (int base = 0; base < T; base += PDMPC_WAVE) {
        const int t = base + lane;
        bool hit = false;
        if (t < T) {
            int j;
            uint32_t shapeB = 0;
            if (t < n0) {
                j = so + t;
            } else if (t < n0 + n1) {
                j = ho + (t - n0);
            } else {
                j = lo + (t - n0 - n1);
                shapeB = 1;
            }
            const lds_d2* sh = sh2 + shapeB * PDMPC_VMAX;
            const d2 q0 = soup[j], q1 = soup[j + 1];
            // (all of the shape's points are read up front, so their LDS reads are issued together instead of one round trip
            // per edge; columns beyond V are read but never used)
            d2 pt[PDMPC_VMAX];
#pragma unroll
            for (int i = 0; i < PDMPC_VMAX; ++i) pt[i] = sh[i];
            hit = interx_segment(pt, V, q0, q1);
        }
        if (wave_any(hit)) return true;
    }
    return false;
}

// intersect_sat.m:1-42 for shape (V1 points) vs one polygon o (V2 points): one lane per separating axis.
// An axis separates iff min1 - max2 > 0 or min2 - max1 > 0 (:33-40); a zero-length edge gives a NaN axis whose
// comparisons are false.  collide <=> no axis of either polygon separates.
__device__ bool sat_pair_wave(const lds_d2* sh, int V1, const lds_d2* o, int V2, int lane) {
    const int A = V1 + V2;
    for (int base = 0; base < A; base += PDMPC_WAVE) {
        const int a = base + lane;
        bool sep = false;
        if (a < A) {
            d2 e0, e1;
            if (a < V1) {
                e0 = sh[a];
                e1 = sh[(a + 1 == V1) ? 0 : a + 1];
            } else {
                const int b = a - V1;
                e0 = o[b];
                e1 = o[(b + 1 == V2) ? 0 : b + 1];
            }
            const double ex = e1.x - e0.x, ey = e1.y - e0.y;
            const double ax = -ey, ay = ex;
            const double nrm = sqrt(ax * ax + ay * ay);
            const double nx = ax / nrm, ny = ay / nrm;
            double minS = 0, maxS = 0, minO = 0, maxO = 0;
            for (int v = 0; v < V1; ++v) {
                const d2 p = sh[v];
                const double d = nx * p.x + ny * p.y;
                if (v == 0) {
                    minS = d;
                    maxS = d;
                } else {
                    minS = (d < minS) ? d : minS;
                    maxS = (d > maxS) ? d : maxS;
                }
            }
            for (int v = 0; v < V2; ++v) {
                const d2 p = o[v];
                const double d = nx * p.x + ny * p.y;
                if (v == 0) {
                    minO = d;
                    maxO = d;
                } else {
                    minO = (d < minO) ? d : minO;
                    maxO = (d > maxO) ? d : maxO;
                }
            }
            sep = (minS - maxO > 0) || (minO - maxS > 0);
        }
        if (wave_any(sep)) return false;
    }
    return true;
}

// are_constraints_satisfied_sat.m:15-35: every polygon of the step's soup (static then dynamic obstacles).
__device__ bool sat_soup_wave(const lds_d2* sh, int V1, const lds_d2* soup, int M, int lane) {
    int pos = 0;
    while (pos < M) {
        int end = M;  // next NaN separator at or after pos
        for (int base = pos; base < M; base += PDMPC_WAVE) {
            const int j = base + lane;
            const bool sepr = (j < M) && is_nan(soup[j].x);
            const unsigned long long b = __ballot(sepr);
            if (b) {
                end = base + (int)__builtin_ctzll(b);
                break;
            }
        }
        const int V2 = end - pos;
        if (V2 > 0 && sat_pair_wave(sh, V1, soup + pos, V2, lane)) return true;
        pos = end + 1;
    }
    return false;
}

// intersect_lanelet_boundary.m:1-56 on the soup [left, NaN, right, NaN]: one lane per boundary segment,
// AABB pre-filter (:20,40) then intersect_sat(shape, segment) with the segment as a 2-point polygon.
__device__ bool sat_boundary_wave(const lds_d2* sh, int V1, const lds_d2* ll, int M, int lane) {
    if (M < 2) return false;
    double max_x = sh[0].x, min_x = sh[0].x, max_y = sh[0].y, min_y = sh[0].y;
    for (int v = 1; v < V1; ++v) {
        const d2 p = sh[v];
        max_x = (p.x > max_x) ? p.x : max_x;
        min_x = (p.x < min_x) ? p.x : min_x;
        max_y = (p.y > max_y) ? p.y : max_y;
        min_y = (p.y < min_y) ? p.y : min_y;
    }
    for (int base = 0; base < M - 1; base += PDMPC_WAVE) {
        const int j = base + lane;
        bool hit = false;
        if (j < M - 1) {
            const d2 q0 = ll[j], q1 = ll[j + 1];
            const bool real = !(is_nan(q0.x) || is_nan(q1.x));
            const bool reject = (max_x < q0.x && max_x < q1.x) || (min_x > q0.x && min_x > q1.x) ||
                                (max_y < q0.y && max_y < q1.y) || (min_y > q0.y && min_y > q1.y);
            if (real && !reject) {
                bool sep = false;
                const int A = V1 + 2;
                for (int a = 0; a < A; ++a) {
                    d2 e0, e1;
                    if (a < V1) {
                        e0 = sh[a];
                        e1 = sh[(a + 1 == V1) ? 0 : a + 1];
                    } else if (a == V1) {
                        e0 = q0;
                        e1 = q1;
                    } else {
                        e0 = q1;
                        e1 = q0;
                    }
                    const double ex = e1.x - e0.x, ey = e1.y - e0.y;
                    const double ax = -ey, ay = ex;
                    const double nrm = sqrt(ax * ax + ay * ay);
                    const double nx = ax / nrm, ny = ay / nrm;
                    double minS = 0, maxS = 0;
                    for (int v = 0; v < V1; ++v) {
                        const d2 p = sh[v];
                        const double d = nx * p.x + ny * p.y;
                        if (v == 0) {
                            minS = d;
                            maxS = d;
                        } else {
                            minS = (d < minS) ? d : minS;
                            maxS = (d > maxS) ? d : maxS;
                        }
                    }
                    const double d0 = nx * q0.x + ny * q0.y;
                    const double d1 = nx * q1.x + ny * q1.y;
                    const double minO = (d1 < d0) ? d1 : d0;
                    const double maxO = (d1 > d0) ? d1 : d0;
                    sep = sep || (minS - maxO > 0) || (minO - maxS > 0);
                }
                hit = !sep;
            }
        }
        if (wave_any(hit)) return true;
    }
    return false;
}

// read-only view of what an edge check needs (shared by the sequencing wave and the helper waves)
struct CheckCtx {
    const lds_d2* l_area;
    const d2* g_area;
    const lds_d2* l_soup;
    const lds_i32* l_soff;
    const lds_i32* l_hoff;
    int areas_in_lds, ll_base, ll_len, Hp, checker;
    lds_d2* sh;     // this wave's shape scratch: A in [0, VMAX), B in [VMAX, 2 VMAX)
    lds_u32* cand;  // this wave's candidate list
    LDS_AS unsigned long long* tally;  // this wave's work counters: edge checks, (shape segment, obstacle segment) pairs
};

// eval_edge_exact (GraphSearch.m:111-196) for node `id` (1-based): true = collision-free.  A pure function of the
// tree and the obstacle soups, which is what allows helper waves to evaluate it ahead of the pop.
// (cn = the node's record, pn = its parent's; the same record in every lane)
template <int CHECKER>
__device__ bool edge_valid_recs(const CheckCtx& C, const NodeRec& cn, const NodeRec& pn, int lane) {
    const uint32_t par = uni_u(cn.parent);
    if (!par) return true;  // root: no edge (GraphSearch.m:137-139)
    const uint32_t cpk = uni_u(cn.packed);
    const int cK = NODE_K(cpk);
    const double pX = pn.x, pY = pn.y;
    const double c = pn.cs, s = pn.sn;  // cos/sin(pYaw), cached when the parent was expanded
    const int m = NODE_MAN(cpk);
    const int ncols = NODE_COLS(cpk);
    if (lane < ncols) {
        const size_t ai = (size_t)m * 3 * PDMPC_VMAX + lane;
        const size_t bi = ai + (size_t)((cK == C.Hp) ? 2 : 1) * PDMPC_VMAX;  // large offset at k == Hp, else without offset
        d2 a, b;
        if (C.areas_in_lds) {
            a = C.l_area[ai];
            b = C.l_area[bi];
        } else {
            a = C.g_area[ai];
            b = C.g_area[bi];
        }
        d2 sa, sb;
        sa.x = c * a.x - s * a.y + pX;  // GraphSearch.m:158
        sa.y = s * a.x + c * a.y + pY;  // :159
        sb.x = c * b.x - s * b.y + pX;  // :162 / :168
        sb.y = s * b.x + c * b.y + pY;  // :163 / :169
        C.sh[lane] = sa;
        C.sh[PDMPC_VMAX + lane] = sb;
    }
    wave_sync();
    const int so = uni_i(C.l_soff[cK - 1]);
    const int M_k = uni_i(C.l_soff[cK]) - so;
    bool hit;
    if (CHECKER == PDMPC_CHECK_INTERX) {
        const int ho = uni_i(C.l_hoff[cK - 1]);
        const int Hk = uni_i(C.l_hoff[cK]) - ho;
        if (lane == 0) {  // the pairs the reference's InterX forms for this edge (InterX.m:63-76): (V - 1) x (M - 1) per soup
            C.tally[0] += 1;
            C.tally[1] += (unsigned long long)(ncols - 1) * (unsigned long long)((M_k > 1 ? M_k - 1 : 0) + (Hk > 1 ? Hk - 1 : 0) + (C.ll_len > 1 ? C.ll_len - 1 : 0));
        }
        hit = interx_check(C.sh, ncols, C.l_soup, so, M_k, ho, Hk, C.ll_base, C.ll_len, lane);
    } else {
        // are_constraints_satisfied_sat.m:15-53
        if (lane == 0) C.tally[0] += 1;
        hit = sat_soup_wave(C.sh, ncols, C.l_soup + so, M_k, lane);
        if (!hit) hit = sat_boundary_wave(C.sh + PDMPC_VMAX, ncols, C.l_soup + C.ll_base, C.ll_len, lane);
    }
    wave_sync();
    return !hit;
}

template <int CHECKER>
__device__ bool edge_valid(const Search& S, const CheckCtx& C, uint32_t id, int lane) {
    const NodeRec cn = node_load(S, id - 1);
    const uint32_t par = uni_u(cn.parent);
    if (!par) return true;
    const NodeRec pn = node_load(S, par - 1);
    return edge_valid_recs<CHECKER>(C, cn, pn, lane);
}

// ---------------------------------------------------------------------------------------------------
// The separating-axis checker in the form the bulk kernel uses (bulk_kernel.hip, bk_check_items): ONE LANE tests one pair — the
// edge's area (its points in registers) against one polygon of the soup, or against one boundary segment —, axis by axis with the
// arithmetic of sat_pair_wave / sat_boundary_wave above (intersect_sat.m:19-40, intersect_lanelet_boundary.m:16-54): the same
// operations on the same numbers, so the same bits.  The loops over the area's points are unrolled with compile-time indices
// (the points are a register array); the polygon's points are read from LDS.

// projections of the area's V1 points on the axis (nx, ny): min and max (intersect_sat.m:26-29)
__device__ __forceinline__ void sat_project_area(const d2 (&pt)[PDMPC_VMAX], int V1, double nx, double ny, double& mn, double& mx) {
    mn = 0.0;
    mx = 0.0;
#pragma unroll
    for (int v = 0; v < PDMPC_VMAX; ++v) {
        const double d = nx * pt[v].x + ny * pt[v].y;
        if (v == 0) {
            mn = d;
            mx = d;
        } else if (v < V1) {
            mn = (d < mn) ? d : mn;
            mx = (d > mx) ? d : mx;
        }
    }
}
// does the axis normal to the edge e0 -> e1 separate the area from the polygon o[0 .. V2)?  (:21-23, 33-40; a zero-length edge gives a
// NaN axis whose comparisons are false)
__device__ __forceinline__ bool sat_axis_separates(const d2 (&pt)[PDMPC_VMAX], int V1, const lds_d2* o, int V2, d2 e0, d2 e1) {
    const double ex = e1.x - e0.x, ey = e1.y - e0.y;
    const double ax = -ey, ay = ex;
    const double nrm = sqrt(ax * ax + ay * ay);
    const double nx = ax / nrm, ny = ay / nrm;
    double minS, maxS, minO = 0, maxO = 0;
    sat_project_area(pt, V1, nx, ny, minS, maxS);
    for (int v = 0; v < V2; ++v) {
        const d2 p = o[v];
        const double d = nx * p.x + ny * p.y;
        if (v == 0) {
            minO = d;
            maxO = d;
        } else {
            minO = (d < minO) ? d : minO;
            maxO = (d > maxO) ? d : maxO;
        }
    }
    return (minS - maxO > 0) || (minO - maxS > 0);
}
// point i of the area (i known at run time only: the points live in registers, so the pick is a chain of selects, not an indexed read)
__device__ __forceinline__ d2 sat_pick(const d2 (&pt)[PDMPC_VMAX], int i) {
    d2 r = pt[0];
#pragma unroll
    for (int v = 1; v < PDMPC_VMAX; ++v) {
        r.x = (v == i) ? pt[v].x : r.x;
        r.y = (v == i) ? pt[v].y : r.y;
    }
    return r;
}
// intersect_sat(area, o): true iff no axis of either polygon separates them.  ONE loop over the V1 + V2 axes (not unrolled: eight
// copies of the projection side by side cost the kernel sixty spilled registers), edges of closed polygons wrap to the first point.
__device__ __forceinline__ bool sat_pair_lane(const d2 (&pt)[PDMPC_VMAX], int V1, const lds_d2* o, int V2) {
    bool sep = false;
#pragma unroll 1
    for (int a = 0; a < V1 + V2 && !sep; ++a) {
        d2 e0, e1;
        if (a < V1) {
            e0 = sat_pick(pt, a);
            e1 = sat_pick(pt, (a + 1 == V1) ? 0 : a + 1);
        } else {
            const int b = a - V1;
            e0 = o[b];
            e1 = o[(b + 1 == V2) ? 0 : b + 1];
        }
        sep = sat_axis_separates(pt, V1, o, V2, e0, e1);
    }
    return !sep;
}
// intersect_lanelet_boundary for ONE boundary segment (q0, q1): the bounding-box reject (:20, 40), then intersect_sat(area, segment)
// with the segment as a two-point polygon (:24, 44).  bb = (min_x, max_x, min_y, max_y) of the area.
__device__ __forceinline__ void sat_area_bbox(const d2 (&pt)[PDMPC_VMAX], int V1, double& min_x, double& max_x, double& min_y, double& max_y) {
    max_x = pt[0].x;
    min_x = pt[0].x;
    max_y = pt[0].y;
    min_y = pt[0].y;
#pragma unroll
    for (int v = 1; v < PDMPC_VMAX; ++v) {
        if (v < V1) {
            const d2 p = pt[v];
            max_x = (p.x > max_x) ? p.x : max_x;
            min_x = (p.x < min_x) ? p.x : min_x;
            max_y = (p.y > max_y) ? p.y : max_y;
            min_y = (p.y < min_y) ? p.y : min_y;
        }
    }
}
__device__ __forceinline__ bool sat_boundary_segment_lane(const d2 (&pt)[PDMPC_VMAX], int V1, double min_x, double max_x, double min_y, double max_y, d2 q0, d2 q1) {
    const bool real = !(is_nan(q0.x) || is_nan(q1.x));
    const bool reject = (max_x < q0.x && max_x < q1.x) || (min_x > q0.x && min_x > q1.x) || (max_y < q0.y && max_y < q1.y) || (min_y > q0.y && min_y > q1.y);
    if (!real || reject) return false;
    bool sep = false;
#pragma unroll 1
    for (int a = 0; a < V1 + 2; ++a) {  // the area's edges, then the segment both ways (intersect_sat.m:19 on a two-point polygon)
        d2 e0, e1;
        if (a < V1) {
            e0 = sat_pick(pt, a);
            e1 = sat_pick(pt, (a + 1 == V1) ? 0 : a + 1);
        } else {
            e0 = a == V1 ? q0 : q1;
            e1 = a == V1 ? q1 : q0;
        }
        const double ex = e1.x - e0.x, ey = e1.y - e0.y;
        const double ax = -ey, ay = ex;
        const double nrm = sqrt(ax * ax + ay * ay);
        const double nx = ax / nrm, ny = ay / nrm;
        double minS, maxS;
        sat_project_area(pt, V1, nx, ny, minS, maxS);
        const double d0 = nx * q0.x + ny * q0.y;
        const double d1 = nx * q1.x + ny * q1.y;
        const double minO = (d1 < d0) ? d1 : d0;
        const double maxO = (d1 > d0) ? d1 : d0;
        sep = sep || (minS - maxO > 0) || (minO - maxS > 0);
    }
    return !sep;
}
