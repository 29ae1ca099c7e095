/* pdmpc_math.h — deterministic double-precision sin/cos shared by host and device.
 *
 * Why this exists: the reference evaluates cos(yaw)/sin(yaw) with MATLAB's libm in
 * expand_node.m:50-51 and GraphSearch.m:155-156.  glibc's sin/cos and ROCm's ocml
 * sin/cos differ in the last ulp for some arguments; a 1-ulp change in a pose moves a
 * cost by 1 ulp, which flips the pop order among near-tied search nodes and produces a
 * mirrored trajectory.  Bit-identical results on CPU and GPU therefore need ONE
 * implementation compiled for both sides with floating-point contraction disabled
 * (-ffp-contract=off): every operation below is a single IEEE-754 double add/mul, so
 * host gcc and device hipcc produce the same bits.
 *
 * Algorithm: the classic fdlibm scheme (Sun Microsystems, 1993, freely distributable):
 * Cody-Waite reduction by pi/2 in up to three stages (valid for |x| < 2^20*pi/2, far
 * beyond any vehicle yaw) followed by degree-13/14 minimax polynomials on [-pi/4, pi/4].
 * Error < 1 ulp; tests/test_math.py checks it against libm.
 *
 * Everything is `static inline` and header-only.  PDMPC_HD expands to
 * `__host__ __device__` when compiled by hipcc.
 */
#ifndef PDMPC_MATH_H
#define PDMPC_MATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PDMPC_HD __host__ __device__
#else
#define PDMPC_HD
#endif

PDMPC_HD static inline uint32_t pdmpc_hi_word(double x) {
    uint64_t u;
    memcpy(&u, &x, sizeof u);
    return (uint32_t)(u >> 32);
}

PDMPC_HD static inline double pdmpc_from_hi_word(uint32_t hi) {
    uint64_t u = ((uint64_t)hi) << 32;
    double x;
    memcpy(&x, &u, sizeof x);
    return x;
}

/* sin on [-pi/4, pi/4]; (x + y) is the reduced argument, y the tail. */
PDMPC_HD static inline double pdmpc_ksin(double x, double y, int have_tail) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    uint32_t ix = pdmpc_hi_word(x) & 0x7fffffffu;
    if (ix < 0x3e400000u) { /* |x| < 2^-27 */
        if ((int)x == 0) return x;
    }
    double z = x * x;
    double v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    if (!have_tail) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

/* cos on [-pi/4, pi/4]. */
PDMPC_HD static inline double pdmpc_kcos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    uint32_t ix = pdmpc_hi_word(x) & 0x7fffffffu;
    if (ix < 0x3e400000u) {
        if ((int)x == 0) return 1.0;
    }
    double z = x * x;
    double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    if (ix < 0x3FD33333u) return 1.0 - (0.5 * z - (z * r - x * y));
    double qx;
    if (ix > 0x3fe90000u)
        qx = 0.28125;
    else
        qx = pdmpc_from_hi_word(ix - 0x00200000u); /* about |x|/4 */
    double hz = 0.5 * z - qx;
    double a = 1.0 - qx;
    return a - (hz - (z * r - x * y));
}

/* Reduce x to y0 + y1 in [-pi/4, pi/4]; returns the quadrant n mod 4 (as signed n). */
PDMPC_HD static inline int pdmpc_rem_pio2(double x, double* y0, double* y1) {
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    const double pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    const double pio2_3 = 2.02226624871116645580e-21, pio2_3t = 8.47842766036889956997e-32;
    uint32_t hx = pdmpc_hi_word(x);
    uint32_t ix = hx & 0x7fffffffu;
    if (ix <= 0x3fe921fbu) { /* |x| <= pi/4 */
        *y0 = x;
        *y1 = 0.0;
        return 0;
    }
    double t = (hx >> 31) ? -x : x;
    int n = (int)(t * invpio2 + 0.5);
    double fn = (double)n;
    double r = t - fn * pio2_1;
    double w = fn * pio2_1t;
    int j = (int)(ix >> 20);
    double z0 = r - w;
    int i = j - (int)((pdmpc_hi_word(z0) >> 20) & 0x7ffu);
    if (i > 16) {
        t = r;
        w = fn * pio2_2;
        r = t - w;
        w = fn * pio2_2t - ((t - r) - w);
        z0 = r - w;
        i = j - (int)((pdmpc_hi_word(z0) >> 20) & 0x7ffu);
        if (i > 49) {
            t = r;
            w = fn * pio2_3;
            r = t - w;
            w = fn * pio2_3t - ((t - r) - w);
            z0 = r - w;
        }
    }
    double z1 = (r - z0) - w;
    if (hx >> 31) {
        *y0 = -z0;
        *y1 = -z1;
        return -n;
    }
    *y0 = z0;
    *y1 = z1;
    return n;
}

/* s = sin(x), c = cos(x).  NaN/Inf give NaN.  Accurate for |x| < 2^20*pi/2 (~1.6e6 rad). */
PDMPC_HD static inline void pdmpc_sincos(double x, double* s, double* c) {
    uint32_t ix = pdmpc_hi_word(x) & 0x7fffffffu;
    if (ix >= 0x7ff00000u) {
        *s = x - x;
        *c = x - x;
        return;
    }
    double y0, y1;
    int n = pdmpc_rem_pio2(x, &y0, &y1);
    double sn = pdmpc_ksin(y0, y1, 1);
    double cs = pdmpc_kcos(y0, y1);
    switch (n & 3) {
        case 0: *s = sn; *c = cs; break;
        case 1: *s = cs; *c = -sn; break;
        case 2: *s = -sn; *c = -cs; break;
        default: *s = -cs; *c = sn; break;
    }
}

#endif /* PDMPC_MATH_H */
