// Double-precision log / exp / log10 / pow and float tanf for the two places of the line path where the reference calls
// libm on values that decide something:
//   * the NFA validation of LSD_REFINE_ADV (OpenCV 3.x lsd.cpp nfa() / log_gamma(), twin in the reference tree:
//     Thirdparty/line_descriptor/src/ED_Lib/NFA.cpp:106-240) - log, exp, log10, pow, sinh in double;
//   * `abs(tan(arcAng)) > 1` of CPartiallyRecoverConnectivity (add_src/PartiallyRecoverConnectivity.cpp:39) - tanf.
// Product code.  Plain-C text so that the CPU suite can compile it for the host (tests/test_f64math_cpu.py builds
// oracle/f64math_check.c around it): set PSL_F64_QUAL to the function qualifiers first.  Every operation is a single
// IEEE operation - the library and the check program are both built with -ffp-contract=off - so host and device results
// are bit-identical by construction.
//
// psl_tanf is glibc's float tanf (sysdeps/ieee754/flt-32/s_tanf.c, k_tanf.c: the fdlibm kernel in plain f32 arithmetic, with
// glibc >= 2.28's double-precision argument reduction) restated for |x| < 120; the check program compares it with the host's libm for EVERY float in
// [0, 8] (the argument is a float number of degrees in [0, 360] times pi/180): it must be bit-identical.
// psl_log / psl_exp / psl_log10 are the fdlibm algorithms (e_log.c, e_exp.c, e_log10.c).  glibc >= 2.28 uses table-driven
// algorithms whose tables are not reproducible offline, so these are pinned as "within 1 ulp of the host's libm" on dense
// samples of the ranges the NFA uses (and bit-identical on most of them); DESIGN.md §3 explains why a last-ulp difference
// can only flip an NFA decision on an exact tie.
#ifndef PSL_F64MATH_H
#define PSL_F64MATH_H

#include <stdint.h>

#ifndef PSL_F64_QUAL
#define PSL_F64_QUAL static inline
#endif

PSL_F64_QUAL uint64_t psl_f64_bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
PSL_F64_QUAL double psl_f64_from(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
PSL_F64_QUAL uint32_t psl_f32_bits(float x) { uint32_t u; __builtin_memcpy(&u, &x, 4); return u; }
PSL_F64_QUAL float psl_f32_from(uint32_t u) { float x; __builtin_memcpy(&x, &u, 4); return x; }

// natural logarithm, x > 0 finite (fdlibm e_log.c)
PSL_F64_QUAL double psl_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, two54 = 1.80143985094819840000e+16,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u = psl_f64_bits(x);
    int32_t hx = (int32_t)(u >> 32);
    int32_t k = 0;
    if (hx < 0x00100000) {  // subnormal (x > 0 is a precondition)
        k -= 54; x *= two54; u = psl_f64_bits(x); hx = (int32_t)(u >> 32);
    }
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int32_t i = (hx + 0x95f64) & 0x100000;
    u = ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32) | (u & 0xffffffffu);  // normalize x or x/2
    x = psl_f64_from(u);
    k += (i >> 20);
    const double f = x - 1.0;
    if ((0x000fffff & (2 + hx)) < 3) {  // |f| < 2**-20
        if (f == 0.0) {
            if (k == 0) return 0.0;
            const double dk = (double)k;
            return dk * ln2_hi + dk * ln2_lo;
        }
        const double R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        const double dk = (double)k;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    const double s = f / (2.0 + f);
    const double dk = (double)k;
    const double z = s * s;
    int32_t ii = hx - 0x6147a;
    const double w = z * z;
    const int32_t j = 0x6b851 - hx;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    ii |= j;
    const double R = t2 + t1;
    if (ii > 0) {
        const double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

// x * 2^n for results that stay normal or underflow gradually (fdlibm scalbn, the cases exp needs)
PSL_F64_QUAL double psl_scalbn(double x, int n) {
    if (n > 1023) { x *= 8.98846567431157953865e+307; n -= 1023; if (n > 1023) { x *= 8.98846567431157953865e+307; n -= 1023; if (n > 1023) n = 1023; } }
    else if (n < -1022) { x *= 2.22507385850720138309e-308 * 9007199254740992.0; n += 1022 - 53; if (n < -1022) { x *= 2.22507385850720138309e-308 * 9007199254740992.0; n += 1022 - 53; if (n < -1022) n = -1022; } }
    return x * psl_f64_from((uint64_t)(0x3ff + n) << 52);
}

// e^x (fdlibm e_exp.c)
PSL_F64_QUAL double psl_exp(double x) {
    const double o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02,
                 ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00,
                 P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    const uint64_t u = psl_f64_bits(x);
    const int xsb = (int)(u >> 63);
    const uint32_t hx = (uint32_t)(u >> 32) & 0x7fffffffu;
    double hi = 0, lo = 0;
    int k = 0;
    if (hx >= 0x40862E42u) {
        if (hx >= 0x7ff00000u) return (u & 0x000fffffffffffffull) ? x + x : (xsb ? 0.0 : x);
        if (x > o_threshold) return 1.0e+300 * 1.0e+300;
        if (x < u_threshold) return 1.0e-300 * 1.0e-300;
    }
    if (hx > 0x3fd62e42u) {  // |x| > 0.5 ln2
        if (hx < 0x3FF0A2B2u) {  // and |x| < 1.5 ln2
            hi = x - (xsb ? -ln2HI : ln2HI); lo = xsb ? -ln2LO : ln2LO; k = 1 - xsb - xsb;
        } else {
            k = (int)(invln2 * x + (xsb ? -0.5 : 0.5));
            const double t = (double)k;
            hi = x - t * ln2HI;
            lo = t * ln2LO;
        }
        x = hi - lo;
    } else if (hx < 0x3e300000u) {  // |x| < 2**-28
        return 1.0 + x;
    }
    const double t = x * x;
    const double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    const double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    return psl_scalbn(y, k);
}

// log10(x), x > 0 finite (fdlibm e_log10.c)
PSL_F64_QUAL double psl_log10(double x) {
    const double two54 = 1.80143985094819840000e+16, ivln10 = 4.34294481903251816668e-01,
                 log10_2hi = 3.01029995663611771306e-01, log10_2lo = 3.69423907715893078616e-13;
    uint64_t u = psl_f64_bits(x);
    int32_t hx = (int32_t)(u >> 32);
    int32_t k = 0;
    if (hx < 0x00100000) { k -= 54; x *= two54; u = psl_f64_bits(x); hx = (int32_t)(u >> 32); }
    k += (hx >> 20) - 1023;
    const int32_t i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
    hx = (hx & 0x000fffff) | ((0x3ff - i) << 20);
    const double y = (double)(k + i);
    x = psl_f64_from(((uint64_t)(uint32_t)hx << 32) | (u & 0xffffffffu));
    const double z = y * log10_2lo + ivln10 * psl_log(x);
    return z + y * log10_2hi;
}

// x^y for x > 0 as exp(y log x).  Only used for the truncation test of the binomial tail (`err`, nfa()), where the value
// enters as 1 - x^y with x^y << 1: the relative error of this form, |y log x| ulp, is far below what that test can see.
PSL_F64_QUAL double psl_pow_pos(double x, double y) { return psl_exp(y * psl_log(x)); }

// (double)a / (double)b, correctly rounded, from r = the correctly rounded 1 / b (a table): quotient estimate, its exact residual, one
// correction - three operations instead of the ~11 of a hardware division sequence.  Equal to the division for EVERY pair
// 0 <= a < PSL_RATIO_AMAX, 1 <= b < PSL_RATIO_BMAX: oracle/f64math_check.c (mode ratio) runs all 1.07e9 of them (the CPU suite).
// The binomial tail of nfa() divides (n - i + 1) by i once per term (k_lsd_nfa_series).
#define PSL_RATIO_AMAX 65536
#define PSL_RATIO_BMAX 16384
PSL_F64_QUAL double psl_ratio_inv(double a, double b, double r) {
    const double q0 = a * r, e = __builtin_fma(-q0, b, a);
    return __builtin_fma(e, r, q0);
}

// sinh(u) for 0 < u <= 1/15 (log_gamma_windschitl's sinh(1/x), x > 15): odd Taylor polynomial, truncation error < 1e-19 relative
PSL_F64_QUAL double psl_sinh_small(double u) {
    const double z = u * u;
    const double p = 1.0 / 6.0 + z * (1.0 / 120.0 + z * (1.0 / 5040.0 + z * (1.0 / 362880.0 + z * (1.0 / 39916800.0))));
    return u + u * z * p;
}

// ------------------------------------------------------------------ tanf (glibc flt-32: s_tanf.c, k_tanf.c, e_rem_pio2f.c)
PSL_F64_QUAL float psl_kernel_tanf(float x, float y, int iy) {
    const float pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
    const float T0 = 3.3333334327e-01f, T1 = 1.3333334029e-01f, T2 = 5.3968254477e-02f, T3 = 2.1869488060e-02f, T4 = 8.8632395491e-03f,
                T5 = 3.5920790397e-03f, T6 = 1.4562094584e-03f, T7 = 5.8804126456e-04f, T8 = 2.4646313977e-04f, T9 = 7.8179444245e-05f,
                T10 = 7.1407252108e-05f, T11 = -1.8558637748e-05f, T12 = 2.5907305826e-05f;
    const int32_t hx = (int32_t)psl_f32_bits(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix < 0x39000000) {  // |x| < 2**-13
        if ((int)x == 0) {
            if ((ix | (iy + 1)) == 0) return 1.0f / __builtin_fabsf(x);
            else if (iy == 1) return x;
            else return -1.0f / x;
        }
    }
    if (ix >= 0x3f2ca140) {  // |x| >= 0.6744
        if (hx < 0) { x = -x; y = -y; }
        const float z0 = pio4 - x;
        const float w0 = pio4lo - y;
        x = z0 + w0;
        y = 0.0f;
        if (__builtin_fabsf(x) < 0x1p-13f) return (float)(1 - ((hx >> 30) & 2)) * (float)iy * (1.0f - 2.0f * (float)iy * x);
    }
    float z = x * x;
    float w = z * z;
    float r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
    float v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
    float s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T0 * s;
    w = x + r;
    if (ix >= 0x3f2ca140) {
        v = (float)iy;
        return (float)(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
    }
    if (iy == 1) return w;
    // -1 / (x + r) accurately
    z = psl_f32_from(psl_f32_bits(w) & 0xfffff000u);
    v = r - (z - x);
    const float a = -1.0f / w;
    const float t = psl_f32_from(psl_f32_bits(a) & 0xfffff000u);
    s = 1.0f + t * z;
    return t + a * (s + t * v);
}

// x - n * pi/2 as y0 + y1, |x| < 120: glibc >= 2.28's __ieee754_rem_pio2f is the double-precision reduction of its sinf / cosf
// (sysdeps/ieee754/flt-32/e_rem_pio2f.c -> reduce_fast, s_sincosf.h): the quadrant from a product pre-scaled by 2^24
PSL_F64_QUAL int psl_rem_pio2f(float x, float* y0, float* y1) {
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double dx = (double)x;
    const double r = dx * hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    const double red = dx - (double)n * hpi;
    *y0 = (float)red;
    *y1 = (float)(red - (double)*y0);
    return n;
}

// tanf(x) for |x| < 120
PSL_F64_QUAL float psl_tanf(float x) {
    const int32_t ix = (int32_t)(psl_f32_bits(x) & 0x7fffffffu);
    if (ix <= 0x3f490fda) return psl_kernel_tanf(x, 0.0f, 1);  // |x| ~<= pi/4
    float y0, y1;
    const int n = psl_rem_pio2f(x, &y0, &y1);
    return psl_kernel_tanf(y0, y1, 1 - ((n & 1) << 1));  // n even: tan, n odd: -1/tan
}

#endif
