// ORACLE (test infrastructure only — never linked into the product library).
// Scalar math conventions of the CPU restatement. Header-only, C99/C++.
//
// The reference calls libm cosf/sinf (src/ORBextractor.cc:113) and OpenCV's cv::fastAtan2 /
// cvRound (src/ORBextractor.cc:81,103,115,119-120). libm's last-ulp behaviour depends on the
// host (glibc ifunc picks FMA or non-FMA builds), so the project pins ONE convention:
//   sinf/cosf  = the double-precision polynomial algorithm glibc >= 2.28 uses
//                (ARM optimized-routines sincosf), evaluated with separate mul/add (no FMA);
//   fastAtan2  = OpenCV 3.2 f32 polynomial, separate mul/add;
//   cvRound    = round-half-to-even (Appendix A.1).
// `oracle/sincosf_check.c` compares pso_sinf/pso_cosf with the host libm for every float in
// [0, 2*pi]; tests/test_oracle_math.py runs a strided version of that check.
#ifndef PSL_MATH_ORACLE_H
#define PSL_MATH_ORACLE_H
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif

static inline int pso_cvround(double v) { return (int)nearbyint(v); }  // default rounding mode = half-even
static inline int pso_cvfloor(double v) { int i = (int)v; return i - (i > v); }
static inline int pso_cvceil(double v) { int i = (int)v; return i + (i < v); }

static inline float pso_fast_atan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    const float eps = (float)2.2204460492503131e-16;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ---- sinf / cosf ------------------------------------------------------------------------------
static const double PSO_HPI_INV = 0x1.45F306DC9C883p+23;  // 2/pi * 2^24
static const double PSO_HPI = 0x1.921FB54442D18p0;        // pi/2
static const double PSO_C0 = 0x1p0, PSO_C1 = -0x1.ffffffd0c621cp-2, PSO_C2 = 0x1.55553e1068f19p-5,
                    PSO_C3 = -0x1.6c087e89a359dp-10, PSO_C4 = 0x1.99343027bf8c3p-16;
static const double PSO_S1 = -0x1.555545995a603p-3, PSO_S2 = 0x1.1107605230bc4p-7,
                    PSO_S3 = -0x1.994eb3774cf24p-13;

static inline uint32_t pso_abstop12(float x) { uint32_t u; memcpy(&u, &x, 4); return (u >> 20) & 0x7ff; }

// n even: sine polynomial of x; n odd: cosine polynomial; neg != 0 negates the cosine result.
static inline float pso_sincos_poly(double x, double x2, int n, int neg) {
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = PSO_S2 + x2 * PSO_S3;
        double x7 = x3 * x2;
        double s = x + x3 * PSO_S1;
        return (float)(s + x7 * s1);
    } else {
        double sg = neg ? -1.0 : 1.0;
        double x4 = x2 * x2;
        double c2 = sg * PSO_C3 + x2 * (sg * PSO_C4);
        double c1 = sg * PSO_C0 + x2 * (sg * PSO_C1);
        double x6 = x4 * x2;
        double c = c1 + x4 * (sg * PSO_C2);
        return (float)(c + x6 * c2);
    }
}

static inline double pso_reduce_fast(double x, int* np) {
    double r = x * PSO_HPI_INV;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return x - n * PSO_HPI;
}

// Valid for |y| < 120 (the path only produces [0, 2*pi]).
static inline float pso_sinf(float y) {
    static const double sign[4] = {1.0, -1.0, -1.0, 1.0};
    double x = y;
    if (pso_abstop12(y) < pso_abstop12(0x1.921FB6p-1f)) {
        if (pso_abstop12(y) < pso_abstop12(0x1p-12f)) return y;
        return pso_sincos_poly(x, x * x, 0, 0);
    }
    int n;
    x = pso_reduce_fast(x, &n);
    return pso_sincos_poly(x * sign[n & 3], x * x, n, n & 2);
}

static inline float pso_cosf(float y) {
    static const double sign[4] = {1.0, -1.0, -1.0, 1.0};
    double x = y;
    if (pso_abstop12(y) < pso_abstop12(0x1.921FB6p-1f)) {
        if (pso_abstop12(y) < pso_abstop12(0x1p-12f)) return 1.0f;
        return pso_sincos_poly(x, x * x, 1, 0);
    }
    int n;
    x = pso_reduce_fast(x, &n);
    return pso_sincos_poly(x * sign[n & 3], x * x, n ^ 1, n & 2);
}

#endif
