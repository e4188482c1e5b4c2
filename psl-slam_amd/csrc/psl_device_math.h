// Scalar math shared by the HIP kernels (and the library's host-side table builders).
// Product code: does not include anything from oracle/.
//
// Every operation that feeds a rounding decision is written with explicitly rounded, never
// contracted, single operations (HIP __fmul_rn / __fadd_rn / __dmul_rn / __dadd_rn on the
// device), so that results are bit-identical to a plain IEEE host evaluation:
//   psl_fast_atan2  cv::fastAtan2, OpenCV 3.2 f32 polynomial (src/ORBextractor.cc:103,
//                   add_src/PartiallyRecoverConnectivity.cpp:37,58)
//   psl_sinf/cosf   libm sinf/cosf as called at src/ORBextractor.cc:113: the double-precision
//                   polynomial algorithm of glibc >= 2.28, valid for |x| < 120
//   psl_cvround     cvRound = round-half-to-even (src/ORBextractor.cc:81,115,119-120)
#ifndef PSL_DEVICE_MATH_H
#define PSL_DEVICE_MATH_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

#if defined(__HIP_DEVICE_COMPILE__)
#define PSL_FMUL(a, b) __fmul_rn((a), (b))
#define PSL_FADD(a, b) __fadd_rn((a), (b))
#define PSL_FSUB(a, b) __fsub_rn((a), (b))
#define PSL_FDIV(a, b) __fdiv_rn((a), (b))
#define PSL_DMUL(a, b) __dmul_rn((a), (b))
#define PSL_DADD(a, b) __dadd_rn((a), (b))
#define PSL_DSUB(a, b) __dsub_rn((a), (b))
#else
#define PSL_FMUL(a, b) ((a) * (b))
#define PSL_FADD(a, b) ((a) + (b))
#define PSL_FSUB(a, b) ((a) - (b))
#define PSL_FDIV(a, b) ((a) / (b))
#define PSL_DMUL(a, b) ((a) * (b))
#define PSL_DADD(a, b) ((a) + (b))
#define PSL_DSUB(a, b) ((a) - (b))
#endif

#define PSL_HD __host__ __device__ static inline

PSL_HD int psl_cvround_f(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float2int_rn(v);
#else
    return (int)__builtin_nearbyintf(v);
#endif
}

PSL_HD int psl_cvround_d(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __double2int_rn(v);
#else
    return (int)__builtin_nearbyint(v);
#endif
}

PSL_HD float psl_fast_atan2(float y, float x) {
    const float k = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * k;   // constant-folded in f32, as in OpenCV
    const float p3 = -0.3258083974640975f * k;
    const float p5 = 0.1555786518463281f * k;
    const float p7 = -0.04432655554792128f * k;
    const float eps = (float)2.2204460492503131e-16;
    float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = PSL_FDIV(ay, PSL_FADD(ax, eps));
        c2 = PSL_FMUL(c, c);
        a = PSL_FMUL(PSL_FADD(PSL_FMUL(PSL_FADD(PSL_FMUL(PSL_FADD(PSL_FMUL(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = PSL_FDIV(ax, PSL_FADD(ay, eps));
        c2 = PSL_FMUL(c, c);
        a = PSL_FSUB(90.f, PSL_FMUL(PSL_FADD(PSL_FMUL(PSL_FADD(PSL_FMUL(PSL_FADD(PSL_FMUL(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = PSL_FSUB(180.f, a);
    if (y < 0) a = PSL_FSUB(360.f, a);
    return a;
}

// n even: sine polynomial of x; n odd: cosine polynomial (negated when neg != 0).
PSL_HD float psl_sincos_poly(double x, double x2, int n, int neg) {
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5,
                 C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        double x3 = PSL_DMUL(x, x2);
        double s1 = PSL_DADD(S2, PSL_DMUL(x2, S3));
        double x7 = PSL_DMUL(x3, x2);
        double s = PSL_DADD(x, PSL_DMUL(x3, S1));
        return (float)PSL_DADD(s, PSL_DMUL(x7, s1));
    } else {
        const double sg = neg ? -1.0 : 1.0;
        double x4 = PSL_DMUL(x2, x2);
        double c2 = PSL_DADD(sg * C3, PSL_DMUL(x2, sg * C4));
        double c1 = PSL_DADD(sg * C0, PSL_DMUL(x2, sg * C1));
        double x6 = PSL_DMUL(x4, x2);
        double c = PSL_DADD(c1, PSL_DMUL(x4, sg * C2));
        return (float)PSL_DADD(c, PSL_DMUL(x6, c2));
    }
}

PSL_HD uint32_t psl_abstop12(float x) {
    union { float f; uint32_t u; } v;
    v.f = x;
    return (v.u >> 20) & 0x7ff;
}

PSL_HD double psl_reduce_fast(double x, int* np) {
    const double HPI_INV = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0;
    double r = PSL_DMUL(x, HPI_INV);
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return PSL_DSUB(x, PSL_DMUL((double)n, HPI));
}

PSL_HD void psl_sincosf(float y, float* sinp, float* cosp) {
    double x = (double)y;
    if (psl_abstop12(y) < psl_abstop12(0x1.921FB6p-1f)) {
        if (psl_abstop12(y) < psl_abstop12(0x1p-12f)) { *sinp = y; *cosp = 1.0f; return; }
        double x2 = PSL_DMUL(x, x);
        *sinp = psl_sincos_poly(x, x2, 0, 0);
        *cosp = psl_sincos_poly(x, x2, 1, 0);
        return;
    }
    int n;
    x = psl_reduce_fast(x, &n);
    const double sg = ((n + 1) & 2) ? -1.0 : 1.0;  // sign table {1,-1,-1,1}[n & 3]
    double xs = PSL_DMUL(x, sg), x2 = PSL_DMUL(x, x);
    *sinp = psl_sincos_poly(xs, x2, n, n & 2);
    *cosp = psl_sincos_poly(xs, x2, n ^ 1, n & 2);
}

#include "psl_atanf.h"  // psl_atanf, psl_atan2f

#endif
