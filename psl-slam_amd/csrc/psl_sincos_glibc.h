// Double-precision sin / cos as glibc >= 2.28 evaluates them (sysdeps/ieee754/dbl-64/s_sin.c: __sin, __cos, do_sin, do_cos,
// reduce_sincos), restated for |x| < 105414350: a 111-entry table of sin / cos at k/128 as double-doubles (regenerated from the
// series by tools/gen_sincostab.py), two short polynomials, and the four-constant Cody-Waite reduction by pi/2.  Product code:
// MergeTwoLines' sin / cos of the merged direction (add_src/uselongline.cpp:320-329) and the rectangle's cos / sin of theta
// (OpenCV lsd.cpp region2rect) - both feed f64 arithmetic whose result can cancel (an end point on the image border) or be
// truncated to int (rect_nfa's corners), so "within an ulp" is not enough there.  oracle/f64math_check.c (mode sincos) compares
// these functions with the host's libm on 6e7 arguments of the ranges used: they must be BIT-IDENTICAL (the CPU suite runs it).
// Plain C text (the check program includes it): set PSL_SC64_QUAL to the function qualifiers first; `tab` = the 444 doubles of
// psl_sincostab.inc (a device pointer on the device).  Single IEEE operations only: build with -ffp-contract=off.
#ifndef PSL_SINCOS_GLIBC_H
#define PSL_SINCOS_GLIBC_H

#ifndef PSL_SC64_QUAL
#define PSL_SC64_QUAL static inline
#endif

PSL_SC64_QUAL double psl_sg_taylor_sin(double xx, double x, double dx) {
    const double s1 = -0x1.5555555555555p-3, s2 = 0x1.1111111110ECEp-7, s3 = -0x1.A01A019DB08B8p-13, s4 = 0x1.71DE27B9A7ED9p-19,
                 s5 = -0x1.ADDFFC2FCDF59p-26;
    const double t = ((((((s5 * xx + s4) * xx + s3) * xx + s2) * xx) + s1) * x - 0.5 * dx) * xx + dx;
    return x + t;
}

PSL_SC64_QUAL int psl_sg_index(double u) {   // low word of big + |x|: |x| rounded to a multiple of 1/128, times 128
    unsigned long long b;
    __builtin_memcpy(&b, &u, 8);
    return (int)(unsigned)b * 4;
}

PSL_SC64_QUAL double psl_sg_do_sin(double x, double dx, const double* tab) {
    const double big = 0x1.8000000000000p45, sn3 = -1.66666666666664880952546298448555E-01, sn5 = 8.33333214285722277379541354343671E-03,
                 cs2 = 4.99999999999999999999950396842453E-01, cs4 = -4.16666666666664434524222570944589E-02,
                 cs6 = 1.38888874007937613028114285595617E-03;
    const double xold = x;
    if (__builtin_fabs(x) < 0.126) return psl_sg_taylor_sin(x * x, x, dx);
    if (x <= 0) dx = -dx;
    const double u = big + __builtin_fabs(x);
    x = __builtin_fabs(x) - (u - big);
    const int k = psl_sg_index(u);
    const double xx = x * x, s = x + (dx + x * xx * (sn3 + xx * sn5)), c = x * dx + xx * (cs2 + xx * (cs4 + xx * cs6));
    const double sn = tab[k], ssn = tab[k + 1], cs = tab[k + 2], ccs = tab[k + 3];
    const double cor = (ssn + s * ccs - sn * c) + cs * s;
    return __builtin_copysign(sn + cor, xold);
}

PSL_SC64_QUAL double psl_sg_do_cos(double x, double dx, const double* tab) {
    const double big = 0x1.8000000000000p45, sn3 = -1.66666666666664880952546298448555E-01, sn5 = 8.33333214285722277379541354343671E-03,
                 cs2 = 4.99999999999999999999950396842453E-01, cs4 = -4.16666666666664434524222570944589E-02,
                 cs6 = 1.38888874007937613028114285595617E-03;
    if (x < 0) dx = -dx;
    const double u = big + __builtin_fabs(x);
    x = __builtin_fabs(x) - (u - big) + dx;
    const int k = psl_sg_index(u);
    const double xx = x * x, s = x + x * xx * (sn3 + xx * sn5), c = xx * (cs2 + xx * (cs4 + xx * cs6));
    const double sn = tab[k], ssn = tab[k + 1], cs = tab[k + 2], ccs = tab[k + 3];
    const double cor = (ccs - s * ssn - cs * c) - sn * s;
    return cs + cor;
}

// x - n pi/2 as a + da, n mod 4 (+ k): |x| < 105414350
PSL_SC64_QUAL int psl_sg_reduce(double x, double* a, double* da, int k) {
    const double hpinv = 0.63661977236758138, toint = 6755399441055744.0, mp1 = 1.5707963407039642, mp2 = -1.3909067564377153e-08,
                 pp3 = -4.9789962314799099e-17, pp4 = -1.9034889620193266e-25;
    const double t = (x * hpinv + toint), xn = t - toint;
    unsigned long long vb;
    __builtin_memcpy(&vb, &t, 8);
    const double y = (x - xn * mp1) - xn * mp2;
    const int n = ((int)(unsigned)vb + k) & 3;
    double t1 = xn * pp3;
    const double t2 = y - t1;
    double db = (y - t2) - t1;
    t1 = xn * pp4;
    const double b = t2 - t1;
    db += (t2 - b) - t1;
    *a = b; *da = db;
    return n;
}

PSL_SC64_QUAL double psl_sg_do_sincos(double a, double da, int n, const double* tab) {
    const double r = (n & 1) ? psl_sg_do_cos(a, da, tab) : psl_sg_do_sin(a, da, tab);
    return (n & 2) ? -r : r;
}

PSL_SC64_QUAL double psl_glibc_sin(double x, const double* tab) {
    const double hp0 = 1.5707963267948966, hp1 = 6.123233995736766e-17;
    const double ax = __builtin_fabs(x);
    if (ax < 0x1p-26) return x;
    if (ax < 0.855469) return psl_sg_do_sin(x, 0, tab);
    if (ax < 2.426265) return __builtin_copysign(psl_sg_do_cos(hp0 - ax, hp1, tab), x);
    double a, da;
    const int n = psl_sg_reduce(x, &a, &da, 0);
    return psl_sg_do_sincos(a, da, n, tab);
}

PSL_SC64_QUAL double psl_glibc_cos(double x, const double* tab) {
    const double hp0 = 1.5707963267948966, hp1 = 6.123233995736766e-17;
    const double ax = __builtin_fabs(x);
    if (ax < 0x1p-27) return 1.0;
    if (ax < 0.855469) return psl_sg_do_cos(x, 0, tab);
    if (ax < 2.426265) {
        const double y = hp0 - ax, a = y + hp1, da = (y - a) + hp1;
        return psl_sg_do_sin(a, da, tab);
    }
    double a, da;
    const int n = psl_sg_reduce(x, &a, &da, 1);
    return psl_sg_do_sincos(a, da, n, tab);
}

#endif
