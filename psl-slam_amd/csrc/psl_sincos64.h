// (float)cos(x) and (float)sin(x) of a double x in [0, 2 pi] - what k_lsd_grad tabulates for region seeds
// (region_grow: `sumdx = cos(reg_angle)` with a double angle, OpenCV 3.x lsd.cpp, summed in float).  Product code.
// The general-range f64 cos / sin of the device library cost ~250 instructions per pixel here; for this range a
// Cody-Waite reduction by pi/2 (two constants; the argument is a float number of degrees times a constant, so it never
// comes closer to a multiple of pi/2 than ~1e-9) and the fdlibm kernel polynomials with fused multiply-adds are enough:
// the result is accurate to ~1 ulp of f64, and only its rounding to f32 is used.  oracle/sincos64_check.c compares it
// with this host's libm for EVERY float number of degrees in [0, 360]: 0 mismatches.
// Plain C so that the check program can include it: set PSL_SC64_QUAL to the function qualifiers first.
#ifndef PSL_SINCOS64_H
#define PSL_SINCOS64_H

#ifndef PSL_SC64_QUAL
#define PSL_SC64_QUAL static inline
#endif

// cos / sin of x in [0, ~4 pi] to about 1 ulp of f64 (n <= 8: n * pio2_1 is exact)
PSL_SC64_QUAL void psl_cos_sin_f64(double x, double* c, double* s) {
    const double n = __builtin_rint(x * 6.36619772367581382433e-01);
    double r = __builtin_fma(-n, 1.57079632673412561417e+00, x);
    r = __builtin_fma(-n, 6.07710050650619224932e-11, r);
    const double z = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double sr = __builtin_fma(z * r, ps, r);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double cr = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
    const int q = (int)n & 3;
    const double cq = (q & 1) ? sr : cr, sq = (q & 1) ? cr : sr;
    *c = (q == 1 || q == 2) ? -cq : cq;
    *s = (q >= 2) ? -sq : sq;
}

PSL_SC64_QUAL void psl_cos_sin_2pi_f32(double x, float* c, float* s) {
    const double n = __builtin_rint(x * 6.36619772367581382433e-01);                 // x * 2/pi
    double r = __builtin_fma(-n, 1.57079632673412561417e+00, x);                     // pio2_1: first 33 bits of pi/2
    r = __builtin_fma(-n, 6.07710050650619224932e-11, r);                            // pio2_1t: pi/2 - pio2_1
    const double z = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double sr = __builtin_fma(z * r, ps, r);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double cr = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
    const int q = (int)n & 3;
    const double cq = (q & 1) ? sr : cr, sq = (q & 1) ? cr : sr;
    *c = (float)((q == 1 || q == 2) ? -cq : cq);
    *s = (float)((q >= 2) ? -sq : sq);
}

#endif
