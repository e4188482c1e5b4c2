// Plain-C text (so that oracle/atanf_check.c can include it with host stand-ins for the PSL_F* operations and PSL_HD):
// included by psl_device_math.h.  Product code.
#ifndef PSL_ATANF_H
#define PSL_ATANF_H
// libm atanf / atan2f as the reference calls them in MergeLines / MergeTwoLines / convertVec4fToKeyLine
// (Thirdparty/line_descriptor .. uselongline.cpp:61-334, add_src/LineExtractor.cpp:411-447): glibc's float implementations
// (sysdeps/ieee754/flt-32/s_atanf.c, e_atan2f.c - the fdlibm algorithm, plain f32 arithmetic, no multiarch variant), restated.
// tests/test_oracle_line_cpu.py compares psl_atanf with this host's libm for EVERY float and psl_atan2f on 2e8 pairs.
PSL_HD float psl_atanf(float x) {
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    uint32_t hx;
    __builtin_memcpy(&hx, &x, 4);
    const uint32_t ix = hx & 0x7fffffffu;
    int id;
    if (ix >= 0x4c000000u) {  // |x| >= 2^25
        if (ix > 0x7f800000u) return PSL_FADD(x, x);
        return (hx >> 31) ? PSL_FSUB(-atanhi[3], atanlo[3]) : PSL_FADD(atanhi[3], atanlo[3]);
    }
    if (ix < 0x3ee00000u) {  // |x| < 0.4375
        if (ix < 0x31000000u) return x;  // |x| < 2^-29
        id = -1;
    } else {
        x = __builtin_fabsf(x);
        if (ix < 0x3f980000u) {  // |x| < 1.1875
            if (ix < 0x3f300000u) { id = 0; x = PSL_FDIV(PSL_FSUB(PSL_FMUL(2.0f, x), 1.0f), PSL_FADD(2.0f, x)); }
            else { id = 1; x = PSL_FDIV(PSL_FSUB(x, 1.0f), PSL_FADD(x, 1.0f)); }
        } else {
            if (ix < 0x401c0000u) { id = 2; x = PSL_FDIV(PSL_FSUB(x, 1.5f), PSL_FADD(1.0f, PSL_FMUL(1.5f, x))); }
            else { id = 3; x = PSL_FDIV(-1.0f, x); }
        }
    }
    const float z = PSL_FMUL(x, x), w = PSL_FMUL(z, z);
    const float s1 = PSL_FMUL(z, PSL_FADD(aT0, PSL_FMUL(w, PSL_FADD(aT2, PSL_FMUL(w, PSL_FADD(aT4, PSL_FMUL(w, PSL_FADD(aT6, PSL_FMUL(w, PSL_FADD(aT8, PSL_FMUL(w, aT10)))))))))));
    const float s2 = PSL_FMUL(w, PSL_FADD(aT1, PSL_FMUL(w, PSL_FADD(aT3, PSL_FMUL(w, PSL_FADD(aT5, PSL_FMUL(w, PSL_FADD(aT7, PSL_FMUL(w, aT9)))))))));
    if (id < 0) return PSL_FSUB(x, PSL_FMUL(x, PSL_FADD(s1, s2)));
    const float r = PSL_FSUB(atanhi[id], PSL_FSUB(PSL_FSUB(PSL_FMUL(x, PSL_FADD(s1, s2)), atanlo[id]), x));
    return (hx >> 31) ? -r : r;
}

PSL_HD float psl_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    int32_t hx, hy;
    __builtin_memcpy(&hx, &x, 4);
    __builtin_memcpy(&hy, &y, 4);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return PSL_FADD(x, y);
    if (hx == 0x3f800000) return psl_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);  // 2 * sign(x) + sign(y)
    if (iy == 0) {
        if (m < 2) return y;
        return m == 2 ? PSL_FADD(pi, tiny) : PSL_FSUB(-pi, tiny);
    }
    if (ix == 0) return hy < 0 ? PSL_FSUB(-pi_o_2, tiny) : PSL_FADD(pi_o_2, tiny);
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
                case 0: return PSL_FADD(pi_o_4, tiny);
                case 1: return PSL_FSUB(-pi_o_4, tiny);
                case 2: return PSL_FADD(PSL_FMUL(3.0f, pi_o_4), tiny);
                default: return PSL_FSUB(PSL_FMUL(-3.0f, pi_o_4), tiny);
            }
        }
        switch (m) {
            case 0: return 0.0f;
            case 1: return -0.0f;
            case 2: return PSL_FADD(pi, tiny);
            default: return PSL_FSUB(-pi, tiny);
        }
    }
    if (iy == 0x7f800000) return hy < 0 ? PSL_FSUB(-pi_o_2, tiny) : PSL_FADD(pi_o_2, tiny);
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = PSL_FADD(pi_o_2, PSL_FMUL(0.5f, pi_lo));
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = psl_atanf(__builtin_fabsf(PSL_FDIV(y, x)));
    switch (m) {
        case 0: return z;
        case 1: return -z;
        case 2: return PSL_FSUB(pi, PSL_FSUB(z, pi_lo));
        default: return PSL_FSUB(PSL_FSUB(z, pi_lo), pi);
    }
}

#endif
