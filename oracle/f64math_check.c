/* Pins psl-slam_amd/csrc/psl_f64math.h against this host's libm (test infrastructure; run by tests/test_f64math_cpu.py).
 *   tanf   : every float in [lo, hi] (default [0, 8]) must be bit-identical to libm's tanf.
 *   log, exp, log10 : dense samples of the ranges nfa() / log_gamma() use; reports the largest difference in ulps
 *            and the share of samples that differ at all (glibc's own algorithms are table-driven and not reproducible
 *            offline: the contract is "within 1 ulp").
 *   sincos : psl_glibc_sin / psl_glibc_cos (psl_sincos_glibc.h) on [-pi/2, pi/2] (MergeTwoLines' `thr`,
 *            add_src/uselongline.cpp:320-329), [0, 9.5] (region2rect's theta) and [-1000, 1000]: bit-identical to libm.
 * usage: f64math_check tanf [lo_bits hi_bits] | f64 [nsamples] | sincos [nsamples]
 * build: gcc -O2 -ffp-contract=off -mfma -o f64math_check f64math_check.c -lm -lpthread */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PSL_F64_QUAL static inline
#include "../psl-slam_amd/csrc/psl_f64math.h"
#define PSL_SC64_QUAL static inline
#include "../psl-slam_amd/csrc/psl_sincos64.h"
#include "../psl-slam_amd/csrc/psl_sincos_glibc.h"
static const double SCTAB[444] = {
#include "../psl-slam_amd/csrc/psl_sincostab.inc"
};

#define NT 8
static uint32_t g_lo, g_hi;
static unsigned long long g_bad[NT];
static uint32_t g_first[NT];

static void* tan_worker(void* arg) {
    const int t = (int)(intptr_t)arg;
    unsigned long long bad = 0;
    uint32_t first = 0;
    for (uint64_t b = (uint64_t)g_lo + t; b <= g_hi; b += NT) {
        float x; uint32_t u = (uint32_t)b; memcpy(&x, &u, 4);
        float a = tanf(x), m = psl_tanf(x);
        uint32_t ua, um; memcpy(&ua, &a, 4); memcpy(&um, &m, 4);
        if (ua != um) { if (!bad) first = u; ++bad; }
    }
    g_bad[t] = bad; g_first[t] = first;
    return NULL;
}

static int64_t ulps(double a, double b) {
    int64_t ia, ib; memcpy(&ia, &a, 8); memcpy(&ib, &b, 8);
    if (ia < 0) ia = INT64_MIN - ia;
    if (ib < 0) ib = INT64_MIN - ib;
    return ia > ib ? ia - ib : ib - ia;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double urand(void) {  /* xorshift64*, uniform in [0, 1) */
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (double)((rng_state * 0x2545F4914F6CDD1Dull) >> 11) * (1.0 / 9007199254740992.0);
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    if (!strcmp(argv[1], "tanf")) {
        float lo = 0.0f, hi = 8.0f;
        memcpy(&g_lo, &lo, 4); memcpy(&g_hi, &hi, 4);
        if (argc > 3) { g_lo = (uint32_t)strtoul(argv[2], 0, 0); g_hi = (uint32_t)strtoul(argv[3], 0, 0); }
        pthread_t th[NT];
        for (int t = 0; t < NT; ++t) pthread_create(&th[t], 0, tan_worker, (void*)(intptr_t)t);
        unsigned long long bad = 0; uint32_t first = 0;
        for (int t = 0; t < NT; ++t) { pthread_join(th[t], 0); if (g_bad[t] && (!bad || g_first[t] < first)) first = g_first[t]; bad += g_bad[t]; }
        printf("tanf floats %llu mismatches %llu first 0x%08x\n", (unsigned long long)g_hi - g_lo + 1, bad, first);
        return bad ? 1 : 0;
    }
    if (!strcmp(argv[1], "ratio")) {   /* psl_ratio_inv(a, b, 1 / b) == a / b for every pair of the table's range */
        static double inv[PSL_RATIO_BMAX];
        for (int b = 1; b < PSL_RATIO_BMAX; ++b) inv[b] = 1.0 / (double)b;
        long bad = 0;
        for (int b = 1; b < PSL_RATIO_BMAX; ++b)
            for (int a = 0; a < PSL_RATIO_AMAX; ++a) {
                const double q = psl_ratio_inv((double)a, (double)b, inv[b]), ref = (double)a / (double)b;
                if (memcmp(&q, &ref, 8)) { if (bad < 5) printf("a %d b %d: %a vs %a\n", a, b, q, ref); ++bad; }
            }
        printf("ratio pairs %ld mismatches %ld\n", (long)(PSL_RATIO_BMAX - 1) * PSL_RATIO_AMAX, bad);
        return bad ? 1 : 0;
    }
    if (!strcmp(argv[1], "sincos")) {   /* psl_glibc_sin / psl_glibc_cos must be bit-identical to libm */
        const long n = argc > 2 ? atol(argv[2]) : 20000000;
        const double ranges[3][2] = {{-1.5707963267948966, 1.5707963267948966}, {0.0, 9.5}, {-1000.0, 1000.0}};
        long bad = 0;
        for (int r = 0; r < 3; ++r)
            for (long i = 0; i < n; ++i) {
                double x = ranges[r][0] + (ranges[r][1] - ranges[r][0]) * urand();
                if (i % 7 == 0) x *= 1e-4;   /* small arguments */
                const double a = sin(x), b = psl_glibc_sin(x, SCTAB), c = cos(x), d = psl_glibc_cos(x, SCTAB);
                if (memcmp(&a, &b, 8) || memcmp(&c, &d, 8)) { if (bad < 5) printf("x %a: sin %a vs %a, cos %a vs %a\n", x, a, b, c, d); ++bad; }
            }
        printf("sincos samples %ld mismatches %ld\n", 3 * n, bad);
        return bad ? 1 : 0;
    }
    const long n = argc > 2 ? atol(argv[2]) : 4000000;
    int64_t mx[3] = {0, 0, 0}; long diff[3] = {0, 0, 0};
    for (long i = 0; i < n; ++i) {
        /* log / log10: integer arguments of log_gamma (1 .. 2e5), x + 5.5, binomial tails (1e-300 .. 1), p and 1 - p */
        double x;
        switch (i & 3) {
            case 0: x = 1.0 + floor(urand() * 200000.0); break;
            case 1: x = 1.0 + urand() * 200000.0; break;
            case 2: x = pow(10.0, -300.0 * urand()); break;
            default: x = urand(); if (x == 0) x = 0.5; break;
        }
        int64_t d = ulps(log(x), psl_log(x)); if (d) ++diff[0]; if (d > mx[0]) mx[0] = d;
        d = ulps(log10(x), psl_log10(x)); if (d) ++diff[2]; if (d > mx[2]) mx[2] = d;
        /* exp: log1term of nfa() lies in [-745, 40] */
        const double e = -745.0 + 785.0 * urand();
        d = ulps(exp(e), psl_exp(e)); if (d) ++diff[1]; if (d > mx[1]) mx[1] = d;
    }
    printf("f64 samples %ld log max_ulp %lld differ %ld exp max_ulp %lld differ %ld log10 max_ulp %lld differ %ld\n", n,
           (long long)mx[0], diff[0], (long long)mx[1], diff[1], (long long)mx[2], diff[2]);
    return (mx[0] <= 1 && mx[1] <= 1 && mx[2] <= 2) ? 0 : 1;
}
