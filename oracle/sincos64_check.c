// One-off pinning tool (test infrastructure): compares the product's restricted-range (float)cos / (float)sin of a double
// (psl-slam_amd/csrc/psl_sincos64.h, used by k_lsd_grad for the seed terms of LSD region growing) with this host's libm
// for EVERY float number of degrees in [0, 360], the argument being (double)deg * DEG_TO_RADS as in the kernels.
// Usage: ./sincos64_check [nthreads]
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "../psl-slam_amd/csrc/psl_sincos64.h"

#define DEG_TO_RADS (3.1415926535897932384626433832795 / 180)  /* PSL_DEG2RAD of line_kernels.h */

typedef struct { uint32_t lo, hi; uint64_t bad_s, bad_c; uint32_t first_bad; double max_ulp; } job_t;

static double ulps(double got, double ref) {  // error in units of the last place of max(|ref|, 2^-10): near a zero of the function
    const double m = fabs(ref) > 0x1p-10 ? fabs(ref) : 0x1p-10;  // the absolute error is what the rectangle step sees
    return fabs(got - ref) / ldexp(1.0, ilogb(m) - 52);
}

static void* run(void* p) {
    job_t* j = (job_t*)p;
    for (uint32_t u = j->lo; u < j->hi; ++u) {
        float deg; memcpy(&deg, &u, 4);
        const double ad = (double)deg * DEG_TO_RADS;
        const float c0 = (float)cos(ad), s0 = (float)sin(ad);
        float c1, s1;
        psl_cos_sin_2pi_f32(ad, &c1, &s1);
        if (memcmp(&s0, &s1, 4)) { if (!j->bad_s && !j->bad_c) j->first_bad = u; j->bad_s++; }
        if (memcmp(&c0, &c1, 4)) { if (!j->bad_s && !j->bad_c) j->first_bad = u; j->bad_c++; }
        if ((u & 7) == 0) {  // the f64 variant used by the rectangle step (theta or theta + pi): error against libm, every 8th angle
            for (int k = 0; k < 2; ++k) {
                const double t = k ? ad + 3.14159265358979323846 : ad;
                double c2, s2;
                psl_cos_sin_f64(t, &c2, &s2);
                const double e1 = ulps(c2, cos(t)), e2 = ulps(s2, sin(t));
                if (e1 > j->max_ulp) j->max_ulp = e1;
                if (e2 > j->max_ulp) j->max_ulp = e2;
            }
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    int nt = argc > 1 ? atoi(argv[1]) : 8;
    float lo = 0.0f, top = 360.0f;
    uint32_t ulo, utop; memcpy(&ulo, &lo, 4); memcpy(&utop, &top, 4);
    utop += 1;
    pthread_t th[64]; job_t jobs[64];
    uint64_t per = ((uint64_t)(utop - ulo) + nt - 1) / nt;
    for (int i = 0; i < nt; ++i) {
        uint64_t a = ulo + per * i, b = ulo + per * (i + 1); if (b > utop) b = utop;
        jobs[i].lo = (uint32_t)a; jobs[i].hi = (uint32_t)b; jobs[i].bad_s = jobs[i].bad_c = 0; jobs[i].first_bad = 0; jobs[i].max_ulp = 0;
        pthread_create(&th[i], 0, run, &jobs[i]);
    }
    uint64_t bs = 0, bc = 0; uint32_t fb = 0;
    for (int i = 0; i < nt; ++i) { pthread_join(th[i], 0); bs += jobs[i].bad_s; bc += jobs[i].bad_c; if (!fb) fb = jobs[i].first_bad; }
    double mu = 0;
    for (int i = 0; i < nt; ++i) if (jobs[i].max_ulp > mu) mu = jobs[i].max_ulp;
    printf("checked %u floats of degrees in [0, 360]: sin mismatches %llu, cos mismatches %llu, first_bad_bits 0x%08x\n",
           utop - ulo, (unsigned long long)bs, (unsigned long long)bc, fb);
    printf("f64 variant on [0, 3 pi): max error %.3f ulp against libm\n", mu);
    return (bs || bc || mu > 3.0) ? 1 : 0;
}
