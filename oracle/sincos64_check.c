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

typedef struct { uint32_t lo, hi; uint64_t bad_s, bad_c; uint32_t first_bad; } job_t;

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
        jobs[i].lo = (uint32_t)a; jobs[i].hi = (uint32_t)b; jobs[i].bad_s = jobs[i].bad_c = 0; jobs[i].first_bad = 0;
        pthread_create(&th[i], 0, run, &jobs[i]);
    }
    uint64_t bs = 0, bc = 0; uint32_t fb = 0;
    for (int i = 0; i < nt; ++i) { pthread_join(th[i], 0); bs += jobs[i].bad_s; bc += jobs[i].bad_c; if (!fb) fb = jobs[i].first_bad; }
    printf("checked %u floats of degrees in [0, 360]: sin mismatches %llu, cos mismatches %llu, first_bad_bits 0x%08x\n",
           utop - ulo, (unsigned long long)bs, (unsigned long long)bc, fb);
    return (bs || bc) ? 1 : 0;
}
