// One-off pinning tool (test infrastructure): compares the oracle's restated sinf/cosf
// (oracle/psl_math_oracle.h) with this host's libm sinf/cosf for EVERY float in [-2*pi, 2*pi].
// Usage: ./sincosf_check [nthreads]
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "psl_math_oracle.h"

typedef struct { uint32_t lo, hi; uint64_t bad_s, bad_c; uint32_t first_bad; } job_t;

static void* run(void* p) {
    job_t* j = (job_t*)p;
    for (uint32_t u = j->lo; u < j->hi; ++u) {
        float x; memcpy(&x, &u, 4);
        for (int sg = 0; sg < 2; ++sg) {  // x and -x: LBD feeds line directions in [-pi, pi]
            float xx = sg ? -x : x;
            float s0 = sinf(xx), c0 = cosf(xx);
            float s1 = pso_sinf(xx), c1 = pso_cosf(xx);
            if (memcmp(&s0, &s1, 4)) { if (!j->bad_s && !j->bad_c) j->first_bad = u; j->bad_s++; }
            if (memcmp(&c0, &c1, 4)) { if (!j->bad_s && !j->bad_c) j->first_bad = u; j->bad_c++; }
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    int nt = argc > 1 ? atoi(argv[1]) : 8;
    float top = 6.2831855f;  // (float)(2*pi), rounds up
    uint32_t utop; memcpy(&utop, &top, 4);
    utop += 1;
    pthread_t th[64]; job_t jobs[64];
    uint64_t per = ((uint64_t)utop + nt - 1) / nt;
    for (int i = 0; i < nt; ++i) {
        jobs[i].lo = (uint32_t)(per * i); uint64_t hi = per * (i + 1); if (hi > utop) hi = utop;
        jobs[i].hi = (uint32_t)hi; jobs[i].bad_s = jobs[i].bad_c = 0; jobs[i].first_bad = 0;
        pthread_create(&th[i], 0, run, &jobs[i]);
    }
    uint64_t bs = 0, bc = 0; uint32_t fb = 0;
    for (int i = 0; i < nt; ++i) { pthread_join(th[i], 0); bs += jobs[i].bad_s; bc += jobs[i].bad_c; if (!fb) fb = jobs[i].first_bad; }
    printf("checked %u floats in [0,2pi]: sinf mismatches %llu, cosf mismatches %llu, first_bad_bits 0x%08x\n",
           utop, (unsigned long long)bs, (unsigned long long)bc, fb);
    return (bs || bc) ? 1 : 0;
}
