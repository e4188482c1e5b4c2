// One-off pinning tool (test infrastructure): compares the product's restated atanf / atan2f (psl-slam_amd/csrc/psl_atanf.h,
// used by the line merging) with this host's libm: atanf for EVERY float, atan2f on 2e8 pairs (random bit patterns and
// pixel-difference-like values).  Usage: ./atanf_check 0|1
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <pthread.h>
#include <stdlib.h>
#define PSL_HD static inline
#define PSL_FMUL(a, b) ((a) * (b))
#define PSL_FADD(a, b) ((a) + (b))
#define PSL_FSUB(a, b) ((a) - (b))
#define PSL_FDIV(a, b) ((a) / (b))
#include "../psl-slam_amd/csrc/psl_atanf.h"

typedef struct { uint32_t lo, hi; uint64_t bad; uint32_t first; int mode; uint64_t seed; } job_t;
static void* run(void* p) {
    job_t* j = (job_t*)p;
    if (j->mode == 0) {
        for (uint64_t u = j->lo; u < j->hi; ++u) {
            uint32_t v = (uint32_t)u; float x; memcpy(&x, &v, 4);
            if (x != x) continue;
            float a = atanf(x), b = psl_atanf(x);
            if (memcmp(&a, &b, 4)) { if (!j->bad) j->first = v; j->bad++; }
        }
    } else {
        uint64_t s = j->seed;
        for (uint64_t n = 0; n < 25000000ull; ++n) {
            s = s * 6364136223846793005ull + 1442695040888963407ull; uint32_t a = (uint32_t)(s >> 32);
            s = s * 6364136223846793005ull + 1442695040888963407ull; uint32_t b = (uint32_t)(s >> 32);
            float y, x;
            if (n & 1) { y = ((int32_t)a >> 8) * (1.0f / 64.0f); x = ((int32_t)b >> 8) * (1.0f / 64.0f); }
            else { memcpy(&y, &a, 4); memcpy(&x, &b, 4); }
            if (x != x || y != y) continue;
            float r0 = atan2f(y, x), r1 = psl_atan2f(y, x);
            if (memcmp(&r0, &r1, 4)) { if (!j->bad) j->first = a; j->bad++; }
        }
    }
    return 0;
}
int main(int argc, char** argv) {
    int mode = argc > 1 ? atoi(argv[1]) : 0;
    pthread_t th[8]; job_t jb[8];
    for (int i = 0; i < 8; i++) {
        jb[i].lo = (uint32_t)(0x20000000ull * i); jb[i].hi = (i == 7) ? 0xffffffffu : (uint32_t)(0x20000000ull * (i + 1));
        jb[i].bad = 0; jb[i].first = 0; jb[i].mode = mode; jb[i].seed = 12345 + 977 * i;
        pthread_create(&th[i], 0, run, &jb[i]);
    }
    uint64_t bad = 0; uint32_t f = 0;
    for (int i = 0; i < 8; i++) { pthread_join(th[i], 0); bad += jb[i].bad; if (!f) f = jb[i].first; }
    printf("%s mismatches %llu first 0x%08x\n", mode ? "atan2f (2e8 pairs)" : "atanf (all floats)", (unsigned long long)bad, f);
    return bad ? 1 : 0;
}
