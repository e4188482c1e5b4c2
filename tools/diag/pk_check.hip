// Checks on the device what lsdg_pops2 (psl-slam_amd/csrc/line_kernels.h) assumes of the packed f32 instructions' op_sel / neg_hi operands:
//   (S.x U.x, S.x U.y) -> (S.y U.y + S.x U.x, -S.y U.x + S.x U.y) -> (t_hi dot, t_lo dot), and the packed add of an SGPR pair.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/diag/pk_check.hip -o /tmp/pk_check && /tmp/pk_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void k(const float2* u, const float2* s, float2 th, float4* out, float2* added) {
    const float2 t = u[threadIdx.x], sv = s[threadIdx.x];
    unsigned long long U = ((unsigned long long)__float_as_uint(t.y) << 32) | __float_as_uint(t.x);
    unsigned long long S = ((unsigned long long)__float_as_uint(sv.y) << 32) | __float_as_uint(sv.x);
    unsigned long long TH = ((unsigned long long)__float_as_uint(th.y) << 32) | __float_as_uint(th.x);
    unsigned long long D, PT, SAB;
    asm volatile(
        "v_pk_mul_f32 %[PT], %[S], %[U] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %[D], %[S], %[U], %[PT] op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]\n\t"
        "v_pk_mul_f32 %[PT], %[TH], %[D] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
        "v_readlane_b32 s68, v54, 5\n\t"
        "v_readlane_b32 s69, v55, 5\n\t"
        "s_nop 4\n\t"
        "v_pk_add_f32 %[S], %[S], s[68:69]\n\t"
        : [S] "+v"(S), [D] "=&{v[56:57]}"(D), [PT] "=&{v[58:59]}"(PT), [SAB] "=&{s[68:69]}"(SAB)
        : [U] "{v[54:55]}"(U), [TH] "v"(TH));
    out[threadIdx.x] = make_float4(__uint_as_float((unsigned)D), __uint_as_float((unsigned)(D >> 32)), __uint_as_float((unsigned)PT), __uint_as_float((unsigned)(PT >> 32)));
    added[threadIdx.x] = make_float2(__uint_as_float((unsigned)S), __uint_as_float((unsigned)(S >> 32)));
}

int main() {
    std::vector<float2> u(64), s(64);
    srand(5);
    for (int i = 0; i < 64; ++i) {
        const float a = 6.2831853f * rand() / RAND_MAX;
        u[i] = make_float2(cosf(a), sinf(a));
        s[i] = make_float2(40.f * rand() / RAND_MAX - 20.f, 40.f * rand() / RAND_MAX - 20.f);
    }
    const float2 th = make_float2(0.41f, 0.418f);
    float2 *du, *ds, *dadd; float4* dout;
    hipMalloc(&du, 512); hipMalloc(&ds, 512); hipMalloc(&dadd, 512); hipMalloc(&dout, 1024);
    hipMemcpy(du, u.data(), 512, hipMemcpyHostToDevice); hipMemcpy(ds, s.data(), 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(du, ds, th, dout, dadd);
    std::vector<float4> out(64); std::vector<float2> add(64);
    if (hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL: no device result\n"); return 2; }
    hipMemcpy(add.data(), dadd, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const float dot = fmaf(s[i].y, u[i].y, s[i].x * u[i].x), cr = fmaf(-s[i].y, u[i].x, s[i].x * u[i].y);
        const float e[4] = {dot, cr, th.x * dot, th.y * dot};
        const float g[4] = {out[i].x, out[i].y, out[i].z, out[i].w};
        for (int q = 0; q < 4; ++q) if (memcmp(&e[q], &g[q], 4)) { if (bad < 8) printf("lane %d value %d: got %g expected %g\n", i, q, g[q], e[q]); ++bad; }
        const float ax = s[i].x + u[5].x, ay = s[i].y + u[5].y;
        if (add[i].x != ax || add[i].y != ay) { if (bad < 8) printf("lane %d add: got %g %g expected %g %g\n", i, add[i].x, add[i].y, ax, ay); ++bad; }
    }
    printf(bad ? "FAIL: %d mismatches\n" : "pk_check ok\n", bad);
    return bad ? 1 : 0;
}
