// Bench harness only (not part of libpslfe): device-side stand-in for Tracking's constant-velocity projection.
// Query i of pair f = keypoint i of frame f-1 (cyclic inside the batch) predicted at the same pixel, window
// th = 15 * scaleFactor[octave], level band octave-1..octave+1, descriptor = that keypoint's descriptor.
// One launch instead of ~20 small framework kernels per step, so the timed step is the hot path itself.
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o tools/bench_kernels/libbench_kernels.so tools/bench_kernels/bench_kernels.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

struct KeyPoint { float x, y, size, angle, response; int32_t octave, class_id; };
struct ProjQuery { float u, v, radius, ur; int32_t min_level, max_level; float angle; int32_t blocks; };

__global__ __launch_bounds__(256) void k_queries_from_prev(const KeyPoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                            const int32_t* __restrict__ counts, int nframes, int cap, int nlevels,
                                                            const float* __restrict__ scale, float th, ProjQuery* __restrict__ q,
                                                            uint8_t* __restrict__ qdesc, int32_t* __restrict__ nq) {
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int p = (f + nframes - 1) % nframes;
    if (i == 0) nq[f] = counts[p];
    if (i >= cap) return;
    const KeyPoint k = kps[(size_t)p * cap + i];
    int o = k.octave;
    o = o < 0 ? 0 : (o >= nlevels ? nlevels - 1 : o);
    ProjQuery r;
    r.u = k.x; r.v = k.y; r.radius = th * scale[o]; r.ur = 0.f;
    r.min_level = k.octave - 1; r.max_level = k.octave + 1; r.angle = k.angle; r.blocks = 1;
    q[(size_t)f * cap + i] = r;
    const uint4* s = reinterpret_cast<const uint4*>(desc + ((size_t)p * cap + i) * 32);
    uint4* d = reinterpret_cast<uint4*>(qdesc + ((size_t)f * cap + i) * 32);
    d[0] = s[0]; d[1] = s[1];
}

extern "C" int bench_queries_from_prev(void* stream, const void* kps, const void* desc, const void* counts, int nframes, int cap, int nlevels,
                                       const void* scale, float th, void* q, void* qdesc, void* nq) {
    k_queries_from_prev<<<dim3((cap + 255) / 256, nframes), 256, 0, (hipStream_t)stream>>>(
        (const KeyPoint*)kps, (const uint8_t*)desc, (const int32_t*)counts, nframes, cap, nlevels, (const float*)scale, th, (ProjQuery*)q,
        (uint8_t*)qdesc, (int32_t*)nq);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
