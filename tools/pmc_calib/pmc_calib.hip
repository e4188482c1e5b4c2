// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the access widths the pslfe kernels use
// (MI355X_MICROARCH.md, HBM section: only 16 B/lane streaming is calibrated; others must be calibrated on a
// known byte count).  Each kernel streams N bytes once with one access width; compare the counter with N.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/pmc_calib/pmc_calib tools/pmc_calib/pmc_calib.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void calib_read_b8(const uint8_t* p, size_t n, uint32_t* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    uint32_t s = 0;
    for (; i < n; i += stride) s += p[i];
    if (s == 0xffffffffu) *sink = s;
}
__global__ void calib_read_b32(const uint32_t* p, size_t n, uint32_t* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    uint32_t s = 0;
    for (; i < n; i += stride) s += p[i];
    if (s == 0xffffffffu) *sink = s;
}
__global__ void calib_read_b128(const uint4* p, size_t n, uint32_t* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    uint32_t s = 0;
    for (; i < n; i += stride) { const uint4 v = p[i]; s += v.x ^ v.y ^ v.z ^ v.w; }
    if (s == 0xffffffffu) *sink = s;
}
__global__ void calib_write_b32(uint32_t* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = (uint32_t)i;
}
__global__ void calib_write_b128(uint4* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}

int main() {
    const size_t N = (size_t)1 << 30;  // 1 GiB, well past the 256 MiB Infinity Cache
    void *a = nullptr, *b = nullptr;
    uint32_t* sink = nullptr;
    if (hipMalloc(&a, N) != hipSuccess || hipMalloc(&b, N) != hipSuccess || hipMalloc((void**)&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(a, 1, N);
    hipMemset(b, 2, N);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        calib_read_b8<<<8192, 256>>>((const uint8_t*)a, N / 4, sink);  // 256 MiB is enough at 1 B/lane
        calib_read_b32<<<8192, 256>>>((const uint32_t*)b, N / 4, sink);
        calib_read_b128<<<8192, 256>>>((const uint4*)a, N / 16, sink);
        calib_write_b32<<<8192, 256>>>((uint32_t*)b, N / 4);
        calib_write_b128<<<8192, 256>>>((uint4*)a, N / 16);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    printf("bytes per launch: read_b8 %zu read_b32 %zu read_b128 %zu write_b32 %zu write_b128 %zu\n", N / 4, N, N, N, N);
    return 0;
}
