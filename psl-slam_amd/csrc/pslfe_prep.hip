// libpslfe: input conversions in front of the extractors. Product code.
// Reference behaviour reproduced (Tracking::GrabImageRGBD src/Tracking.cc:214-240):
//   cvtColor(mImGray, mImGray, CV_RGB2GRAY / CV_BGR2GRAY)  :219-232  OpenCV 8-bit fixed point, Appendix A:
//        gray = (R*4899 + G*9617 + B*1868 + (1<<13)) >> 14
//   imDepth.convertTo(imDepth, CV_32F, mDepthMapFactor)      :234-235  (float)d * (float)factor
// Both are pure streaming kernels (HBM-bound): 3 B in / 1 B out and 2 B in / 4 B out per pixel.
#include "pslfe_internal.h"
#include "psl_device_math.h"

// One thread converts 4 pixels: three dword loads (12 B) -> one dword store.  Rows are handled flat when the
// image is packed (stride == 3*w); otherwise per row.
__global__ __launch_bounds__(256) void k_rgb_to_gray(const uint8_t* __restrict__ rgb, size_t npix4, int swap_rb, uint8_t* __restrict__ gray) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix4) return;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(rgb) + i * 3;
    const uint32_t a = p[0], b = p[1], c = p[2];
    // bytes: a = c0.0 c0.1 c0.2 c1.0 | b = c1.1 c1.2 c2.0 c2.1 | c = c2.2 c3.0 c3.1 c3.2
    const int cr = swap_rb ? 1868 : 4899, cb = swap_rb ? 4899 : 1868;
    auto g = [&](uint32_t x, uint32_t y, uint32_t z) { return (uint32_t)((int)x * cr + (int)y * 9617 + (int)z * cb + (1 << 13)) >> 14; };
    const uint32_t g0 = g(a & 255, (a >> 8) & 255, (a >> 16) & 255);
    const uint32_t g1 = g(a >> 24, b & 255, (b >> 8) & 255);
    const uint32_t g2 = g((b >> 16) & 255, b >> 24, c & 255);
    const uint32_t g3 = g((c >> 8) & 255, (c >> 16) & 255, c >> 24);
    reinterpret_cast<uint32_t*>(gray)[i] = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
}

__global__ __launch_bounds__(256) void k_rgb_to_gray_rows(const uint8_t* __restrict__ rgb, int w, int h, int stride, size_t frame_stride,
                                                           int swap_rb, uint8_t* __restrict__ gray) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const uint8_t* p = rgb + (size_t)blockIdx.z * frame_stride + (size_t)y * stride + 3 * x;
    const int r = swap_rb ? p[2] : p[0], g = p[1], b = swap_rb ? p[0] : p[2];
    gray[((size_t)blockIdx.z * h + y) * w + x] = (uint8_t)((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14);
}

__global__ __launch_bounds__(256) void k_depth_to_float(const uint16_t* __restrict__ d, size_t n, float factor, float* __restrict__ out) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 < n) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(d + i);
        float2 o;
        o.x = PSL_FMUL((float)(v & 0xffff), factor);
        o.y = PSL_FMUL((float)(v >> 16), factor);
        *reinterpret_cast<float2*>(out + i) = o;
    } else if (i < n) {
        out[i] = PSL_FMUL((float)d[i], factor);
    }
}

extern "C" {

int pslfe_rgb_to_gray_device(pslfe_ctx* ctx, const uint8_t* d_rgb, int nframes, int w, int h, int stride, size_t frame_stride,
                             int is_rgb, uint8_t* d_gray) {
    PSL_REQUIRE(ctx && d_rgb && d_gray, PSLFE_E_INVALID, "pslfe_rgb_to_gray_device: NULL argument");
    PSL_REQUIRE(nframes >= 1 && w > 0 && h > 0 && stride >= 3 * w && frame_stride >= (size_t)stride * (h - 1) + 3 * (size_t)w, PSLFE_E_INVALID,
                "pslfe_rgb_to_gray_device: %d frames %dx%d stride %d", nframes, w, h, stride);
    PSL_HIP(hipSetDevice(ctx->device));
    const bool packed = stride == 3 * w && frame_stride == (size_t)stride * h && ((size_t)w * h) % 4 == 0 &&
                        ((uintptr_t)d_rgb & 3) == 0 && ((uintptr_t)d_gray & 3) == 0;
    {
        PSL_STAGE_BEGIN(ctx, "prep.gray");
        if (packed) {
            const size_t n4 = (size_t)nframes * w * h / 4;
            k_rgb_to_gray<<<(unsigned)((n4 + 255) / 256), 256, 0, ctx->stream>>>(d_rgb, n4, is_rgb ? 0 : 1, d_gray);
        } else {
            k_rgb_to_gray_rows<<<dim3((w + 255) / 256, h, nframes), 256, 0, ctx->stream>>>(d_rgb, w, h, stride, frame_stride, is_rgb ? 0 : 1, d_gray);
        }
        PSL_STAGE_END(ctx, "prep.gray");
    }
    PSL_HIP(hipGetLastError());
    return PSLFE_OK;
}

int pslfe_depth_to_float_device(pslfe_ctx* ctx, const uint16_t* d_depth, size_t n, float factor, float* d_out) {
    PSL_REQUIRE(ctx && d_depth && d_out, PSLFE_E_INVALID, "pslfe_depth_to_float_device: NULL argument");
    PSL_REQUIRE(((uintptr_t)d_depth & 3) == 0 && ((uintptr_t)d_out & 7) == 0, PSLFE_E_INVALID, "pslfe_depth_to_float_device: unaligned buffers");
    if (n == 0) return PSLFE_OK;
    PSL_HIP(hipSetDevice(ctx->device));
    {
        PSL_STAGE_BEGIN(ctx, "prep.depth");
        k_depth_to_float<<<(unsigned)(((n + 1) / 2 + 255) / 256), 256, 0, ctx->stream>>>(d_depth, n, factor, d_out);
        PSL_STAGE_END(ctx, "prep.depth");
    }
    PSL_HIP(hipGetLastError());
    return PSLFE_OK;
}

// Host-pointer conveniences (one frame): H2D, kernel, D2H.
int pslfe_rgb_to_gray(pslfe_ctx* ctx, const uint8_t* rgb, int w, int h, int stride, int is_rgb, uint8_t* gray) {
    PSL_REQUIRE(ctx && rgb && gray, PSLFE_E_INVALID, "pslfe_rgb_to_gray: NULL argument");
    PSL_REQUIRE(w > 0 && h > 0 && stride >= 3 * w, PSLFE_E_INVALID, "pslfe_rgb_to_gray: %dx%d stride %d", w, h, stride);
    PSL_HIP(hipSetDevice(ctx->device));
    uint8_t *d_in = nullptr, *d_out = nullptr;
    const size_t bytes = (size_t)stride * h;
    PSL_HIP(hipMalloc((void**)&d_in, bytes));
    hipError_t e = hipMalloc((void**)&d_out, (size_t)w * h);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, rgb, (size_t)stride * (h - 1) + 3 * (size_t)w, hipMemcpyHostToDevice, ctx->stream);
    int rc = PSLFE_OK;
    if (e != hipSuccess) { pslfe_set_error("pslfe_rgb_to_gray: %s", hipGetErrorString(e)); rc = PSLFE_E_HIP; }
    if (!rc) rc = pslfe_rgb_to_gray_device(ctx, d_in, 1, w, h, stride, bytes, is_rgb, d_out);
    if (!rc) {
        e = hipMemcpyAsync(gray, d_out, (size_t)w * h, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { pslfe_set_error("pslfe_rgb_to_gray: %s", hipGetErrorString(e)); rc = PSLFE_E_HIP; }
    }
    hipStreamSynchronize(ctx->stream);
    hipFree(d_in); hipFree(d_out);
    return rc;
}

int pslfe_depth_to_float(pslfe_ctx* ctx, const uint16_t* depth, size_t n, float factor, float* out) {
    PSL_REQUIRE(ctx && (n == 0 || (depth && out)), PSLFE_E_INVALID, "pslfe_depth_to_float: NULL argument");
    if (n == 0) return PSLFE_OK;
    PSL_HIP(hipSetDevice(ctx->device));
    uint16_t* d_in = nullptr; float* d_out = nullptr;
    PSL_HIP(hipMalloc((void**)&d_in, n * 2 + 4));
    hipError_t e = hipMalloc((void**)&d_out, n * 4 + 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, depth, n * 2, hipMemcpyHostToDevice, ctx->stream);
    int rc = PSLFE_OK;
    if (e != hipSuccess) { pslfe_set_error("pslfe_depth_to_float: %s", hipGetErrorString(e)); rc = PSLFE_E_HIP; }
    if (!rc) rc = pslfe_depth_to_float_device(ctx, d_in, n, factor, d_out);
    if (!rc) {
        e = hipMemcpyAsync(out, d_out, n * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { pslfe_set_error("pslfe_depth_to_float: %s", hipGetErrorString(e)); rc = PSLFE_E_HIP; }
    }
    hipStreamSynchronize(ctx->stream);
    hipFree(d_in); hipFree(d_out);
    return rc;
}

}  // extern "C"
