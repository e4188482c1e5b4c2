// libpslfe: line extractor object (== ORB_SLAM2::LINEextractor) over the HIP kernels. Product code.
// Reference: add_src/LineExtractor.cpp:6-25, 325-366; add_inc/LineExtractor.h:160-255.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "line_kernels.h"
#include "pslfe_internal.h"

struct pslfe_line {
    pslfe_ctx* ctx = nullptr;
    int numOctaves = 1, nfeatures = 200, max_batch = 1;
    float scale = 1.2f;
    double min_line_length = 0;
    std::vector<float> scaleF, invScaleF, sigma2, invSigma2;

    int gw = 0, gh = 0;
    LineParams P;
    size_t in_fstride = 0;
    int in_pitch = 0;
    int last_nframes = 0;

    uint8_t* d_in = nullptr;
    double* d_scaled = nullptr;
    float* d_angdeg = nullptr;
    double* d_modgrad = nullptr;
    uint8_t* d_used = nullptr;
    uint32_t* d_reg = nullptr;
    float* d_seg = nullptr;
    int* d_nseg = nullptr;

    void release() {
        hipFree(d_in); hipFree(d_scaled); hipFree(d_angdeg); hipFree(d_modgrad); hipFree(d_used); hipFree(d_reg);
        hipFree(d_seg); hipFree(d_nseg);
        d_in = nullptr; d_scaled = nullptr; d_angdeg = nullptr; d_modgrad = nullptr; d_used = nullptr; d_reg = nullptr;
        d_seg = nullptr; d_nseg = nullptr;
    }

    int prepare(int w, int h) {
        if (w == gw && h == gh) return PSLFE_OK;
        PSL_REQUIRE(w >= 16 && h >= 16 && w <= 8192 && h <= 8192, PSLFE_E_INVALID, "line: image %dx%d out of range", w, h);
        LineParams Q;
        memset(&Q, 0, sizeof(Q));
        Q.w = w; Q.h = h;
        Q.W = (int)nearbyint(w * 0.8);
        Q.H = (int)nearbyint(h * 0.8);
        Q.maxseg = 8192;
        Q.maxkl = 2048;
        Q.nfeatures = nfeatures;
        {   // getGaussianKernel(7, 0.75, CV_64F): sigma = SIGMA_SCALE / SCALE, ksize = 1 + 2*ceil(sigma*sqrt(2*3*ln 10))
            const double sigma = 0.6 / 0.8;
            const int ksize = 1 + 2 * (int)ceil(sigma * sqrt(2 * 3.0 * log(10.0)));
            PSL_REQUIRE(ksize == 7, PSLFE_E_INVALID, "line: unexpected LSD kernel size %d", ksize);
            const double scale2X = -0.5 / (sigma * sigma);
            double sum = 0;
            for (int i = 0; i < 7; ++i) { const double x = i - 3.0; Q.gk[i] = exp(scale2X * x * x); sum += Q.gk[i]; }
            sum = 1. / sum;
            for (int i = 0; i < 7; ++i) Q.gk[i] *= sum;
        }
        Q.prec = PSL_PI * 22.5 / 180;
        Q.p = 22.5 / 180;
        Q.rho = 2.0 / sin(Q.prec);
        const double LOG_NT = 5 * (log10((double)Q.W) + log10((double)Q.H)) / 2 + log10(11.0);
        Q.min_reg_size = (int)(size_t)(-LOG_NT / log10(Q.p));
        {   // 8-bit GaussianBlur 5x5 sigma 1 -> integer kernel (OpenCV 3.2)
            float cf[5];
            double sum = 0;
            for (int i = 0; i < 5; ++i) { const double x = i - 2.0; cf[i] = (float)exp(-0.5 * x * x); sum += cf[i]; }
            sum = 1. / sum;
            for (int i = 0; i < 5; ++i) { cf[i] = (float)(cf[i] * sum); Q.lbdK[i] = (int)nearbyint((double)cf[i] * 256.0); }
        }
        {   // BinaryDescriptor ctor (binary_descriptor_custom.cpp:219-261), integer divisions as written there
            const int wb = 7, nb = 9;
            double u = (wb * 3 - 1) / 2, sigma = (wb * 2 + 1) / 2, inv = -1 / (2 * sigma * sigma);
            for (int i = 0; i < wb * 3; ++i) { const double d = i - u; Q.gaussL[i] = (float)exp(d * d * inv); }
            u = (nb * wb - 1) / 2; sigma = u; inv = -1 / (2 * sigma * sigma);
            for (int i = 0; i < nb * wb; ++i) { const double d = i - u; Q.gaussG[i] = (float)exp(d * d * inv); }
        }
        PSL_HIP(hipSetDevice(ctx->device));
        PSL_HIP(hipStreamSynchronize(ctx->stream));
        release();
        const size_t F = (size_t)max_batch, npx = (size_t)Q.W * Q.H;
        in_pitch = (int)psl_align_up(w, 16);
        in_fstride = psl_align_up((size_t)in_pitch * h, 256);
        PSL_HIP(hipMalloc((void**)&d_in, in_fstride * F));
        PSL_HIP(hipMalloc((void**)&d_scaled, npx * F * sizeof(double)));
        PSL_HIP(hipMalloc((void**)&d_angdeg, npx * F * sizeof(float)));
        PSL_HIP(hipMalloc((void**)&d_modgrad, npx * F * sizeof(double)));
        PSL_HIP(hipMalloc((void**)&d_used, npx * F));
        PSL_HIP(hipMalloc((void**)&d_reg, npx * F * sizeof(uint32_t)));
        PSL_HIP(hipMalloc((void**)&d_seg, (size_t)Q.maxseg * 4 * sizeof(float) * F));
        PSL_HIP(hipMalloc((void**)&d_nseg, F * sizeof(int)));
        P = Q;
        gw = w; gh = h;
        last_nframes = 0;
        return PSLFE_OK;
    }

    int run_lsd(const uint8_t* d_gray, int nframes, int w, int h, int stride, size_t frame_stride) {
        int rc = prepare(w, h);
        if (rc) return rc;
        PSL_HIP(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        const unsigned F = (unsigned)nframes;
        dim3 grid((P.W + 63) / 64, (P.H + 3) / 4, F);
        {
            PSL_STAGE_BEGIN(ctx, "line.lsd_scale");
            k_lsd_scale<<<grid, 256, 0, st>>>(P, d_gray, stride, frame_stride, d_scaled);
            PSL_STAGE_END(ctx, "line.lsd_scale");
        }
        {
            PSL_STAGE_BEGIN(ctx, "line.lsd_grad");
            k_lsd_grad<<<grid, 256, 0, st>>>(P, d_scaled, d_angdeg, d_modgrad);
            PSL_STAGE_END(ctx, "line.lsd_grad");
        }
        {
            PSL_STAGE_BEGIN(ctx, "line.lsd_grow");
            k_lsd_grow<<<F, 64, 0, st>>>(P, d_angdeg, d_modgrad, d_used, d_reg, d_seg, d_nseg);
            PSL_STAGE_END(ctx, "line.lsd_grow");
        }
        PSL_HIP(hipGetLastError());
        last_nframes = nframes;
        return PSLFE_OK;
    }

    int upload(const uint8_t* gray, int nframes, int w, int h, int stride, size_t frame_stride) {
        int rc = prepare(w, h);
        if (rc) return rc;
        PSL_HIP(hipSetDevice(ctx->device));
        for (int f = 0; f < nframes; ++f)
            PSL_HIP(hipMemcpy2DAsync(d_in + (size_t)f * in_fstride, in_pitch, gray + (size_t)f * frame_stride, stride, w, h,
                                     hipMemcpyHostToDevice, ctx->stream));
        return PSLFE_OK;
    }
};

extern "C" {

int pslfe_line_create(pslfe_ctx* ctx, int numOctaves, float scale, int nLSDFeature, double min_line_length, int max_batch,
                      pslfe_line** out) {
    PSL_REQUIRE(ctx && out, PSLFE_E_INVALID, "pslfe_line_create: NULL argument");
    *out = nullptr;
    PSL_REQUIRE(numOctaves >= 1 && numOctaves <= PSLFE_MAX_LEVELS && nLSDFeature >= 1 && nLSDFeature <= 2048 && max_batch >= 1 && max_batch <= 65535,
                PSLFE_E_INVALID, "pslfe_line_create: numOctaves %d nLSDFeature %d max_batch %d", numOctaves, nLSDFeature, max_batch);
    // LINEextractor::operator() calls detect(image, kls, scale, numOctaves) whose `int scale` parameter
    // truncates 1.2 to 1 and every RGB-D YAML sets LINEextractor.nLevels: 1 (add_src/LineExtractor.cpp:336-337)
    PSL_REQUIRE(numOctaves == 1, PSLFE_E_INVALID, "pslfe_line_create: only numOctaves == 1 is supported (all reference configurations)");
    pslfe_line* l = new pslfe_line();
    l->ctx = ctx; l->numOctaves = numOctaves; l->scale = scale; l->nfeatures = nLSDFeature; l->min_line_length = min_line_length;
    l->max_batch = max_batch;
    // add_src/LineExtractor.cpp:8-24
    l->scaleF.resize(numOctaves); l->sigma2.resize(numOctaves); l->invScaleF.resize(numOctaves); l->invSigma2.resize(numOctaves);
    l->scaleF[0] = 1.0f; l->sigma2[0] = 1.0f;
    for (int i = 1; i < numOctaves; ++i) { l->scaleF[i] = l->scaleF[i - 1] * scale; l->sigma2[i] = l->scaleF[i] * l->scaleF[i]; }
    for (int i = 0; i < numOctaves; ++i) { l->invScaleF[i] = 1.0f / l->scaleF[i]; l->invSigma2[i] = 1.0f / l->sigma2[i]; }
    *out = l;
    return PSLFE_OK;
}

void pslfe_line_destroy(pslfe_line* line) {
    if (!line) return;
    hipSetDevice(line->ctx->device);
    hipStreamSynchronize(line->ctx->stream);
    line->release();
    delete line;
}

int pslfe_line_levels(const pslfe_line* line) { return line ? line->numOctaves : PSLFE_E_INVALID; }
float pslfe_line_scale_factor(const pslfe_line* line) { return line ? line->scale : 0.f; }
int pslfe_line_scale_factors(const pslfe_line* line, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2) {
    PSL_REQUIRE(line, PSLFE_E_INVALID, "pslfe_line_scale_factors: line is NULL");
    for (int i = 0; i < line->numOctaves; ++i) {
        if (scale) scale[i] = line->scaleF[i];
        if (inv_scale) inv_scale[i] = line->invScaleF[i];
        if (sigma2) sigma2[i] = line->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = line->invSigma2[i];
    }
    return PSLFE_OK;
}

int pslfe_lsd_detect(pslfe_line* line, const uint8_t* gray, int w, int h, int stride, float* segments, int cap, int* n) {
    PSL_REQUIRE(line && n, PSLFE_E_INVALID, "pslfe_lsd_detect: NULL argument");
    *n = 0;
    if (!gray || w <= 0 || h <= 0) return PSLFE_OK;
    PSL_REQUIRE(stride >= w, PSLFE_E_INVALID, "pslfe_lsd_detect: stride %d < width %d", stride, w);
    int rc = line->upload(gray, 1, w, h, stride, (size_t)stride * h);
    if (rc) return rc;
    rc = line->run_lsd(line->d_in, 1, w, h, line->in_pitch, line->in_fstride);
    if (rc) return rc;
    hipStream_t st = line->ctx->stream;
    int cnt = 0;
    PSL_HIP(hipMemcpyAsync(&cnt, line->d_nseg, sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    *n = cnt;
    PSL_REQUIRE(cnt <= cap, PSLFE_E_CAPACITY, "pslfe_lsd_detect: %d segments, capacity %d", cnt, cap);
    if (cnt > 0 && segments) {
        PSL_HIP(hipMemcpyAsync(segments, line->d_seg, (size_t)cnt * 4 * sizeof(float), hipMemcpyDeviceToHost, st));
        PSL_HIP(hipStreamSynchronize(st));
    }
    return PSLFE_OK;
}

int pslfe_line_debug_gradient(pslfe_line* line, int frame, int* W, int* H, double* scaled, float* angle_deg, double* modgrad) {
    PSL_REQUIRE(line && W && H, PSLFE_E_INVALID, "pslfe_line_debug_gradient: NULL argument");
    PSL_REQUIRE(line->last_nframes > 0 && frame >= 0 && frame < line->last_nframes, PSLFE_E_STATE, "pslfe_line_debug_gradient: frame %d", frame);
    PSL_HIP(hipSetDevice(line->ctx->device));
    PSL_HIP(hipStreamSynchronize(line->ctx->stream));
    *W = line->P.W; *H = line->P.H;
    const size_t npx = (size_t)line->P.W * line->P.H;
    if (scaled) PSL_HIP(hipMemcpy(scaled, line->d_scaled + frame * npx, npx * sizeof(double), hipMemcpyDeviceToHost));
    if (angle_deg) PSL_HIP(hipMemcpy(angle_deg, line->d_angdeg + frame * npx, npx * sizeof(float), hipMemcpyDeviceToHost));
    if (modgrad) PSL_HIP(hipMemcpy(modgrad, line->d_modgrad + frame * npx, npx * sizeof(double), hipMemcpyDeviceToHost));
    return PSLFE_OK;
}

}  // extern "C"
