// libpslfe: ORB extractor object (== ORB_SLAM2::ORBextractor) over the HIP kernels. Product code.
// Reference: src/ORBextractor.cc, include/ORBextractor.h.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "orb_kernels.h"
#include "pslfe_internal.h"

namespace {

inline int cv_round(double v) { return (int)nearbyint(v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }

// Coefficient tables of cv::resize INTER_LINEAR (OpenCV 3.2, 11-bit fixed point; Appendix A.3).
void build_linear_table(int ssize, int dsize, bool clamp_coef, std::vector<int>& ofs, std::vector<short>& coef) {
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    ofs.assign(dsize, 0);
    coef.assign((size_t)dsize * 2, 0);
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor(f);
        f -= s;
        if (clamp_coef) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        ofs[d] = s;
        coef[2 * d] = (short)cv_round((1.f - f) * 2048.f);
        coef[2 * d + 1] = (short)cv_round(f * 2048.f);
    }
}

// Integer kernel of the 8-bit GaussianBlur path of OpenCV 3.2: round(getGaussianKernel(f32) * 256).
void build_gauss_q8(int ksize, double sigma, int* K) {
    std::vector<float> cf(ksize);
    const double scale2X = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < ksize; ++i) {
        const double x = i - (ksize - 1) * 0.5;
        cf[i] = (float)exp(scale2X * x * x);
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < ksize; ++i) {
        cf[i] = (float)(cf[i] * sum);
        K[i] = cv_round((double)cf[i] * 256.0);
    }
}

template <typename T>
int dev_alloc(T** p, size_t count) {
    if (*p) { hipFree(*p); *p = nullptr; }
    if (count == 0) count = 1;
    PSL_HIP(hipMalloc((void**)p, count * sizeof(T)));
    return PSLFE_OK;
}

}  // namespace

struct pslfe_orb {
    pslfe_ctx* ctx = nullptr;
    int nfeatures = 0, nlevels = 0, iniTh = 0, minTh = 0, max_batch = 1;
    double scaleFactor = 1.2;  // include/ORBextractor.h:98: the member is a double
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> quota;
    int umax[16];

    int gw = 0, gh = 0;  // geometry currently prepared
    OrbParams P;
    size_t pyr_fstride = 0, blur_fstride = 0;
    int oct_bs = 256;

    uint8_t* d_pyr = nullptr;
    uint8_t* d_blur = nullptr;
    int* d_cellcnt = nullptr;
    int* d_celloff = nullptr;
    uint32_t* d_cellcand = nullptr;
    uint32_t* d_cand = nullptr;
    uint16_t* d_knode = nullptr;
    uint32_t* d_lvlkp = nullptr;
    int* d_lvlcnt = nullptr;
    PslKeyPoint* d_kps = nullptr;
    uint8_t* d_desc = nullptr;
    int* d_counts = nullptr;
    uint32_t* d_celltab = nullptr;  // [ncells] level | row << 8 | column << 20
    int* d_xofs[PSLFE_MAX_LEVELS] = {};
    short2* d_alpha[PSLFE_MAX_LEVELS] = {};
    int* d_yofs[PSLFE_MAX_LEVELS] = {};
    short2* d_beta[PSLFE_MAX_LEVELS] = {};
    uint8_t* d_in = nullptr;  // staging for the host-buffer entry points
    size_t in_fstride = 0;
    int in_pitch = 0;
    int last_nframes = 0;
    FrameSrc last_src = {};

    void release() {
        hipFree(d_pyr); hipFree(d_blur); hipFree(d_cellcnt); hipFree(d_celloff); hipFree(d_cellcand);
        hipFree(d_cand); hipFree(d_knode); hipFree(d_lvlkp); hipFree(d_lvlcnt); hipFree(d_kps);
        hipFree(d_desc); hipFree(d_counts); hipFree(d_in); hipFree(d_celltab);
        d_celltab = nullptr;
        d_pyr = d_blur = d_desc = d_in = nullptr;
        d_cellcnt = d_celloff = d_lvlcnt = d_counts = nullptr;
        d_cellcand = d_cand = d_lvlkp = nullptr;
        d_knode = nullptr; d_kps = nullptr;
        for (int l = 0; l < PSLFE_MAX_LEVELS; ++l) {
            hipFree(d_xofs[l]); hipFree(d_alpha[l]); hipFree(d_yofs[l]); hipFree(d_beta[l]);
            d_xofs[l] = d_yofs[l] = nullptr; d_alpha[l] = d_beta[l] = nullptr;
        }
    }

    // Derives every size the kernels need for a w x h input (src/ORBextractor.cc:765-787, 541-545,
    // 1107-1115) and (re)allocates the HBM buffers for max_batch frames.
    int prepare(int w, int h) {
        if (w == gw && h == gh) return PSLFE_OK;
        PSL_REQUIRE(w > 0 && h > 0 && w <= 4096 && h <= 4096, PSLFE_E_INVALID, "orb: image %dx%d out of range (<=4096)", w, h);
        OrbParams Q;
        memset(&Q, 0, sizeof(Q));
        Q.nlevels = nlevels; Q.iniTh = std::min(std::max(iniTh, 0), 255); Q.minTh = std::min(std::max(minTh, 0), 255);
        build_gauss_q8(7, 2.0, Q.blurK);
        for (int v = 0; v < 16; ++v) Q.umax[v] = umax[v];
        size_t pyr_off = 0, blur_off = 0;
        int cell_off = 0, cand_off = 0, kp_off = 0, tile_off = 0, cellcap = 1, max_kpcap = 0;
        for (int l = 0; l < nlevels; ++l) {
            OrbLevelP& L = Q.lv[l];
            const float s = invScale[l];
            L.w = cv_round((float)w * s);
            L.h = cv_round((float)h * s);
            L.pitch = (int)psl_align_up(L.w, 16);
            L.img_off = (unsigned)pyr_off;
            if (l > 0) pyr_off += psl_align_up((size_t)L.pitch * L.h, 256);
            L.blur_off = (unsigned)blur_off;
            blur_off += psl_align_up((size_t)L.pitch * L.h, 256);
            L.maxBX = L.w - PSL_EDGE;
            L.maxBY = L.h - PSL_EDGE;
            const float width = (float)(L.maxBX - PSL_EDGE), height = (float)(L.maxBY - PSL_EDGE);
            const float W = 30;
            L.nCols = (int)(width / W);
            L.nRows = (int)(height / W);
            PSL_REQUIRE(L.nCols >= 1 && L.nRows >= 1, PSLFE_E_INVALID,
                        "orb: pyramid level %d (%dx%d) is smaller than one 30-px FAST cell plus borders; "
                        "the reference divides by zero here", l, L.w, L.h);
            L.wCell = (int)ceil(width / L.nCols);
            L.hCell = (int)ceil(height / L.nRows);
            PSL_REQUIRE(L.wCell <= PSL_MAXCELL && L.hCell <= PSL_MAXCELL, PSLFE_E_INVALID, "orb: FAST cell %dx%d too large", L.wCell, L.hCell);
            cellcap = std::max(cellcap, ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2));
            L.cell_off = cell_off;
            cell_off += L.nCols * L.nRows;
            L.quota = quota[l];
            L.nIni = (int)roundf((float)(L.maxBX - PSL_EDGE) / (L.maxBY - PSL_EDGE));
            PSL_REQUIRE(L.nIni >= 1, PSLFE_E_INVALID,
                        "orb: level %d is more than twice as tall as wide; the reference divides by zero here", l);
            L.hX = (float)(L.maxBX - PSL_EDGE) / L.nIni;
            L.kp_cap = std::max(L.quota + 3, 4 * L.nIni);
            max_kpcap = std::max(max_kpcap, L.kp_cap);
            L.kp_off = kp_off;
            kp_off += L.kp_cap;
            L.scale = scale[l];
            L.kpsize = (float)(int)(31 * scale[l]);
            L.tiles_x = (L.w + 63) / 64;
            L.tiles_y = (L.h + PSL_BLUR_TH - 1) / PSL_BLUR_TH;
            L.tile_off = tile_off;
            tile_off += L.tiles_x * L.tiles_y;
        }
        PSL_REQUIRE(max_kpcap <= 512, PSLFE_E_INVALID,
                    "orb: %d features on one level exceed the 512-node octree workgroup (nfeatures too large)", max_kpcap);
        oct_bs = max_kpcap <= 256 ? 256 : 512;
        Q.ncells = cell_off;
        Q.cellcap = cellcap;
        for (int l = 0; l < nlevels; ++l) {
            OrbLevelP& L = Q.lv[l];
            L.cand_off = cand_off;
            L.cand_cap = L.nCols * L.nRows * cellcap;
            cand_off += L.cand_cap;
        }
        Q.cand_total = cand_off;
        Q.kp_total = kp_off;
        Q.out_cap = kp_off;
        Q.ntiles = tile_off;

        PSL_HIP(hipSetDevice(ctx->device));
        PSL_HIP(hipStreamSynchronize(ctx->stream));
        // From here on buffers are freed and re-allocated one by one: forget the old geometry first, so that a failure below
        // (dev_alloc returns early) cannot leave `w == gw && h == gh` true over freed or undersized buffers.
        gw = gh = 0;
        last_nframes = 0;
        const int rc_alloc = allocate(Q, w, h, pyr_off, blur_off, cellcap);
        if (rc_alloc) { release(); (void)hipGetLastError(); return rc_alloc; }  // the failed hipMalloc must not surface in a later hipGetLastError()
        P = Q;
        gw = w; gh = h;
        return PSLFE_OK;
    }

    int allocate(const OrbParams& Q, int w, int h, size_t pyr_off, size_t blur_off, int cellcap) {
        const size_t F = (size_t)max_batch;
        pyr_fstride = psl_align_up(pyr_off, 256);
        blur_fstride = psl_align_up(blur_off, 256);
        int rc;
        if ((rc = dev_alloc(&d_pyr, pyr_fstride * F))) return rc;
        if ((rc = dev_alloc(&d_blur, blur_fstride * F))) return rc;
        if ((rc = dev_alloc(&d_cellcnt, (size_t)Q.ncells * F))) return rc;
        if ((rc = dev_alloc(&d_celloff, (size_t)Q.ncells * F))) return rc;
        if ((rc = dev_alloc(&d_cellcand, (size_t)Q.ncells * cellcap * F))) return rc;
        if ((rc = dev_alloc(&d_cand, (size_t)Q.cand_total * F))) return rc;
        if ((rc = dev_alloc(&d_knode, (size_t)Q.cand_total * F))) return rc;
        if ((rc = dev_alloc(&d_lvlkp, (size_t)Q.kp_total * F))) return rc;
        if ((rc = dev_alloc(&d_lvlcnt, (size_t)nlevels * F))) return rc;
        if ((rc = dev_alloc(&d_kps, (size_t)Q.out_cap * F))) return rc;
        if ((rc = dev_alloc(&d_desc, (size_t)Q.out_cap * 32 * F))) return rc;
        if ((rc = dev_alloc(&d_counts, F))) return rc;
        in_pitch = (int)psl_align_up(w, 16);
        in_fstride = psl_align_up((size_t)in_pitch * h, 256);
        if ((rc = dev_alloc(&d_in, in_fstride * F))) return rc;
        {
            std::vector<uint32_t> tab((size_t)Q.ncells);
            for (int l = 0; l < nlevels; ++l)
                for (int c = 0; c < Q.lv[l].nCols * Q.lv[l].nRows; ++c)
                    tab[(size_t)Q.lv[l].cell_off + c] = (uint32_t)l | ((uint32_t)(c / Q.lv[l].nCols) << 8) | ((uint32_t)(c % Q.lv[l].nCols) << 20);
            if ((rc = dev_alloc(&d_celltab, tab.size()))) return rc;
            PSL_HIP(hipMemcpy(d_celltab, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        for (int l = 1; l < nlevels; ++l) {
            std::vector<int> xo, yo;
            std::vector<short> al, be;
            build_linear_table(Q.lv[l - 1].w, Q.lv[l].w, true, xo, al);
            build_linear_table(Q.lv[l - 1].h, Q.lv[l].h, false, yo, be);
            if ((rc = dev_alloc(&d_xofs[l], xo.size()))) return rc;
            if ((rc = dev_alloc(&d_alpha[l], xo.size()))) return rc;
            if ((rc = dev_alloc(&d_yofs[l], yo.size()))) return rc;
            if ((rc = dev_alloc(&d_beta[l], yo.size()))) return rc;
            PSL_HIP(hipMemcpy(d_xofs[l], xo.data(), xo.size() * sizeof(int), hipMemcpyHostToDevice));
            PSL_HIP(hipMemcpy(d_alpha[l], al.data(), al.size() * sizeof(short), hipMemcpyHostToDevice));
            PSL_HIP(hipMemcpy(d_yofs[l], yo.data(), yo.size() * sizeof(int), hipMemcpyHostToDevice));
            PSL_HIP(hipMemcpy(d_beta[l], be.data(), be.size() * sizeof(short), hipMemcpyHostToDevice));
        }
        return PSLFE_OK;
    }

    int run(const uint8_t* d_gray, int nframes, int w, int h, int stride, size_t frame_stride) {
        int rc = prepare(w, h);
        if (rc) return rc;
        PSL_HIP(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        FrameSrc S;
        S.img0 = d_gray; S.stride0 = stride; S.fstride0 = frame_stride; S.pyr = d_pyr; S.pyr_fstride = pyr_fstride;
        S.nframes = nframes;
        S.xcd = nframes >= 8 ? 1 : 0;
        const unsigned F = (unsigned)nframes;
        // grid over (items, frames): XCD-aware for many frames (orb_kernels.h: psl_item_frame)
        auto G = [&](unsigned items) { return S.xcd ? dim3(8, items, (F + 7) / 8) : dim3(items, F); };
        {
            PSL_STAGE_BEGIN(ctx, "orb.pyramid");
            for (int l = 1; l < nlevels; ++l) {
                const unsigned nb = (unsigned)(((P.lv[l].pitch + 63) / 64) * ((P.lv[l].h + PSL_PYR_BH - 1) / PSL_PYR_BH));
                k_pyr_resize_tiled<<<G(nb), 256, 0, st>>>(P, S, l, d_xofs[l], d_alpha[l], d_yofs[l], d_beta[l]);
            }
            PSL_STAGE_END(ctx, "orb.pyramid");
        }
        {
            PSL_STAGE_BEGIN(ctx, "orb.fast");
            k_fast_cells4<<<G(P.ncells), 256, 0, st>>>(P, S, d_celltab, d_cellcnt, d_cellcand, 0);
            PSL_STAGE_END(ctx, "orb.fast");
        }
        {
            PSL_STAGE_BEGIN(ctx, "orb.octree");
            if (oct_bs == 256)
                k_octree<256><<<dim3(nlevels, F), 256, 0, st>>>(P, d_cellcnt, d_cellcand, d_celloff, d_cand, d_knode, d_lvlkp, d_lvlcnt);
            else
                k_octree<512><<<dim3(nlevels, F), 512, 0, st>>>(P, d_cellcnt, d_cellcand, d_celloff, d_cand, d_knode, d_lvlkp, d_lvlcnt);
            PSL_STAGE_END(ctx, "orb.octree");
        }
        {
            PSL_STAGE_BEGIN(ctx, "orb.blur");
            k_blur7<<<G(P.ntiles), 256, 0, st>>>(P, S, d_blur, blur_fstride);
            PSL_STAGE_END(ctx, "orb.blur");
        }
        {
            PSL_STAGE_BEGIN(ctx, "orb.describe");
            k_orient_describe<<<G((P.out_cap + 3) / 4), 256, 0, st>>>(P, S, d_blur, blur_fstride, d_lvlkp, d_lvlcnt, d_kps, d_desc, d_counts);
            PSL_STAGE_END(ctx, "orb.describe");
        }
        PSL_HIP(hipGetLastError());
        last_nframes = nframes;
        last_src = S;
        return PSLFE_OK;
    }
};

// internal view of the last batch for pslfe_match.hip (pslfe_frame_set_from_orb)
int pslfe_orb_internal_last(pslfe_orb* orb, const PslKeyPoint** kps, const uint8_t** desc, const int** counts, int* cap,
                            int* nframes, pslfe_ctx** ctx) {
    PSL_REQUIRE(orb->last_nframes > 0, PSLFE_E_STATE, "no batch extracted yet");
    *kps = orb->d_kps; *desc = orb->d_desc; *counts = orb->d_counts; *cap = orb->P.out_cap; *nframes = orb->last_nframes;
    *ctx = orb->ctx;
    return PSLFE_OK;
}

extern "C" {

int pslfe_orb_create(pslfe_ctx* ctx, int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                     int max_batch, pslfe_orb** out) {
    PSL_REQUIRE(ctx && out, PSLFE_E_INVALID, "pslfe_orb_create: NULL argument");
    *out = nullptr;
    PSL_REQUIRE(nfeatures >= 1 && nlevels >= 1 && nlevels <= PSLFE_MAX_LEVELS && scaleFactor > 1.0f && max_batch >= 1 && max_batch <= 65535,
                PSLFE_E_INVALID, "pslfe_orb_create: nfeatures %d scaleFactor %g nlevels %d max_batch %d", nfeatures, scaleFactor, nlevels, max_batch);
    pslfe_orb* o = new pslfe_orb();
    o->ctx = ctx;
    o->nfeatures = nfeatures; o->nlevels = nlevels; o->iniTh = iniThFAST; o->minTh = minThFAST; o->max_batch = max_batch;
    o->scaleFactor = scaleFactor;
    // src/ORBextractor.cc:415-432
    o->scale.resize(nlevels); o->sigma2.resize(nlevels); o->invScale.resize(nlevels); o->invSigma2.resize(nlevels);
    o->scale[0] = 1.0f; o->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; ++i) {
        o->scale[i] = (float)(o->scale[i - 1] * o->scaleFactor);
        o->sigma2[i] = o->scale[i] * o->scale[i];
    }
    for (int i = 0; i < nlevels; ++i) { o->invScale[i] = 1.0f / o->scale[i]; o->invSigma2[i] = 1.0f / o->sigma2[i]; }
    // :435-446
    o->quota.resize(nlevels);
    const float factor = (float)(1.0f / o->scaleFactor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) {
        o->quota[l] = cv_round(nDesired);
        sum += o->quota[l];
        nDesired *= factor;
    }
    o->quota[nlevels - 1] = std::max(nfeatures - sum, 0);
    // :452-469
    const int HP = 15;
    int v, v0, vmax = cv_floor(HP * sqrt(2.f) / 2 + 1), vmin = cv_ceil(HP * sqrt(2.f) / 2);
    const double hp2 = HP * HP;
    for (v = 0; v <= vmax; ++v) o->umax[v] = cv_round(sqrt(hp2 - v * v));
    for (v = HP, v0 = 0; v >= vmin; --v) {
        while (o->umax[v0] == o->umax[v0 + 1]) ++v0;
        o->umax[v] = v0;
        ++v0;
    }
    *out = o;
    return PSLFE_OK;
}

void pslfe_orb_destroy(pslfe_orb* orb) {
    if (!orb) return;
    hipSetDevice(orb->ctx->device);
    hipStreamSynchronize(orb->ctx->stream);
    orb->release();
    delete orb;
}

int pslfe_orb_levels(const pslfe_orb* orb) { return orb ? orb->nlevels : PSLFE_E_INVALID; }
float pslfe_orb_scale_factor(const pslfe_orb* orb) { return orb ? (float)orb->scaleFactor : 0.f; }

int pslfe_orb_scale_factors(const pslfe_orb* orb, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2) {
    PSL_REQUIRE(orb, PSLFE_E_INVALID, "pslfe_orb_scale_factors: orb is NULL");
    for (int i = 0; i < orb->nlevels; ++i) {
        if (scale) scale[i] = orb->scale[i];
        if (inv_scale) inv_scale[i] = orb->invScale[i];
        if (sigma2) sigma2[i] = orb->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = orb->invSigma2[i];
    }
    return PSLFE_OK;
}

int pslfe_orb_features_per_level(const pslfe_orb* orb, int* quota) {
    PSL_REQUIRE(orb && quota, PSLFE_E_INVALID, "pslfe_orb_features_per_level: NULL argument");
    for (int i = 0; i < orb->nlevels; ++i) quota[i] = orb->quota[i];
    return PSLFE_OK;
}

int pslfe_orb_max_keypoints(pslfe_orb* orb, int w, int h) {
    PSL_REQUIRE(orb, PSLFE_E_INVALID, "pslfe_orb_max_keypoints: orb is NULL");
    int rc = orb->prepare(w, h);
    if (rc) return rc;
    return orb->P.out_cap;
}

int pslfe_orb_extract_batch_device(pslfe_orb* orb, const uint8_t* d_gray, int nframes, int w, int h, int stride, size_t frame_stride) {
    PSL_REQUIRE(orb && d_gray, PSLFE_E_INVALID, "pslfe_orb_extract_batch_device: NULL argument");
    PSL_REQUIRE(nframes >= 1 && nframes <= orb->max_batch, PSLFE_E_INVALID, "pslfe_orb_extract_batch_device: nframes %d (max_batch %d)", nframes, orb->max_batch);
    PSL_REQUIRE(stride >= w && (nframes == 1 || frame_stride >= (size_t)stride * h), PSLFE_E_INVALID, "pslfe_orb_extract_batch_device: strides");
    return orb->run(d_gray, nframes, w, h, stride, frame_stride);
}

int pslfe_orb_results_device(pslfe_orb* orb, const PslKeyPoint** d_kps, const uint8_t** d_desc, const int32_t** d_counts, int* kp_cap) {
    PSL_REQUIRE(orb, PSLFE_E_INVALID, "pslfe_orb_results_device: orb is NULL");
    PSL_REQUIRE(orb->last_nframes > 0, PSLFE_E_STATE, "pslfe_orb_results_device: no batch extracted yet");
    if (d_kps) *d_kps = orb->d_kps;
    if (d_desc) *d_desc = orb->d_desc;
    if (d_counts) *d_counts = orb->d_counts;
    if (kp_cap) *kp_cap = orb->P.out_cap;
    return PSLFE_OK;
}

int pslfe_orb_fetch(pslfe_orb* orb, int frame, PslKeyPoint* kps, uint8_t* desc, int cap, int* n) {
    PSL_REQUIRE(orb && n, PSLFE_E_INVALID, "pslfe_orb_fetch: NULL argument");
    PSL_REQUIRE(orb->last_nframes > 0, PSLFE_E_STATE, "pslfe_orb_fetch: no batch extracted yet");
    PSL_REQUIRE(frame >= 0 && frame < orb->last_nframes, PSLFE_E_INVALID, "pslfe_orb_fetch: frame %d of %d", frame, orb->last_nframes);
    PSL_HIP(hipSetDevice(orb->ctx->device));
    hipStream_t st = orb->ctx->stream;
    int cnt = 0;
    PSL_HIP(hipMemcpyAsync(&cnt, orb->d_counts + frame, sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    *n = cnt;
    PSL_REQUIRE(cnt <= cap, PSLFE_E_CAPACITY, "pslfe_orb_fetch: %d keypoints, capacity %d", cnt, cap);
    if (cnt > 0) {
        const size_t o = (size_t)frame * orb->P.out_cap;
        if (kps) PSL_HIP(hipMemcpyAsync(kps, orb->d_kps + o, (size_t)cnt * sizeof(PslKeyPoint), hipMemcpyDeviceToHost, st));
        if (desc) PSL_HIP(hipMemcpyAsync(desc, orb->d_desc + o * 32, (size_t)cnt * 32, hipMemcpyDeviceToHost, st));
        PSL_HIP(hipStreamSynchronize(st));
    }
    return PSLFE_OK;
}

int pslfe_orb_extract_batch(pslfe_orb* orb, const uint8_t* gray, int nframes, int w, int h, int stride, size_t frame_stride,
                            PslKeyPoint* kps, uint8_t* desc, int cap, int32_t* counts) {
    PSL_REQUIRE(orb && gray && counts, PSLFE_E_INVALID, "pslfe_orb_extract_batch: NULL argument");
    PSL_REQUIRE(nframes >= 1 && nframes <= orb->max_batch, PSLFE_E_INVALID, "pslfe_orb_extract_batch: nframes %d (max_batch %d)", nframes, orb->max_batch);
    PSL_REQUIRE(stride >= w, PSLFE_E_INVALID, "pslfe_orb_extract_batch: stride %d < width %d", stride, w);
    int rc = orb->prepare(w, h);
    if (rc) return rc;
    PSL_HIP(hipSetDevice(orb->ctx->device));
    hipStream_t st = orb->ctx->stream;
    for (int f = 0; f < nframes; ++f)
        PSL_HIP(hipMemcpy2DAsync(orb->d_in + (size_t)f * orb->in_fstride, orb->in_pitch, gray + (size_t)f * frame_stride, stride, w, h,
                                 hipMemcpyHostToDevice, st));
    rc = orb->run(orb->d_in, nframes, w, h, orb->in_pitch, orb->in_fstride);
    if (rc) return rc;
    PSL_HIP(hipMemcpyAsync(counts, orb->d_counts, (size_t)nframes * sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    for (int f = 0; f < nframes; ++f) {
        PSL_REQUIRE(counts[f] <= cap, PSLFE_E_CAPACITY, "pslfe_orb_extract_batch: frame %d has %d keypoints, capacity %d", f, counts[f], cap);
        if (counts[f] == 0) continue;
        const size_t o = (size_t)f * orb->P.out_cap;
        if (kps) PSL_HIP(hipMemcpyAsync(kps + (size_t)f * cap, orb->d_kps + o, (size_t)counts[f] * sizeof(PslKeyPoint), hipMemcpyDeviceToHost, st));
        if (desc) PSL_HIP(hipMemcpyAsync(desc + (size_t)f * cap * 32, orb->d_desc + o * 32, (size_t)counts[f] * 32, hipMemcpyDeviceToHost, st));
    }
    PSL_HIP(hipStreamSynchronize(st));
    return PSLFE_OK;
}

int pslfe_orb_extract(pslfe_orb* orb, const uint8_t* gray, int w, int h, int stride, PslKeyPoint* kps, uint8_t* desc, int cap, int* n) {
    PSL_REQUIRE(orb && n, PSLFE_E_INVALID, "pslfe_orb_extract: NULL argument");
    *n = 0;
    if (!gray || w <= 0 || h <= 0) return PSLFE_OK;  // src/ORBextractor.cc:1046: empty image -> silent return
    int32_t cnt = 0;
    int rc = pslfe_orb_extract_batch(orb, gray, 1, w, h, stride, (size_t)stride * h, kps, desc, cap, &cnt);
    *n = cnt;
    return rc;
}

// ---- stage taps for the parity tests ---------------------------------------------------------------
int pslfe_orb_debug_level_size(pslfe_orb* orb, int level, int* w, int* h) {
    PSL_REQUIRE(orb && w && h, PSLFE_E_INVALID, "pslfe_orb_debug_level_size: NULL argument");
    PSL_REQUIRE(orb->gw > 0, PSLFE_E_STATE, "pslfe_orb_debug_level_size: no geometry prepared");
    PSL_REQUIRE(level >= 0 && level < orb->nlevels, PSLFE_E_INVALID, "pslfe_orb_debug_level_size: level %d", level);
    *w = orb->P.lv[level].w; *h = orb->P.lv[level].h;
    return PSLFE_OK;
}

int pslfe_orb_debug_level_image(pslfe_orb* orb, int frame, int level, int blurred, uint8_t* out, int out_stride) {
    PSL_REQUIRE(orb && out, PSLFE_E_INVALID, "pslfe_orb_debug_level_image: NULL argument");
    PSL_REQUIRE(orb->last_nframes > 0, PSLFE_E_STATE, "pslfe_orb_debug_level_image: no batch extracted yet");
    PSL_REQUIRE(frame >= 0 && frame < orb->last_nframes && level >= 0 && level < orb->nlevels, PSLFE_E_INVALID, "pslfe_orb_debug_level_image: frame/level");
    const OrbLevelP& L = orb->P.lv[level];
    PSL_REQUIRE(out_stride >= L.w, PSLFE_E_INVALID, "pslfe_orb_debug_level_image: stride");
    PSL_HIP(hipSetDevice(orb->ctx->device));
    PSL_HIP(hipStreamSynchronize(orb->ctx->stream));
    const uint8_t* src;
    size_t pitch;
    if (blurred) { src = orb->d_blur + (size_t)frame * orb->blur_fstride + L.blur_off; pitch = L.pitch; }
    else if (level == 0) { src = orb->last_src.img0 + (size_t)frame * orb->last_src.fstride0; pitch = orb->last_src.stride0; }
    else { src = orb->d_pyr + (size_t)frame * orb->pyr_fstride + L.img_off; pitch = L.pitch; }
    PSL_HIP(hipMemcpy2D(out, out_stride, src, pitch, L.w, L.h, hipMemcpyDeviceToHost));
    return PSLFE_OK;
}

int pslfe_orb_debug_candidates(pslfe_orb* orb, int frame, int level, int32_t* xys, int cap, int* n) {
    PSL_REQUIRE(orb && n, PSLFE_E_INVALID, "pslfe_orb_debug_candidates: NULL argument");
    PSL_REQUIRE(orb->last_nframes > 0, PSLFE_E_STATE, "pslfe_orb_debug_candidates: no batch extracted yet");
    PSL_REQUIRE(frame >= 0 && frame < orb->last_nframes && level >= 0 && level < orb->nlevels, PSLFE_E_INVALID, "pslfe_orb_debug_candidates: frame/level");
    const OrbParams& P = orb->P;
    const OrbLevelP& L = P.lv[level];
    PSL_HIP(hipSetDevice(orb->ctx->device));
    PSL_HIP(hipStreamSynchronize(orb->ctx->stream));
    const int ncell = L.nCols * L.nRows;
    std::vector<int> cnt(ncell);
    PSL_HIP(hipMemcpy(cnt.data(), orb->d_cellcnt + (size_t)frame * P.ncells + L.cell_off, ncell * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<uint32_t> cells((size_t)ncell * P.cellcap);
    PSL_HIP(hipMemcpy(cells.data(), orb->d_cellcand + ((size_t)frame * P.ncells + L.cell_off) * P.cellcap, cells.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    int total = 0;
    for (int c = 0; c < ncell; ++c)
        for (int t = 0; t < cnt[c]; ++t, ++total)
            if (xys && total < cap) {
                const uint32_t k = cells[(size_t)c * P.cellcap + t];
                xys[3 * total] = k & 0xfff; xys[3 * total + 1] = (k >> 12) & 0xfff; xys[3 * total + 2] = k >> 24;
            }
    *n = total;
    return PSLFE_OK;
}

int pslfe_orb_debug_level_keypoints(pslfe_orb* orb, int frame, int level, int32_t* xys, int cap, int* n) {
    PSL_REQUIRE(orb && n, PSLFE_E_INVALID, "pslfe_orb_debug_level_keypoints: NULL argument");
    PSL_REQUIRE(orb->last_nframes > 0, PSLFE_E_STATE, "pslfe_orb_debug_level_keypoints: no batch extracted yet");
    PSL_REQUIRE(frame >= 0 && frame < orb->last_nframes && level >= 0 && level < orb->nlevels, PSLFE_E_INVALID, "pslfe_orb_debug_level_keypoints: frame/level");
    const OrbParams& P = orb->P;
    const OrbLevelP& L = P.lv[level];
    PSL_HIP(hipSetDevice(orb->ctx->device));
    PSL_HIP(hipStreamSynchronize(orb->ctx->stream));
    int cnt = 0;
    PSL_HIP(hipMemcpy(&cnt, orb->d_lvlcnt + (size_t)frame * P.nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
    std::vector<uint32_t> k(std::max(cnt, 1));
    PSL_HIP(hipMemcpy(k.data(), orb->d_lvlkp + (size_t)frame * P.kp_total + L.kp_off, (size_t)cnt * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int t = 0; t < cnt && xys && t < cap; ++t) {
        xys[3 * t] = k[t] & 0xfff; xys[3 * t + 1] = (k[t] >> 12) & 0xfff; xys[3 * t + 2] = k[t] >> 24;
    }
    *n = cnt;
    return PSLFE_OK;
}

}  // extern "C"
