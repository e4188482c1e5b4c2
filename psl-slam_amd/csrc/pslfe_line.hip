// libpslfe: line extractor object (== ORB_SLAM2::LINEextractor) over the HIP kernels. Product code.
// Reference: add_src/LineExtractor.cpp:6-25, 325-366; add_inc/LineExtractor.h:160-255.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "line_kernels2.h"
#include "line_kernels3.h"
#include "pslfe_internal.h"

#ifndef PSL_NFA_COUNT_WGS
#define PSL_NFA_COUNT_WGS 4   // workgroups (16 scan groups each) per frame of a many-frames k_lsd_nfa_count launch: 12288 dense frames, 8 waves per SIMD: 1: 38.2 ms, 2: 37.3, 3: 36.9,
                              // 4: 36.7; at 4 waves per SIMD 4: 38.1, 8: 39.9 (the default until round 3), 16: 45.2, 32: 59.1 (profiles/r03z_ab_nfa_grid.log)
#endif
#ifndef PSL_GROW_LDS_USED
#define PSL_GROW_LDS_USED 1                  // launches with helper waves keep the `used` bits in LDS (0: in memory, as the many-frames launches do; A/B)
#endif
#define PSL_GROW_LDS_USED_MAX (144u * 1024u)   // of the CU's 160 KB (the kernel's own arrays take 4 KB): scaled images up to ~1.18 M pixels
#ifndef PSL_GROW_HELPER_FRAMES
#define PSL_GROW_HELPER_FRAMES 64   // launches of at most this many frames run k_lsd_grow4 with helper waves (measured: tools/helper_sweep.sh)
#endif
struct pslfe_line {
    pslfe_ctx* ctx = nullptr;
    int numOctaves = 1, nfeatures = 200, max_batch = 1;
    float scale = 1.2f;
    double min_line_length = 0;
    int refine = 2;  // LSD_REFINE_ADV (what the stock contrib LSDDetector constructs), 1 = LSD_REFINE_STD: pslfe_line_set_refine
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
    float2* d_trig = nullptr;     // (cosf, sinf) of the level-line angle per scaled pixel
    float2* d_seedt = nullptr;
    uint8_t* d_used = nullptr;    // LSD `used` map, one byte per scaled pixel
    uint32_t* d_reg = nullptr;
    LsdnTables NT = {};           // LSD_REFINE_ADV: log_gamma / log(p) tables of nfa() (NT.lg in HBM)
    double* d_lgamma = nullptr;
    size_t lds_used_attr = 64u * 1024u;   // dynamic LDS k_lsd_grow4<3, 1> has been allowed so far
    double* d_sctab = nullptr;    // psl_sincostab.inc
    double* d_rects = nullptr;    // LSD_REFINE_ADV: rectangles of k_lsd_grow4 for k_lsd_nfa
    int* d_nrect = nullptr;
    int* d_weight = nullptr;   // [F] defined pixels per frame (k_lsd_grad) and [F] the frames by decreasing weight (k_frame_order)
    int* d_order = nullptr;
    float* d_segtmp = nullptr;
    uint8_t* d_keep = nullptr;
    int2* d_counts = nullptr;     // (n, k) of the five trial rectangles of a rect_improve phase
    double* d_vals = nullptr;     // their nfa() values
    double2* d_sstate = nullptr;  // (first term, p / (1 - p)) of the binomial tails that have to be summed
    LsdnSeries* d_slist = nullptr;  // ... as one list per frame, by predicted length (d_stmp: in item order, before the bucketing)
    LsdnSeries* d_stmp = nullptr;
    int* d_lcount = nullptr;        // class offsets in the list, [F][PSL_NFA_NCLS + 1]
    float* d_seg = nullptr;
    int* d_nseg = nullptr;
    MergeScratch M = {};
    PslKeyLine* d_kls = nullptr;
    uint8_t* d_ldesc = nullptr;
    float* d_fdesc = nullptr;
    double* d_lineEq = nullptr;
    int* d_nkl = nullptr;
    int* d_status = nullptr;
    short2* d_dxy = nullptr;
    float* d_rawfans = nullptr;
    float* d_fans = nullptr;
    int* d_nfans = nullptr;
    float* d_tmplines = nullptr;  // [NMAX][4] staging for the host-pointer pairing entry point

    void release() {
        hipFree(M.lines0); hipFree(M.lines1); hipFree(M.merged); hipFree(M.angles); hipFree(M.length); hipFree(M.order); hipFree(M.pos);
        hipFree(M.adj); hipFree(M.code); hipFree(M.clist); hipFree(M.coff); hipFree(M.work); hipFree(M.bits); hipFree(M.stage);
        M = MergeScratch{};
        hipFree(d_kls); hipFree(d_ldesc); hipFree(d_fdesc); hipFree(d_lineEq); hipFree(d_nkl); hipFree(d_status); hipFree(d_dxy);
        hipFree(d_rawfans); hipFree(d_fans); hipFree(d_nfans); hipFree(d_tmplines);
        d_kls = nullptr; d_ldesc = nullptr; d_fdesc = nullptr; d_lineEq = nullptr; d_nkl = nullptr; d_status = nullptr; d_dxy = nullptr;
        d_rawfans = nullptr; d_fans = nullptr; d_nfans = nullptr; d_tmplines = nullptr;
        hipFree(d_trig); d_trig = nullptr;
        hipFree(d_seedt); d_seedt = nullptr;
        hipFree(d_used); d_used = nullptr;
        hipFree(d_in); hipFree(d_scaled); hipFree(d_angdeg); hipFree(d_modgrad); hipFree(d_reg);
        hipFree(d_seg); hipFree(d_nseg); hipFree(d_rects); hipFree(d_nrect); hipFree(d_weight); hipFree(d_order); hipFree(d_segtmp); hipFree(d_keep); hipFree(d_lgamma); hipFree(d_sctab);
        d_sctab = nullptr; hipFree(d_counts); hipFree(d_vals); hipFree(d_sstate); hipFree(d_slist); hipFree(d_stmp); hipFree(d_lcount);
        d_counts = nullptr; d_vals = nullptr; d_sstate = nullptr; d_slist = nullptr; d_stmp = nullptr; d_lcount = nullptr;
        d_lgamma = nullptr;
        d_in = nullptr; d_scaled = nullptr; d_angdeg = nullptr; d_modgrad = nullptr; d_reg = nullptr;
        d_seg = nullptr; d_nseg = nullptr; d_rects = nullptr; d_nrect = nullptr; d_weight = nullptr; d_order = nullptr; d_segtmp = nullptr; d_keep = nullptr;
        gw = gh = 0;   // no geometry is prepared any more: the next call allocates again (or fails again) instead of
        last_nframes = 0;  // launching on freed memory
    }

    // allocation failure inside prepare(): leave the object empty, not half-built (the early return `w == gw && h == gh` of the
    // next call must not see the old geometry with freed or undersized buffers)
    int fail_prepare(hipError_t e, const char* what) {
        release();
        (void)hipGetLastError();
        pslfe_set_error("line: allocating %s for %d frames failed: %s", what, max_batch, hipGetErrorString(e));
        return PSLFE_E_HIP;
    }

    int prepare(int w, int h) {
        if (w == gw && h == gh) return PSLFE_OK;
        PSL_REQUIRE(w >= 16 && h >= 16 && w <= 8192 && h <= 8192, PSLFE_E_INVALID, "line: image %dx%d out of range", w, h);
        LineParams Q;
        memset(&Q, 0, sizeof(Q));
        Q.w = w; Q.h = h;
        Q.W = (int)nearbyint(w * 0.8);
        Q.H = (int)nearbyint(h * 0.8);
        Q.maxseg = PSL_MERGE_NMAX;
        Q.maxkl = std::max(1024, nfeatures);
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
        {   // largest q with sqrt(q) <= rho (sqrt correctly rounded and monotone, on the host as on the device): the
            // gradient threshold can then be decided on the squared magnitude, without a square root
            double q = Q.rho * Q.rho;
            while (sqrt(nextafter(q, INFINITY)) <= Q.rho) q = nextafter(q, INFINITY);
            while (sqrt(q) > Q.rho) q = nextafter(q, 0.0);
            Q.rho_q = q;
        }
        const double LOG_NT = 5 * (log10((double)Q.W) + log10((double)Q.H)) / 2 + log10(11.0);
        Q.min_reg_size = (int)(size_t)(-LOG_NT / log10(Q.p));
        Q.log_nt = LOG_NT;
        Q.refine = refine;
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
        release();  // also forgets the old geometry: a failure below leaves an empty object, never a half-built one
#define PSL_ALLOC(ptr, bytes)                                                   \
    do {                                                                        \
        const hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));               \
        if (e_ != hipSuccess) return fail_prepare(e_, #ptr);                    \
    } while (0)
        const size_t F = (size_t)max_batch, npx = (size_t)Q.W * Q.H;
        in_pitch = (int)psl_align_up(w, 16);
        in_fstride = psl_align_up((size_t)in_pitch * h, 256);
        PSL_ALLOC(d_in, in_fstride * F);
        PSL_ALLOC(d_scaled, npx * std::min<size_t>(F, PSL_LSD_SUBBATCH) * sizeof(double));   // one sub-batch of the f64 working image (run_lsd)
        PSL_ALLOC(d_angdeg, npx * F * sizeof(float));
        PSL_ALLOC(d_modgrad, npx * F * sizeof(double));
        PSL_ALLOC(d_trig, npx * F * sizeof(float2));
        PSL_ALLOC(d_seedt, npx * F * sizeof(float2));
        PSL_ALLOC(d_used, npx * F);
        PSL_ALLOC(d_reg, npx * F * sizeof(uint32_t));
        PSL_ALLOC(d_seg, (size_t)Q.maxseg * 4 * sizeof(float) * F);
        PSL_ALLOC(d_nseg, F * sizeof(int));
        PSL_ALLOC(d_rects, (size_t)Q.maxseg * PSL_LSD_RECT_F64 * sizeof(double) * F);
        PSL_ALLOC(d_nrect, F * sizeof(int));
        PSL_ALLOC(d_weight, F * sizeof(int));
        PSL_ALLOC(d_order, F * sizeof(int));
        PSL_ALLOC(d_segtmp, (size_t)Q.maxseg * 4 * sizeof(float) * F);
        PSL_ALLOC(d_keep, (size_t)Q.maxseg * F);
        PSL_ALLOC(d_counts, (size_t)Q.maxseg * 5 * sizeof(int2) * F);
        PSL_ALLOC(d_vals, (size_t)Q.maxseg * 5 * sizeof(double) * F);
        PSL_ALLOC(d_sstate, (size_t)Q.maxseg * 5 * sizeof(double2) * F);
        PSL_ALLOC(d_slist, (size_t)Q.maxseg * 5 * sizeof(LsdnSeries) * F);
        PSL_ALLOC(d_stmp, (size_t)Q.maxseg * 5 * sizeof(LsdnSeries) * F);
        PSL_ALLOC(d_lcount, F * (PSL_NFA_NCLS + 1) * sizeof(int));
        {
            static const double sctab[444] = {
#include "psl_sincostab.inc"
            };
            PSL_ALLOC(d_sctab, sizeof(sctab));
            const hipError_t e_ = hipMemcpy(d_sctab, sctab, sizeof(sctab), hipMemcpyHostToDevice);
            if (e_ != hipSuccess) return fail_prepare(e_, "d_sctab (upload)");
            Q.sctab = d_sctab;
        }
        {   // nfa() tables: the same functions the device would evaluate, here on the host (bit-identical: single IEEE operations)
            const int lgn = 1 << 16;
            std::vector<double> lg((size_t)lgn, 0.0);
            for (int i = 1; i < lgn; ++i) lg[i] = lsdn_log_gamma((double)i);
            double pj = Q.p;
            for (int j = 0; j < PSL_NFA_NP; ++j, pj = pj / 2) {
                lg.push_back(0);  // placeholders, filled below: the three log tables follow the log_gamma table in the same allocation
            }
            lg.resize((size_t)lgn + 3 * PSL_NFA_NP + PSL_RATIO_BMAX);
            for (int i = 1; i < PSL_RATIO_BMAX; ++i) lg[(size_t)lgn + 3 * PSL_NFA_NP + i] = 1.0 / (double)i;   // psl_ratio_inv's table (k_lsd_nfa_series)
            pj = Q.p;
            for (int j = 0; j < PSL_NFA_NP; ++j, pj = pj / 2) {
                lg[(size_t)lgn + j] = psl_log(pj); lg[(size_t)lgn + PSL_NFA_NP + j] = psl_log(1.0 - pj); lg[(size_t)lgn + 2 * PSL_NFA_NP + j] = psl_log10(pj);
            }
            PSL_ALLOC(d_lgamma, lg.size() * sizeof(double));
            const hipError_t e_ = hipMemcpy(d_lgamma, lg.data(), lg.size() * sizeof(double), hipMemcpyHostToDevice);
            if (e_ != hipSuccess) return fail_prepare(e_, "d_lgamma (upload)");
            NT.lg = d_lgamma; NT.logs = d_lgamma + lgn; NT.lg_n = lgn; NT.p0 = Q.p; NT.log_nt = Q.log_nt; NT.inv = d_lgamma + lgn + 3 * PSL_NFA_NP;
        }
        const size_t N = PSL_MERGE_NMAX;
        PSL_ALLOC(M.lines0, F * N * 4 * sizeof(float));
        PSL_ALLOC(M.lines1, F * N * 4 * sizeof(float));
        PSL_ALLOC(M.merged, F * N * 4 * sizeof(float));
        PSL_ALLOC(M.angles, F * N * sizeof(float));
        PSL_ALLOC(M.length, F * N * sizeof(float));
        PSL_ALLOC(M.order, F * N * sizeof(int));
        PSL_ALLOC(M.pos, F * N * sizeof(int));
        PSL_ALLOC(M.adj, F * N * (N / 32) * sizeof(uint32_t));
        PSL_ALLOC(M.code, F * N * sizeof(int));
        PSL_ALLOC(M.clist, F * PSL_MERGE_CLMAX * sizeof(int));
        PSL_ALLOC(M.coff, F * (2 * N + 2) * sizeof(int));
        PSL_ALLOC(M.work, F * 4 * N * sizeof(int));
        PSL_ALLOC(M.bits, F * (N / 32) * sizeof(uint32_t));
        PSL_ALLOC(M.stage, F * N * sizeof(PslKeyLine));
        PSL_ALLOC(d_kls, F * Q.maxkl * sizeof(PslKeyLine));
        PSL_ALLOC(d_ldesc, F * Q.maxkl * 32);
        PSL_ALLOC(d_fdesc, F * Q.maxkl * 72 * sizeof(float));
        PSL_ALLOC(d_lineEq, F * Q.maxkl * 3 * sizeof(double));
        PSL_ALLOC(d_nkl, F * sizeof(int));
        PSL_ALLOC(d_status, F * sizeof(int));
        PSL_ALLOC(d_dxy, F * (size_t)w * h * sizeof(short2));
        PSL_ALLOC(d_rawfans, F * PSL_FAN_CAP * 4 * sizeof(float));
        PSL_ALLOC(d_fans, F * PSL_FAN_CAP * 4 * sizeof(float));
        PSL_ALLOC(d_nfans, F * sizeof(int));
        PSL_ALLOC(d_tmplines, N * 4 * sizeof(float));
#undef PSL_ALLOC
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
        {
            // LSD steps 1 + 2 in sub-batches of PSL_LSD_SUBBATCH frames: the f64 working image (1.57 MB per frame) only lives between the two kernels, so
            // d_scaled holds one sub-batch (3.2 GB instead of 19 GB at 12288 frames); 2048 frames x 192 tiles still fill the chip many times over
            PSL_HIP(hipMemsetAsync(d_used, 0, (size_t)P.W * P.H * F, st));  // the `used` map: 1 byte per scaled pixel
            P.singles = F <= PSL_GROW_HELPER_FRAMES;
            P.full_grad = nframes == 1;  // pslfe_line_debug_gradient reads the whole magnitude image of a single-frame call
            const bool ordered = PSL_FRAME_ORDER && F > PSL_GROW_HELPER_FRAMES;   // many-frames launches: k_lsd_grow4 takes the heaviest frames first
            if (ordered) PSL_HIP(hipMemsetAsync(d_weight, 0, (size_t)F * sizeof(int), st));
            const unsigned tx = (P.W + 63) / 64, ty = (P.H + 15) / 16, gy = (P.H + PSL_GRAD_TH - 1) / PSL_GRAD_TH;
            const size_t npx = (size_t)P.W * P.H;
            for (unsigned f0 = 0; f0 < F; f0 += PSL_LSD_SUBBATCH) {
                const unsigned n = std::min<unsigned>(PSL_LSD_SUBBATCH, F - f0);
                const int xcd = n >= 8 ? 1 : 0;
                const size_t po = (size_t)f0 * npx;
                {
                    PSL_STAGE_BEGIN(ctx, "line.lsd_scale");
                    k_lsd_scale_tiled<<<xcd ? dim3(8, tx * ty, (n + 7) / 8) : dim3(tx, ty, n), 256, 0, st>>>(P, d_gray + (size_t)f0 * frame_stride, stride, frame_stride,
                                                                                                             d_scaled, (int)n, xcd);
                    PSL_STAGE_END(ctx, "line.lsd_scale");
                }
                {
                    PSL_STAGE_BEGIN(ctx, "line.lsd_grad");
                    k_lsd_grad<<<xcd ? dim3(8, tx * gy, (n + 7) / 8) : dim3(tx, gy, n), 256, 0, st>>>(P, d_scaled, d_angdeg + po, d_modgrad + po, d_trig + po, d_seedt + po,
                                                                                                      d_used + po, ordered ? d_weight + f0 : nullptr, (int)n, xcd);
                    PSL_STAGE_END(ctx, "line.lsd_grad");
                }
            }
            if (ordered) k_frame_order<<<1, 1024, 0, st>>>(d_weight, (int)F, P.W * P.H, d_order);
        }
        P.refine = refine;
        {
            PSL_STAGE_BEGIN(ctx, "line.lsd_grow");
            // LSD_REFINE_ADV: the kernel leaves rectangles (d_rects / d_nrect) for the NFA validation below
            if (F <= PSL_GROW_HELPER_FRAMES) {  // few workgroups per XCD: three more waves each keep that XCD's L2 warm in front of the chain (line_kernels.h)
                const size_t ubytes = (((size_t)P.W * P.H + 31) >> 5) * 4;   // the `used` bits of the frame in LDS (24 KB at 640x480, 96 KB at 1280x960)
                if (PSL_GROW_LDS_USED && ubytes <= PSL_GROW_LDS_USED_MAX) {
                    if (ubytes > lds_used_attr) {   // more than the default 64 KB of dynamic LDS needs the attribute (once per size)
                        PSL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lsd_grow4<3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ubytes));
                        lds_used_attr = ubytes;
                    }
                    k_lsd_grow4<3, 1><<<F, 256, ubytes, st>>>(P, d_angdeg, d_modgrad, d_trig, d_used, d_seedt, d_reg, d_seg, refine >= 2 ? d_nrect : d_nseg, d_rects, (int)F, nullptr);
                } else {
                    k_lsd_grow4<3, 0><<<F, 256, 0, st>>>(P, d_angdeg, d_modgrad, d_trig, d_used, d_seedt, d_reg, d_seg, refine >= 2 ? d_nrect : d_nseg, d_rects, (int)F, nullptr);
                }
            } else
                k_lsd_grow4<0, 0><<<F, 64, 0, st>>>(P, d_angdeg, d_modgrad, d_trig, d_used, d_seedt, d_reg, d_seg, refine >= 2 ? d_nrect : d_nseg, d_rects, (int)F,
                                                 PSL_FRAME_ORDER ? d_order : nullptr);
            PSL_STAGE_END(ctx, "line.lsd_grow");
        }
        if (refine >= 2) {
            // rect_improve + NFA: per phase a pixel-scan launch (16 lanes = rectangle x trial), two nfa() launches (thread = evaluation:
            // set-up, then the binomial tails drawn from a shared counter) and a selection launch (thread = rectangle), line_kernels3.h.  A few hundred rectangles per frame: a many-frames launch fills
            // the chip by frames, a single frame by chunks.  (The stage timers record the launches of a phase as they are issued;
            // with profiling on, every stage costs two event records.)
            const dim3 gc(F >= 64 ? PSL_NFA_COUNT_WGS : 128, F), gs(F >= 64 ? 1 : 4, F);
#define PSL_NFA_PHASE(PH)                                                                                                \
    {                                                                                                                    \
        PSL_STAGE_BEGIN(ctx, "line.nfa_count");                                                                          \
        k_lsd_nfa_count<PH><<<gc, 256, 0, st>>>(P, d_angdeg, d_rects, d_nrect, d_keep, d_counts);                       \
        PSL_STAGE_END(ctx, "line.nfa_count");                                                                            \
    }                                                                                                                    \
    {                                                                                                                    \
        PSL_STAGE_BEGIN(ctx, "line.nfa_eval");                                                                           \
        k_lsd_nfa_setup<PH><<<F, 256, 0, st>>>(P, NT, d_rects, d_nrect, d_keep, d_counts, d_vals, d_sstate, d_stmp, d_slist, d_lcount); \
        k_lsd_nfa_series<PH><<<dim3(PSL_NFA_NCLS, (F + PSL_NFA_FG - 1) / PSL_NFA_FG), 256, 0, st>>>(P, NT, (int)F, d_slist, d_lcount, d_sstate); \
        k_lsd_nfa_select<PH><<<gs, 256, 0, st>>>(P, NT.log_nt, d_rects, d_nrect, d_keep, d_vals, d_sstate, d_segtmp);           \
        PSL_STAGE_END(ctx, "line.nfa_eval");                                                                             \
    }
            PSL_NFA_PHASE(PSL_NFA_FIRST)
            PSL_NFA_PHASE(-1)
            PSL_NFA_PHASE(0)
            PSL_NFA_PHASE(1)
            PSL_NFA_PHASE(2)
            PSL_NFA_PHASE(3)
#undef PSL_NFA_PHASE
            k_lsd_emit<<<F, 256, 0, st>>>(P, d_nrect, d_segtmp, d_keep, d_seg, d_nseg);
        }
        PSL_HIP(hipGetLastError());
        last_nframes = nframes;
        return PSLFE_OK;
    }

    // optimizeAndMergeLines_lsd + KeyLines + top-N + line equations on the segment lists in d_seg/d_nseg
    int run_merge(int nframes) {
        PSL_HIP(hipSetDevice(ctx->device));
        PSL_STAGE_BEGIN(ctx, "line.merge");
        k_line_merge<PSL_MERGE_LDSN_SMALL><<<nframes, 256, 0, ctx->stream>>>(P, M, d_seg, d_nseg, d_kls, d_lineEq, d_nkl, d_status);
        k_line_merge<PSL_MERGE_LDSN><<<nframes, 256, 0, ctx->stream>>>(P, M, d_seg, d_nseg, d_kls, d_lineEq, d_nkl, d_status);
        PSL_STAGE_END(ctx, "line.merge");
        PSL_HIP(hipGetLastError());
        return PSLFE_OK;
    }

    // BinaryDescriptor::compute on the keylines in d_kls/d_nkl
    int run_lbd(const uint8_t* d_gray, int nframes, int stride, size_t frame_stride, bool want_float) {
        PSL_HIP(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        {
            PSL_STAGE_BEGIN(ctx, "line.lbd_pre");
            const unsigned tx = (P.w + 63) / 64, ty = (P.h + 31) / 32;
            const int xcd = nframes >= 8 ? 1 : 0;
            k_lbd_pre<<<xcd ? dim3(8, tx * ty, (nframes + 7) / 8) : dim3(tx, ty, nframes), 256, 0, st>>>(P, d_gray, stride, frame_stride, d_dxy, nframes, xcd);
            PSL_STAGE_END(ctx, "line.lbd_pre");
        }
        {
            PSL_STAGE_BEGIN(ctx, "line.lbd");
            const int per_frame = std::min(P.maxkl, std::max(P.nfeatures, 1));
            k_lbd<<<dim3((per_frame + 3) / 4, nframes), 256, 0, st>>>(P, d_dxy, d_kls, d_nkl, d_ldesc, want_float ? d_fdesc : nullptr);
            PSL_STAGE_END(ctx, "line.lbd");
        }
        PSL_HIP(hipGetLastError());
        return PSLFE_OK;
    }

    int run_pair(int nframes, float radius, float fanThr) {
        PSL_HIP(hipSetDevice(ctx->device));
        PSL_STAGE_BEGIN(ctx, "line.pair");
        // mLines = (startPointX, startPointY, endPointX, endPointY) of every keyline (src/Frame.cc:355-373):
        // these are 4 consecutive floats at offset 7 of the 17-word KeyLine record
        k_lil_pair<<<nframes, 256, 0, ctx->stream>>>(reinterpret_cast<const float*>(d_kls) + 7, (size_t)P.maxkl * 17, 17, d_nkl, 0, radius, fanThr,
                                                     P.w, P.h, d_rawfans, d_fans, PSL_FAN_CAP, d_nfans);
        PSL_STAGE_END(ctx, "line.pair");
        PSL_HIP(hipGetLastError());
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
    // LINEextractor::operator() calls detect(image, kls, scale, numOctaves) (add_src/LineExtractor.cpp:336-337) whose `int scale` parameter
    // truncates the float member 1.2 to 1.  With numOctaves > 1 the stock contrib LSDDetector the reference links then builds its pyramid with
    // pyrDown(m, m, Size(cols / 1, rows / 1)) - a destination of the SOURCE's size, which pyrDown's size assertion (|2 dst - src| <= 2)
    // rejects with a cv::Exception (opencv_contrib 3.x LSDDetector.cpp: computeGaussianPyramid; the vendored twin, which the reference does not
    // call, discards that Size through a comma expression: Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:71).  So the reference's own
    // call cannot produce a result for numOctaves > 1 - every RGB-D YAML sets LINEextractor.nLevels: 1 - and an error here is its behaviour.
    PSL_REQUIRE(numOctaves == 1, PSLFE_E_INVALID,
                "pslfe_line_create: numOctaves %d: the reference's LSDDetector::detect(image, kls, (int)1.2f, numOctaves) throws for numOctaves > 1 "
                "(pyrDown to the source's own size); only numOctaves == 1 yields lines", numOctaves);
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

int pslfe_line_set_refine(pslfe_line* line, int refine) {
    PSL_REQUIRE(line, PSLFE_E_INVALID, "pslfe_line_set_refine: line is NULL");
    PSL_REQUIRE(refine == PSLFE_LSD_REFINE_STD || refine == PSLFE_LSD_REFINE_ADV, PSLFE_E_INVALID, "pslfe_line_set_refine: mode %d", refine);
    line->refine = refine;
    return PSLFE_OK;
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
    if (scaled) {   // the working image is kept for one sub-batch only (run_lsd)
        PSL_REQUIRE(line->last_nframes <= PSL_LSD_SUBBATCH, PSLFE_E_STATE, "pslfe_line_debug_gradient: the scaled image is kept for launches of at most %d frames", PSL_LSD_SUBBATCH);
        PSL_HIP(hipMemcpy(scaled, line->d_scaled + frame * npx, npx * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (angle_deg) PSL_HIP(hipMemcpy(angle_deg, line->d_angdeg + frame * npx, npx * sizeof(float), hipMemcpyDeviceToHost));
    if (modgrad) PSL_HIP(hipMemcpy(modgrad, line->d_modgrad + frame * npx, npx * sizeof(double), hipMemcpyDeviceToHost));
    return PSLFE_OK;
}


// ---- full extractor ------------------------------------------------------------------------------
int pslfe_line_extract_batch_device(pslfe_line* line, const uint8_t* d_gray, int nframes, int w, int h, int stride, size_t frame_stride) {
    PSL_REQUIRE(line && d_gray, PSLFE_E_INVALID, "pslfe_line_extract_batch_device: NULL argument");
    PSL_REQUIRE(nframes >= 1 && nframes <= line->max_batch, PSLFE_E_INVALID, "pslfe_line_extract_batch_device: nframes %d (max_batch %d)", nframes, line->max_batch);
    PSL_REQUIRE(stride >= w && (nframes == 1 || frame_stride >= (size_t)stride * h), PSLFE_E_INVALID, "pslfe_line_extract_batch_device: strides");
    int rc = line->run_lsd(d_gray, nframes, w, h, stride, frame_stride);
    if (rc) return rc;
    if ((rc = line->run_merge(nframes))) return rc;
    return line->run_lbd(d_gray, nframes, stride, frame_stride, false);
}

int pslfe_line_results_device(pslfe_line* line, const PslKeyLine** d_kls, const uint8_t** d_desc, const double** d_lineEq,
                              const int32_t** d_counts, int* kl_cap) {
    PSL_REQUIRE(line, PSLFE_E_INVALID, "pslfe_line_results_device: line is NULL");
    PSL_REQUIRE(line->last_nframes > 0, PSLFE_E_STATE, "pslfe_line_results_device: no batch extracted yet");
    if (d_kls) *d_kls = line->d_kls;
    if (d_desc) *d_desc = line->d_ldesc;
    if (d_lineEq) *d_lineEq = line->d_lineEq;
    if (d_counts) *d_counts = line->d_nkl;
    if (kl_cap) *kl_cap = line->P.maxkl;
    return PSLFE_OK;
}

int pslfe_line_fetch(pslfe_line* line, int frame, PslKeyLine* kls, uint8_t* desc, double* lineEq, int cap, int* n, int* status) {
    PSL_REQUIRE(line && n, PSLFE_E_INVALID, "pslfe_line_fetch: NULL argument");
    PSL_REQUIRE(line->last_nframes > 0 && frame >= 0 && frame < line->last_nframes, PSLFE_E_STATE, "pslfe_line_fetch: frame %d", frame);
    PSL_HIP(hipSetDevice(line->ctx->device));
    hipStream_t st = line->ctx->stream;
    int cnt = 0, stt = 0;
    PSL_HIP(hipMemcpyAsync(&cnt, line->d_nkl + frame, sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipMemcpyAsync(&stt, line->d_status + frame, sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    *n = cnt;
    if (status) *status = stt;
    PSL_REQUIRE(cnt <= cap, PSLFE_E_CAPACITY, "pslfe_line_fetch: %d keylines, capacity %d", cnt, cap);
    if (cnt > 0) {
        const size_t o = (size_t)frame * line->P.maxkl;
        if (kls) PSL_HIP(hipMemcpyAsync(kls, line->d_kls + o, (size_t)cnt * sizeof(PslKeyLine), hipMemcpyDeviceToHost, st));
        if (desc) PSL_HIP(hipMemcpyAsync(desc, line->d_ldesc + o * 32, (size_t)cnt * 32, hipMemcpyDeviceToHost, st));
        if (lineEq) PSL_HIP(hipMemcpyAsync(lineEq, line->d_lineEq + o * 3, (size_t)cnt * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
        PSL_HIP(hipStreamSynchronize(st));
    }
    return PSLFE_OK;
}

int pslfe_line_extract(pslfe_line* line, const uint8_t* gray, int w, int h, int stride, PslKeyLine* kls, uint8_t* desc, double* lineEq,
                       int cap, int* n) {
    PSL_REQUIRE(line && n, PSLFE_E_INVALID, "pslfe_line_extract: NULL argument");
    *n = 0;
    if (!gray || w <= 0 || h <= 0) return PSLFE_OK;  // add_src/LineExtractor.cpp:327: empty image -> silent return
    PSL_REQUIRE(stride >= w, PSLFE_E_INVALID, "pslfe_line_extract: stride %d < width %d", stride, w);
    int rc = line->upload(gray, 1, w, h, stride, (size_t)stride * h);
    if (rc) return rc;
    rc = pslfe_line_extract_batch_device(line, line->d_in, 1, w, h, line->in_pitch, line->in_fstride);
    if (rc) return rc;
    return pslfe_line_fetch(line, 0, kls, desc, lineEq, cap, n, nullptr);
}

// ---- stage entry points (also used by the parity tests) ---------------------------------------------
int pslfe_line_optimize_and_merge(pslfe_line* line, const float* segments, int nseg, int w, int h, PslKeyLine* kls, int cap, int* n) {
    PSL_REQUIRE(line && n && (nseg == 0 || segments), PSLFE_E_INVALID, "pslfe_line_optimize_and_merge: NULL argument");
    *n = 0;
    PSL_REQUIRE(nseg >= 0 && nseg <= PSL_MERGE_NMAX, PSLFE_E_CAPACITY, "pslfe_line_optimize_and_merge: %d segments (max %d)", nseg, PSL_MERGE_NMAX);
    int rc = line->prepare(w, h);
    if (rc) return rc;
    PSL_HIP(hipSetDevice(line->ctx->device));
    hipStream_t st = line->ctx->stream;
    if (nseg) PSL_HIP(hipMemcpyAsync(line->d_seg, segments, (size_t)nseg * 4 * sizeof(float), hipMemcpyHostToDevice, st));
    PSL_HIP(hipMemcpyAsync(line->d_nseg, &nseg, sizeof(int), hipMemcpyHostToDevice, st));
    PSL_HIP(hipStreamSynchronize(st));
    // the top-N cut belongs to LINEextractor::operator(); this entry point is optimizeAndMergeLines_lsd alone
    const int keep = line->P.nfeatures;
    line->P.nfeatures = line->P.maxkl;
    rc = line->run_merge(1);
    line->P.nfeatures = keep;
    if (rc) return rc;
    line->last_nframes = 1;
    return pslfe_line_fetch(line, 0, kls, nullptr, nullptr, cap, n, nullptr);
}

int pslfe_lbd_compute(pslfe_line* line, const uint8_t* gray, int w, int h, int stride, const PslKeyLine* kls, int nkl, uint8_t* desc,
                      float* fdesc) {
    PSL_REQUIRE(line && gray && (nkl == 0 || (kls && desc)), PSLFE_E_INVALID, "pslfe_lbd_compute: NULL argument");
    if (nkl == 0) return PSLFE_OK;  // upstream prints "keypoint list is empty" and returns (binary_descriptor_custom.cpp:559-563)
    PSL_REQUIRE(stride >= w, PSLFE_E_INVALID, "pslfe_lbd_compute: stride %d < width %d", stride, w);
    int rc = line->upload(gray, 1, w, h, stride, (size_t)stride * h);
    if (rc) return rc;
    PSL_REQUIRE(nkl <= line->P.maxkl, PSLFE_E_CAPACITY, "pslfe_lbd_compute: %d keylines (max %d)", nkl, line->P.maxkl);
    hipStream_t st = line->ctx->stream;
    PSL_HIP(hipMemcpyAsync(line->d_kls, kls, (size_t)nkl * sizeof(PslKeyLine), hipMemcpyHostToDevice, st));
    PSL_HIP(hipMemcpyAsync(line->d_nkl, &nkl, sizeof(int), hipMemcpyHostToDevice, st));
    PSL_HIP(hipStreamSynchronize(st));
    const int keep = line->P.nfeatures;
    line->P.nfeatures = std::max(keep, nkl);
    rc = line->run_lbd(line->d_in, 1, line->in_pitch, line->in_fstride, fdesc != nullptr);
    line->P.nfeatures = keep;
    if (rc) return rc;
    PSL_HIP(hipMemcpyAsync(desc, line->d_ldesc, (size_t)nkl * 32, hipMemcpyDeviceToHost, st));
    if (fdesc) PSL_HIP(hipMemcpyAsync(fdesc, line->d_fdesc, (size_t)nkl * 72 * sizeof(float), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    line->last_nframes = 1;
    return PSLFE_OK;
}

int pslfe_line_debug_sobel(pslfe_line* line, int frame, int16_t* dx, int16_t* dy) {
    PSL_REQUIRE(line && dx && dy, PSLFE_E_INVALID, "pslfe_line_debug_sobel: NULL argument");
    PSL_REQUIRE(line->last_nframes > 0 && frame >= 0 && frame < line->last_nframes, PSLFE_E_STATE, "pslfe_line_debug_sobel: frame %d", frame);
    PSL_HIP(hipSetDevice(line->ctx->device));
    PSL_HIP(hipStreamSynchronize(line->ctx->stream));
    const size_t npx = (size_t)line->P.w * line->P.h;
    std::vector<short2> tmp(npx);
    PSL_HIP(hipMemcpy(tmp.data(), line->d_dxy + frame * npx, npx * sizeof(short2), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < npx; ++i) { dx[i] = tmp[i].x; dy[i] = tmp[i].y; }
    return PSLFE_OK;
}

// == CPartiallyRecoverConnectivity(mLines, radius, fans, img, fanThr): host matrix in, fans rows out
int pslfe_lil_pair(pslfe_line* line, const float* lines, int nlines, float radius, float fanThr, int imgCols, int imgRows, float* fans,
                   int cap, int* nfans) {
    PSL_REQUIRE(line && nfans && (nlines == 0 || lines), PSLFE_E_INVALID, "pslfe_lil_pair: NULL argument");
    *nfans = 0;
    if (nlines == 0) return PSLFE_OK;
    PSL_REQUIRE(nlines <= PSL_MERGE_NMAX, PSLFE_E_CAPACITY, "pslfe_lil_pair: %d lines (max %d)", nlines, PSL_MERGE_NMAX);
    int rc = line->prepare(line->gw > 0 ? line->gw : std::max(imgCols, 16), line->gh > 0 ? line->gh : std::max(imgRows, 16));
    if (rc) return rc;
    PSL_HIP(hipSetDevice(line->ctx->device));
    hipStream_t st = line->ctx->stream;
    PSL_HIP(hipMemcpyAsync(line->d_tmplines, lines, (size_t)nlines * 4 * sizeof(float), hipMemcpyHostToDevice, st));
    PSL_HIP(hipStreamSynchronize(st));
    {
        PSL_STAGE_BEGIN(line->ctx, "line.pair");
        k_lil_pair<<<1, 256, 0, st>>>(line->d_tmplines, 0, 4, nullptr, nlines, radius, fanThr, imgCols, imgRows, line->d_rawfans, line->d_fans,
                                      PSL_FAN_CAP, line->d_nfans);
        PSL_STAGE_END(line->ctx, "line.pair");
    }
    PSL_HIP(hipGetLastError());
    int k = 0;
    PSL_HIP(hipMemcpyAsync(&k, line->d_nfans, sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    *nfans = k;
    PSL_REQUIRE(k <= cap, PSLFE_E_CAPACITY, "pslfe_lil_pair: %d fans, capacity %d", k, cap);
    if (k > 0 && fans) {
        PSL_HIP(hipMemcpyAsync(fans, line->d_fans, (size_t)k * 4 * sizeof(float), hipMemcpyDeviceToHost, st));
        PSL_HIP(hipStreamSynchronize(st));
    }
    return PSLFE_OK;
}

// Pairing of every frame of the last extracted batch, HBM resident (mLines = keyline endpoints, src/Frame.cc:504-505)
int pslfe_line_pair_batch_device(pslfe_line* line, float radius, float fanThr) {
    PSL_REQUIRE(line, PSLFE_E_INVALID, "pslfe_line_pair_batch_device: line is NULL");
    PSL_REQUIRE(line->last_nframes > 0, PSLFE_E_STATE, "pslfe_line_pair_batch_device: no batch extracted yet");
    return line->run_pair(line->last_nframes, radius, fanThr);
}

int pslfe_line_fans_device(pslfe_line* line, const float** d_fans, const int32_t** d_nfans, int* fan_stride) {
    PSL_REQUIRE(line, PSLFE_E_INVALID, "pslfe_line_fans_device: line is NULL");
    PSL_REQUIRE(line->last_nframes > 0, PSLFE_E_STATE, "pslfe_line_fans_device: no batch extracted yet");
    if (d_fans) *d_fans = line->d_fans;
    if (d_nfans) *d_nfans = line->d_nfans;
    if (fan_stride) *fan_stride = PSL_FAN_CAP;
    return PSLFE_OK;
}

int pslfe_line_fans_fetch(pslfe_line* line, int frame, float* fans, int cap, int* nfans) {
    PSL_REQUIRE(line && nfans, PSLFE_E_INVALID, "pslfe_line_fans_fetch: NULL argument");
    PSL_REQUIRE(line->last_nframes > 0 && frame >= 0 && frame < line->last_nframes, PSLFE_E_STATE, "pslfe_line_fans_fetch: frame %d", frame);
    PSL_HIP(hipSetDevice(line->ctx->device));
    hipStream_t st = line->ctx->stream;
    int k = 0;
    PSL_HIP(hipMemcpyAsync(&k, line->d_nfans + frame, sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    *nfans = k;
    PSL_REQUIRE(k <= cap, PSLFE_E_CAPACITY, "pslfe_line_fans_fetch: %d fans, capacity %d", k, cap);
    if (k > 0 && fans) {
        PSL_HIP(hipMemcpyAsync(fans, line->d_fans + (size_t)frame * PSL_FAN_CAP * 4, (size_t)k * 4 * sizeof(float), hipMemcpyDeviceToHost, st));
        PSL_HIP(hipStreamSynchronize(st));
    }
    return PSLFE_OK;
}


// == lmatcher.match(mLastFrame.mLdesc, mCurrentFrame.mLdesc, nnr, matches_12) (src/Tracking.cc:901 ->
//    LSDmatcher::match -> matchNNR, add_src/LSDmatcher.cpp:354-413) for every frame f of the last batch
//    against frame (f - shift) mod nframes; HBM resident.  d_matches12: [nframes][cap] (row f indexed by the
//    LAST frame's line, value = index into frame f's lines or -1), d_nmatches: [nframes].
int pslfe_line_match_batch_device(pslfe_line* line, int shift, float nnr, int32_t* d_matches12, int32_t* d_nmatches) {
    PSL_REQUIRE(line && d_matches12 && d_nmatches, PSLFE_E_INVALID, "pslfe_line_match_batch_device: NULL argument");
    PSL_REQUIRE(line->last_nframes > 0, PSLFE_E_STATE, "pslfe_line_match_batch_device: no batch extracted yet");
    PSL_HIP(hipSetDevice(line->ctx->device));
    hipStream_t st = line->ctx->stream;
    const int F = line->last_nframes, cap = line->P.maxkl;
    PSL_HIP(hipMemsetAsync(d_nmatches, 0, (size_t)F * sizeof(int), st));
    {
        PSL_STAGE_BEGIN(line->ctx, "line.match");
        const int qmax = std::min(cap, line->P.nfeatures);
        k_line_match_batch<<<dim3(F, (qmax + 3) / 4), 256, 0, st>>>(line->d_ldesc, line->d_nkl, cap, F, shift, nnr, d_matches12, d_nmatches);
        PSL_STAGE_END(line->ctx, "line.match");
    }
    PSL_HIP(hipGetLastError());
    return PSLFE_OK;
}

}  // extern "C"
