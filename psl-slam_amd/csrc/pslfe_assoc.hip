// libpslfe: the small per-frame association routines on top of the descriptor matchers. Product code.
//   LSDmatcher::SearchByGeomNApearance    add_src/LSDmatcher.cpp:36-110 (+ computeAngle2D :20-34)
//   LSDmatcher::FrameBFMatch              add_src/LSDmatcher.cpp:492-516 (+ lineDescriptorMAD :660-685)
//   Map::AssociatePlanesByBoundary        src/Map.cc:204-272 (live) /
//   InsectLineMatch::SearchMapInsectline  add_src/InsectlineMatch.cpp:9-59 (dead upstream, H14)
// Inputs are a few hundred lines / a few dozen planes: one small workgroup each, host pointers in/out.
#include <math.h>
#include <string.h>

#include <vector>

#include "pslfe_internal.h"
#include "psl_device_math.h"

// gates of SearchByGeomNApearance after matchNNR (:56-106)
__global__ __launch_bounds__(256) void k_line_geom_gate(const PslKeyLine* __restrict__ kl_last, int n1, const PslKeyLine* __restrict__ kl_cur, int n2,
                                                         const uint8_t* __restrict__ has_mapline, const int* __restrict__ knn_idx,
                                                         const int* __restrict__ knn_dist, float desc_th, double deltaWidth, double deltaHeight,
                                                         double cos_th_angle, int* __restrict__ matches12, int* __restrict__ assigned,
                                                         int* __restrict__ lmatches) {
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    for (int i = threadIdx.x; i < n2; i += 256) assigned[i] = -1;
    __syncthreads();
    for (int i1 = threadIdx.x; i1 < n1; i1 += 256) {
        // matchNNR (:354-376)
        int i2 = -1;
        if (n2 >= 2 && (float)knn_dist[2 * i1] < PSL_FMUL((float)knn_dist[2 * i1 + 1], desc_th)) i2 = knn_idx[2 * i1];
        int out = i2;
        if (has_mapline[i1] && i2 >= 0 && kl_cur[i2].startPointX != 0) {
            const PslKeyLine c = kl_cur[i2], l = kl_last[i1];
            const double vc0 = (double)PSL_FSUB(c.ePointInOctaveX, c.sPointInOctaveX), vc1 = (double)PSL_FSUB(c.ePointInOctaveY, c.sPointInOctaveY);
            const double vl0 = (double)PSL_FSUB(l.ePointInOctaveX, l.sPointInOctaveX), vl1 = (double)PSL_FSUB(l.ePointInOctaveY, l.sPointInOctaveY);
            const double dot = PSL_DADD(PSL_DMUL(vc0, vl0), PSL_DMUL(vc1, vl1));
            const double mA = __dsqrt_rn(PSL_DADD(PSL_DMUL(vc0, vc0), PSL_DMUL(vc1, vc1)));
            const double mB = __dsqrt_rn(PSL_DADD(PSL_DMUL(vl0, vl0), PSL_DMUL(vl1, vl1)));
            const double angle = fabs(dot / PSL_DMUL(mA, mB));
            if (angle < cos_th_angle) out = -1;
            else {
                const bool far_s = (double)__builtin_fabsf(PSL_FSUB(c.sPointInOctaveX, l.sPointInOctaveX)) > deltaWidth ||
                                   (double)__builtin_fabsf(PSL_FSUB(c.sPointInOctaveY, l.sPointInOctaveY)) > deltaHeight;
                const bool far_e = (double)__builtin_fabsf(PSL_FSUB(c.ePointInOctaveX, l.ePointInOctaveX)) > deltaWidth ||
                                   (double)__builtin_fabsf(PSL_FSUB(c.ePointInOctaveY, l.ePointInOctaveY)) > deltaHeight;
                if (far_s && far_e) out = -1;
                else { atomicMax(&assigned[i2], i1); atomicAdd(&s_n, 1); }  // later i1 overwrites earlier (:104)
            }
        }
        matches12[i1] = out;
    }
    __syncthreads();
    if (threadIdx.x == 0) *lmatches = s_n;
}

// FrameBFMatch after knnMatch: MAD of (d1 - d0), then the three gates (:503-515)
__global__ __launch_bounds__(256) void k_frame_bf_gate(const int* __restrict__ knn_idx, const int* __restrict__ knn_dist, int n1, float nnratio,
                                                        float TH, float* __restrict__ scratch, int* __restrict__ lineMatches) {
    __shared__ float s_med;
    float* d12 = scratch;        // [n1]
    float* dev = scratch + n1;   // [n1]
    const int tid = threadIdx.x;
    for (int i = tid; i < n1; i += 256) d12[i] = PSL_FSUB((float)knn_dist[2 * i + 1], (float)knn_dist[2 * i]);
    __syncthreads();
    for (int i = tid; i < n1; i += 256) {  // the element of rank n1/2 of the sorted values
        const float v = d12[i];
        int r = 0;
        for (int j = 0; j < n1; ++j) { const float u = d12[j]; r += (u < v) || (u == v && j < i); }
        if (r == n1 / 2) s_med = v;
    }
    __syncthreads();
    const double med = (double)s_med;
    for (int i = tid; i < n1; i += 256) dev[i] = __builtin_fabsf((float)PSL_DSUB((double)d12[i], med));
    __syncthreads();
    for (int i = tid; i < n1; i += 256) {
        const float v = dev[i];
        int r = 0;
        for (int j = 0; j < n1; ++j) { const float u = dev[j]; r += (u < v) || (u == v && j < i); }
        if (r == n1 / 2) s_med = v;
    }
    __syncthreads();
    const double nn12_th = PSL_DMUL(PSL_DMUL(1.4826, (double)s_med), 0.5);
    for (int i = tid; i < n1; i += 256) {
        const float a = (float)knn_dist[2 * i], b = (float)knn_dist[2 * i + 1];
        lineMatches[i] = ((double)PSL_FSUB(b, a) > nn12_th && a < TH && a < PSL_FMUL(nnratio, b)) ? knn_idx[2 * i] : -1;
    }
}

// plane association: sequential by definition (running threshold), tiny -> one thread
__global__ void k_associate_planes(const float* __restrict__ planes, const double* __restrict__ pts, int N, const float* __restrict__ map,
                                   const uint8_t* __restrict__ bad, int M, float dTh, float aTh, int live, int* __restrict__ assoc,
                                   int* __restrict__ nmatches) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int nm = 0;
    for (int i = 0; i < N; ++i) {
        assoc[i] = -1;
        const float* pM = planes + 4 * i;
        const double* P = pts + 15 * i;
        float ldTh = dTh;
        bool found = false;
        for (int j = 0; j < M; ++j) {
            if (!live && bad && bad[j]) continue;
            float w0 = map[4 * j], w1 = map[4 * j + 1], w2 = map[4 * j + 2], w3 = map[4 * j + 3];
            if (live && w3 < 0) { w0 = -w0; w1 = -w1; w2 = -w2; w3 = -w3; }
            const float angle = PSL_FADD(PSL_FADD(PSL_FMUL(pM[0], w0), PSL_FMUL(pM[1], w1)), PSL_FMUL(pM[2], w2));
            if (angle > aTh || angle < -aTh) {
                float d5[5];
                for (int k = 0; k < 5; ++k)
                    d5[k] = (float)PSL_DADD(PSL_DADD(PSL_DADD(PSL_DMUL((double)w0, P[3 * k]), PSL_DMUL((double)w1, P[3 * k + 1])), PSL_DMUL((double)w2, P[3 * k + 2])), (double)w3);
                const float dis = PSL_FADD(PSL_FADD(PSL_FADD(PSL_FADD(d5[0], d5[1]), d5[2]), d5[3]), d5[4]) / 5;
                if (live) {
                    if (__builtin_fabsf(dis) < dTh) { dTh = dis; assoc[i] = j; ++nm; }
                } else {
                    if (__builtin_fabsf(dis) < ldTh) { ldTh = dis; assoc[i] = j; found = true; }
                }
            }
        }
        if (!live && found) ++nm;
    }
    *nmatches = nm;
}

namespace {
struct DevBuf {  // the host-pointer entry points' device buffers: carved from the context's scratch arena (no hipMalloc / hipFree per call)
    pslfe_ctx* ctx;
    explicit DevBuf(pslfe_ctx* c) : ctx(c) {}
    template <typename T>
    T* up(const T* host, size_t count, hipStream_t st, hipError_t* e) { return psl_scratch_up(ctx, host, count, st, e); }
};
}  // namespace

extern "C" {

int pslfe_line_search_by_geom_appearance(pslfe_ctx* ctx, const PslKeyLine* kl_last, const uint8_t* desc_last, int n1, const PslKeyLine* kl_cur,
                                         const uint8_t* desc_cur, int n2, const uint8_t* has_mapline, float desc_th, float min_x, float max_x,
                                         float min_y, float max_y, int32_t* matches12, int32_t* assigned, int* lmatches) {
    PSL_REQUIRE(ctx && lmatches && (n1 == 0 || (kl_last && desc_last && has_mapline && matches12)) && (n2 == 0 || (kl_cur && desc_cur && assigned)),
                PSLFE_E_INVALID, "pslfe_line_search_by_geom_appearance: NULL argument");
    *lmatches = 0;
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    for (int i = 0; i < n2; ++i) assigned[i] = -1;
    if (n1 <= 0 || n2 <= 0) return PSLFE_OK;  // mLdesc.empty() -> 0 (:40-43)
    PSL_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    { const int rc_ = psl_scratch_begin(ctx); if (rc_) return rc_; }
    DevBuf B(ctx);
    hipError_t e = hipSuccess;
    PslKeyLine* dl = B.up(kl_last, n1, st, &e);
    PslKeyLine* dc = B.up(kl_cur, n2, st, &e);
    uint8_t* dd1 = B.up(desc_last, (size_t)n1 * 32, st, &e);
    uint8_t* dd2 = B.up(desc_cur, (size_t)n2 * 32, st, &e);
    uint8_t* dh = B.up(has_mapline, n1, st, &e);
    int* didx = B.up((const int*)nullptr, (size_t)n1 * 2, st, &e);
    int* ddist = B.up((const int*)nullptr, (size_t)n1 * 2, st, &e);
    int* dm = B.up((const int*)nullptr, n1, st, &e);
    int* da = B.up((const int*)nullptr, n2, st, &e);
    int* dn = B.up((const int*)nullptr, 1, st, &e);
    PSL_REQUIRE(e == hipSuccess, PSLFE_E_HIP, "pslfe_line_search_by_geom_appearance: %s", hipGetErrorString(e));
    int rc = pslfe_hamming_knn2_device(ctx, dd1, n1, dd2, n2, didx, ddist);
    if (rc) return rc;
    const double deltaWidth = (max_x - min_x) * 0.1, deltaHeight = (max_y - min_y) * 0.1;
    const double cos_th = cos(20.0 / 180.0 * M_PI);
    k_line_geom_gate<<<1, 256, 0, st>>>(dl, n1, dc, n2, dh, didx, ddist, desc_th, deltaWidth, deltaHeight, cos_th, dm, da, dn);
    PSL_HIP(hipGetLastError());
    PSL_HIP(hipMemcpyAsync(matches12, dm, (size_t)n1 * sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipMemcpyAsync(assigned, da, (size_t)n2 * sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipMemcpyAsync(lmatches, dn, sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    return PSLFE_OK;
}

int pslfe_line_frame_bf_match(pslfe_ctx* ctx, const uint8_t* desc1, int n1, const uint8_t* desc2, int n2, float nnratio, float TH,
                              int32_t* line_matches) {
    PSL_REQUIRE(ctx && (n1 == 0 || (desc1 && line_matches)), PSLFE_E_INVALID, "pslfe_line_frame_bf_match: NULL argument");
    for (int i = 0; i < n1; ++i) line_matches[i] = -1;
    if (n1 <= 0 || n2 < 2) return PSLFE_OK;  // knnMatch(k=2) needs two train rows; upstream UB (H12)
    PSL_REQUIRE(desc2, PSLFE_E_INVALID, "pslfe_line_frame_bf_match: desc2 is NULL");
    PSL_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    { const int rc_ = psl_scratch_begin(ctx); if (rc_) return rc_; }
    DevBuf B(ctx);
    hipError_t e = hipSuccess;
    uint8_t* dd1 = B.up(desc1, (size_t)n1 * 32, st, &e);
    uint8_t* dd2 = B.up(desc2, (size_t)n2 * 32, st, &e);
    int* didx = B.up((const int*)nullptr, (size_t)n1 * 2, st, &e);
    int* ddist = B.up((const int*)nullptr, (size_t)n1 * 2, st, &e);
    float* ds = B.up((const float*)nullptr, (size_t)n1 * 2, st, &e);
    int* dm = B.up((const int*)nullptr, n1, st, &e);
    PSL_REQUIRE(e == hipSuccess, PSLFE_E_HIP, "pslfe_line_frame_bf_match: %s", hipGetErrorString(e));
    int rc = pslfe_hamming_knn2_device(ctx, dd1, n1, dd2, n2, didx, ddist);
    if (rc) return rc;
    k_frame_bf_gate<<<1, 256, 0, st>>>(didx, ddist, n1, nnratio, TH, ds, dm);
    PSL_HIP(hipGetLastError());
    PSL_HIP(hipMemcpyAsync(line_matches, dm, (size_t)n1 * sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    return PSLFE_OK;
}

int pslfe_associate_planes(pslfe_ctx* ctx, const float* planes, const double* points, int nplanes, const float* map_planes, const uint8_t* map_bad,
                           int nmap, float dTh, float aTh, int live, int32_t* assoc, int* nmatches) {
    PSL_REQUIRE(ctx && nmatches && (nplanes == 0 || (planes && points && assoc)) && (nmap == 0 || map_planes), PSLFE_E_INVALID,
                "pslfe_associate_planes: NULL argument");
    *nmatches = 0;
    for (int i = 0; i < nplanes; ++i) assoc[i] = -1;
    if (nplanes <= 0 || nmap <= 0) return PSLFE_OK;
    PSL_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    { const int rc_ = psl_scratch_begin(ctx); if (rc_) return rc_; }
    DevBuf B(ctx);
    hipError_t e = hipSuccess;
    float* dp = B.up(planes, (size_t)nplanes * 4, st, &e);
    double* dq = B.up(points, (size_t)nplanes * 15, st, &e);
    float* dw = B.up(map_planes, (size_t)nmap * 4, st, &e);
    uint8_t* db = map_bad ? B.up(map_bad, nmap, st, &e) : nullptr;
    int* da = B.up((const int*)nullptr, nplanes, st, &e);
    int* dn = B.up((const int*)nullptr, 1, st, &e);
    PSL_REQUIRE(e == hipSuccess, PSLFE_E_HIP, "pslfe_associate_planes: %s", hipGetErrorString(e));
    k_associate_planes<<<1, 64, 0, st>>>(dp, dq, nplanes, dw, db, nmap, dTh, aTh, live, da, dn);
    PSL_HIP(hipGetLastError());
    PSL_HIP(hipMemcpyAsync(assoc, da, (size_t)nplanes * sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipMemcpyAsync(nmatches, dn, sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    return PSLFE_OK;
}

}  // extern "C"
