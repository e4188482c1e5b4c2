// libpslfe: grid-guided line matchers. Product code.
//   LineIterator                                  add_src/lineIterator.cpp:34-77
//   Frame::AssignFeaturesToGridForLine            src/Frame.cc:286-309
//   Frame::GetFeaturesInAreaForLine               src/Frame.cc:752-826
//   LSDmatcher::SearchByProjection(cur,last,th)   add_src/LSDmatcher.cpp:112-215  (mode 0)
//   LSDmatcher::SearchByProjection(F,MLs,..,th)   add_src/LSDmatcher.cpp:260-352  (mode 1)
// A frame holds <= a few hundred lines; one wave owns the frame.  Queries are processed in the
// reference's order (first-come-first-served on taken lines); for each query lane 0 assembles the
// candidate list exactly as GetFeaturesInAreaForLine does (3 probe points, de-duplicated, push order)
// and the 64 lanes evaluate the candidates in parallel (key = distance << 16 | list position).
#include <math.h>
#include <string.h>

#include <vector>

#include "pslfe_internal.h"
#include "psl_device_math.h"

#define PSL_LG_COLS 64
#define PSL_LG_ROWS 48
#define PSL_LG_CELLS (PSL_LG_COLS * PSL_LG_ROWS)
#define PSL_LINE_MAX 2048   // lines per frame handled by the matcher
#define PSL_LINE_TH 95      // hard-coded acceptance threshold (add_src/LSDmatcher.cpp:205, 342)

struct LineBres {  // add_src/lineIterator.cpp
    bool steep; double dx, dy, error; int maxX, ystep, y, x;
    __device__ void init(double x1, double y1, double x2, double y2) {
        steep = fabs(PSL_DSUB(y2, y1)) > fabs(PSL_DSUB(x2, x1));
        if (steep) { double t = x1; x1 = y1; y1 = t; t = x2; x2 = y2; y2 = t; }
        if (x1 > x2) { double t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
        dx = PSL_DSUB(x2, x1); dy = fabs(PSL_DSUB(y2, y1));
        error = dx / 2.0; ystep = (y1 < y2) ? 1 : -1;
        x = (int)x1; y = (int)y1; maxX = (int)x2;
    }
    __device__ bool next(int* px, int* py) {
        if (x > maxX) return false;
        if (steep) { *px = y; *py = x; } else { *px = x; *py = y; }
        error = PSL_DSUB(error, dy);
        if (error < 0) { y += ystep; error = PSL_DADD(error, dx); }
        x++;
        return true;
    }
};

struct LineMatchArgs {
    const PslKeyLine* kls; const uint8_t* desc; const double* eq; const double* dir3d; int n;
    float minX, minY, invW, invH;
    const PslLineQuery* q; const uint8_t* qdesc; int nq;
    const uint8_t* taken; float nnratio; double cos_gate;
    int* gstart; int* gidx; int gcap;   // CSR scratch in HBM: start [CELLS+1], idx [gcap]
    int* match; int* assigned; int* nmatches;
};

template <int MODE>
__global__ __launch_bounds__(64) void k_line_proj_match(LineMatchArgs A) {
    __shared__ int s_cnt[PSL_LG_CELLS + 1];
    __shared__ int s_owner[PSL_LINE_MAX];
    __shared__ uint8_t s_blocked[PSL_LINE_MAX];
    __shared__ uint16_t s_cand[PSL_LINE_MAX];
    __shared__ uint32_t s_seen[PSL_LINE_MAX / 32];
    __shared__ int s_ncand;
    const int lane = threadIdx.x;
    const int n = A.n < PSL_LINE_MAX ? A.n : PSL_LINE_MAX;
    // ---- AssignFeaturesToGridForLine: count, scan, fill, sort (lists end up ascending by line index)
    for (int c = lane; c <= PSL_LG_CELLS; c += 64) s_cnt[c] = 0;
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < n; i += 64) {
        LineBres it;
        it.init((double)PSL_FMUL(A.kls[i].startPointX, A.invW), (double)PSL_FMUL(A.kls[i].startPointY, A.invH),
                (double)PSL_FMUL(A.kls[i].endPointX, A.invW), (double)PSL_FMUL(A.kls[i].endPointY, A.invH));
        int px, py;
        while (it.next(&px, &py))
            if (px >= 0 && px < PSL_LG_COLS && py >= 0 && py < PSL_LG_ROWS) atomicAdd(&s_cnt[px * PSL_LG_ROWS + py], 1);
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        int acc = 0;
        for (int c = 0; c < PSL_LG_CELLS; ++c) { const int t = s_cnt[c]; s_cnt[c] = acc; A.gstart[c] = acc; acc += t; }
        s_cnt[PSL_LG_CELLS] = acc; A.gstart[PSL_LG_CELLS] = acc;
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < n; i += 64) {
        LineBres it;
        it.init((double)PSL_FMUL(A.kls[i].startPointX, A.invW), (double)PSL_FMUL(A.kls[i].startPointY, A.invH),
                (double)PSL_FMUL(A.kls[i].endPointX, A.invW), (double)PSL_FMUL(A.kls[i].endPointY, A.invH));
        int px, py;
        while (it.next(&px, &py))
            if (px >= 0 && px < PSL_LG_COLS && py >= 0 && py < PSL_LG_ROWS) {
                const int p = atomicAdd(&s_cnt[px * PSL_LG_ROWS + py], 1);
                if (p < A.gcap) A.gidx[p] = i;
            }
    }
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < PSL_LG_CELLS; c += 64) {  // s_cnt[c] is now the END of cell c; start from gstart
        const int lo = A.gstart[c], hi = min(s_cnt[c], A.gcap);
        for (int i = lo + 1; i < hi; ++i) {
            const int v = A.gidx[i];
            int j = i - 1;
            while (j >= lo && A.gidx[j] > v) { A.gidx[j + 1] = A.gidx[j]; --j; }
            A.gidx[j + 1] = v;
        }
    }
    for (int i = lane; i < n; i += 64) { s_owner[i] = -1; s_blocked[i] = (A.taken && A.taken[i]) ? 1 : 0; }
    __builtin_amdgcn_wave_barrier();

    int nmatches = 0;
    for (int qi = 0; qi < A.nq; ++qi) {
        const PslLineQuery q = A.q[qi];
        // ---- GetFeaturesInAreaForLine (literal, lane 0)
        if (lane == 0) {
            for (int w = 0; w < (n + 31) / 32; ++w) s_seen[w] = 0;
            int nc = 0;
            const float xs[3] = {q.x1, (float)((double)PSL_FADD(q.x1, q.x2) / 2.0), q.x2};
            const float ys[3] = {q.y1, (float)((double)PSL_FADD(q.y1, q.y2) / 2.0), q.y2};
            float d1x = PSL_FSUB(q.x1, q.x2), d1y = PSL_FSUB(q.y1, q.y2);
            const float n1 = sqrtf(PSL_FADD(PSL_FMUL(d1x, d1x), PSL_FMUL(d1y, d1y)));
            d1x = PSL_FDIV(d1x, n1); d1y = PSL_FDIV(d1y, n1);
            const float r = q.radius;
            for (int p = 0; p < 3; ++p) {
                const int minCX = max(0, (int)__builtin_floorf(PSL_FMUL(PSL_FSUB(PSL_FSUB(xs[p], A.minX), r), A.invW)));
                if (minCX >= PSL_LG_COLS) continue;
                const int maxCX = min(PSL_LG_COLS - 1, (int)__builtin_ceilf(PSL_FMUL(PSL_FADD(PSL_FSUB(xs[p], A.minX), r), A.invW)));
                if (maxCX < 0) continue;
                const int minCY = max(0, (int)__builtin_floorf(PSL_FMUL(PSL_FSUB(PSL_FSUB(ys[p], A.minY), r), A.invH)));
                if (minCY >= PSL_LG_ROWS) continue;
                const int maxCY = min(PSL_LG_ROWS - 1, (int)__builtin_ceilf(PSL_FMUL(PSL_FADD(PSL_FSUB(ys[p], A.minY), r), A.invH)));
                if (maxCY < 0) continue;
                for (int ix = minCX; ix <= maxCX; ++ix) {
                    const int e1 = min(A.gstart[ix * PSL_LG_ROWS + maxCY + 1], A.gcap);
                    for (int e = A.gstart[ix * PSL_LG_ROWS + minCY]; e < e1; ++e) {
                        const int j = A.gidx[e];
                        if ((s_seen[j >> 5] >> (j & 31)) & 1u) continue;
                        float d2x = PSL_FSUB(A.kls[j].startPointX, A.kls[j].endPointX), d2y = PSL_FSUB(A.kls[j].startPointY, A.kls[j].endPointY);
                        const float n2 = sqrtf(PSL_FADD(PSL_FMUL(d2x, d2x), PSL_FMUL(d2y, d2y)));
                        d2x = PSL_FDIV(d2x, n2); d2y = PSL_FDIV(d2y, n2);
                        const float cosS = __builtin_fabsf(PSL_FADD(PSL_FMUL(d1x, d2x), PSL_FMUL(d1y, d2y)));
                        if (cosS < q.th_cos) continue;
                        const float dist = (float)PSL_DADD(PSL_DADD(PSL_DMUL(A.eq[3 * j], (double)xs[p]), PSL_DMUL(A.eq[3 * j + 1], (double)ys[p])), A.eq[3 * j + 2]);
                        if (__builtin_fabsf(dist) < r) { s_cand[nc++] = (uint16_t)j; s_seen[j >> 5] |= 1u << (j & 31); }
                    }
                }
            }
            s_ncand = nc;
        }
        __builtin_amdgcn_wave_barrier();
        const int nc = s_ncand;
        // ---- candidates in parallel
        uint32_t k1 = 0xffffffffu, k2 = 0xffffffffu;
        uint32_t qd[8];
        const uint32_t* QD = reinterpret_cast<const uint32_t*>(A.qdesc) + (size_t)qi * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) qd[k] = QD[k];
        for (int ci = lane; ci < nc; ci += 64) {
            const int i2 = s_cand[ci];
            if (s_blocked[i2]) continue;
            const PslKeyLine kl = A.kls[i2];
            if (MODE == 0) {
                const double vc0 = (double)PSL_FSUB(kl.ePointInOctaveX, kl.sPointInOctaveX), vc1 = (double)PSL_FSUB(kl.ePointInOctaveY, kl.sPointInOctaveY);
                const double vl0 = (double)q.vx, vl1 = (double)q.vy;
                const double dot = PSL_DADD(PSL_DMUL(vc0, vl0), PSL_DMUL(vc1, vl1));
                const double den = PSL_DMUL(__dsqrt_rn(PSL_DADD(PSL_DMUL(vc0, vc0), PSL_DMUL(vc1, vc1))), __dsqrt_rn(PSL_DADD(PSL_DMUL(vl0, vl0), PSL_DMUL(vl1, vl1))));
                if (fabs(dot / den) < A.cos_gate) continue;
                const float mx = fmaxf(q.length, kl.lineLength), mn = fminf(q.length, kl.lineLength);
                if ((double)PSL_FDIV(mn, mx) < 0.75) continue;
            } else {
                const double* f = A.dir3d + 3 * (size_t)i2;
                const float dot = (float)PSL_DADD(PSL_DADD(PSL_DMUL(f[0], q.wdir[0]), PSL_DMUL(f[1], q.wdir[1])), PSL_DMUL(f[2], q.wdir[2]));
                const float mag_f = (float)__dsqrt_rn(PSL_DADD(PSL_DADD(PSL_DMUL(f[0], f[0]), PSL_DMUL(f[1], f[1])), PSL_DMUL(f[2], f[2])));
                const float mag_ml = (float)__dsqrt_rn(PSL_DADD(PSL_DADD(PSL_DMUL(q.wdir[0], q.wdir[0]), PSL_DMUL(q.wdir[1], q.wdir[1])), PSL_DMUL(q.wdir[2], q.wdir[2])));
                const float angle = __builtin_fabsf(PSL_FDIV(dot, PSL_FMUL(mag_f, mag_ml)));
                if ((double)angle < A.cos_gate) continue;
            }
            const uint32_t* D = reinterpret_cast<const uint32_t*>(A.desc) + (size_t)i2 * 8;
            int d = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) d += __popc(qd[k] ^ D[k]);
            const uint32_t key = ((uint32_t)d << 16) | (uint32_t)ci;
            if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t o1 = __shfl_xor(k1, o), o2 = __shfl_xor(k2, o);
            const uint32_t lo = min(k1, o1), hi = max(k1, o1);
            k2 = min(hi, min(k2, o2));
            k1 = lo;
        }
        int pick = -1;
        if (k1 != 0xffffffffu && (int)(k1 >> 16) <= PSL_LINE_TH) {
            pick = s_cand[k1 & 0xffff];
            if (MODE == 1 && k2 != 0xffffffffu) {
                const int l1 = A.kls[pick].octave, l2 = A.kls[s_cand[k2 & 0xffff]].octave;
                if (l1 == l2 && (float)(k1 >> 16) > PSL_FMUL(A.nnratio, (float)(k2 >> 16))) pick = -1;
            }
        }
        if (lane == 0) {
            A.match[qi] = pick;
            if (pick >= 0) { s_owner[pick] = qi; s_blocked[pick] = q.blocks != 0; }
        }
        nmatches += pick >= 0;
        __builtin_amdgcn_wave_barrier();
    }
    if (A.assigned) for (int i = lane; i < n; i += 64) A.assigned[i] = s_owner[i];
    if (lane == 0) *A.nmatches = nmatches;
}

namespace {
struct DevBufs {  // device buffers of the host-pointer entry point: carved from the context's scratch arena
    pslfe_ctx* ctx;
    explicit DevBufs(pslfe_ctx* c) : ctx(c) {}
    template <typename T>
    T* up(const T* host, size_t count, hipStream_t st, hipError_t* e) { return psl_scratch_up(ctx, host, count, st, e); }
};
}  // namespace

extern "C" {

int pslfe_line_search_by_projection(pslfe_ctx* ctx, const PslKeyLine* kls, const uint8_t* desc, const double* lineEq, const double* dir3d, int n,
                                    float min_x, float min_y, float max_x, float max_y, const PslLineQuery* queries, const uint8_t* qdesc,
                                    int nq, const uint8_t* taken, int mode, float nnratio, int32_t* match, int32_t* assigned, int* nmatches,
                                    int32_t* grid_start, int32_t* grid_idx, int grid_cap, int* grid_n) {
    PSL_REQUIRE(ctx && nmatches && (nq == 0 || (queries && qdesc && match)) && (n == 0 || (kls && desc && lineEq)), PSLFE_E_INVALID,
                "pslfe_line_search_by_projection: NULL argument");
    PSL_REQUIRE(mode == 0 || (mode == 1 && (n == 0 || dir3d)), PSLFE_E_INVALID, "pslfe_line_search_by_projection: mode %d", mode);
    PSL_REQUIRE(n >= 0 && n <= PSL_LINE_MAX && nq >= 0, PSLFE_E_CAPACITY, "pslfe_line_search_by_projection: %d lines (max %d)", n, PSL_LINE_MAX);
    PSL_REQUIRE(max_x > min_x && max_y > min_y, PSLFE_E_INVALID, "pslfe_line_search_by_projection: empty image bounds");
    *nmatches = 0;
    for (int i = 0; i < nq; ++i) match[i] = -1;
    if (assigned) for (int i = 0; i < n; ++i) assigned[i] = -1;
    if (n == 0) return PSLFE_OK;
    PSL_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    { const int rc_ = psl_scratch_begin(ctx); if (rc_) return rc_; }
    DevBufs B(ctx);
    hipError_t e = hipSuccess;
    LineMatchArgs A;
    memset(&A, 0, sizeof(A));
    A.kls = B.up(kls, n, st, &e);
    A.desc = B.up(desc, (size_t)n * 32, st, &e);
    A.eq = B.up(lineEq, (size_t)n * 3, st, &e);
    A.dir3d = dir3d ? B.up(dir3d, (size_t)n * 3, st, &e) : nullptr;
    A.n = n;
    A.minX = min_x; A.minY = min_y;
    A.invW = (float)PSL_LG_COLS / (float)(max_x - min_x);
    A.invH = (float)PSL_LG_ROWS / (float)(max_y - min_y);
    A.q = B.up(queries, nq, st, &e);
    A.qdesc = B.up(qdesc, (size_t)nq * 32, st, &e);
    A.nq = nq;
    A.taken = taken ? B.up(taken, n, st, &e) : nullptr;
    A.nnratio = nnratio;
    A.cos_gate = mode == 0 ? cos(10.0 / 180.0 * M_PI) : cos(15.0 / 180.0 * M_PI);
    A.gcap = n * (PSL_LG_COLS + PSL_LG_ROWS);  // a Bresenham walk visits at most max(cols, rows) + 1 cells
    A.gstart = B.up((const int*)nullptr, PSL_LG_CELLS + 1, st, &e);
    A.gidx = B.up((const int*)nullptr, A.gcap, st, &e);
    A.match = B.up((const int*)nullptr, nq ? nq : 1, st, &e);
    A.assigned = B.up((const int*)nullptr, n, st, &e);
    A.nmatches = B.up((const int*)nullptr, 1, st, &e);
    PSL_REQUIRE(e == hipSuccess, PSLFE_E_HIP, "pslfe_line_search_by_projection: %s", hipGetErrorString(e));
    {
        PSL_STAGE_BEGIN(ctx, "line.proj_match");
        if (mode == 0) k_line_proj_match<0><<<1, 64, 0, st>>>(A); else k_line_proj_match<1><<<1, 64, 0, st>>>(A);
        PSL_STAGE_END(ctx, "line.proj_match");
    }
    PSL_HIP(hipGetLastError());
    if (nq) PSL_HIP(hipMemcpyAsync(match, A.match, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, st));
    if (assigned) PSL_HIP(hipMemcpyAsync(assigned, A.assigned, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipMemcpyAsync(nmatches, A.nmatches, sizeof(int), hipMemcpyDeviceToHost, st));
    if (grid_start) PSL_HIP(hipMemcpyAsync(grid_start, A.gstart, (PSL_LG_CELLS + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    if (grid_start && grid_n) *grid_n = grid_start[PSL_LG_CELLS];
    if (grid_start && grid_idx) {
        const int m = std::min(std::min(grid_start[PSL_LG_CELLS], grid_cap), A.gcap);
        if (m > 0) PSL_HIP(hipMemcpy(grid_idx, A.gidx, (size_t)m * sizeof(int), hipMemcpyDeviceToHost));
    }
    return PSLFE_OK;
}

}  // extern "C"
