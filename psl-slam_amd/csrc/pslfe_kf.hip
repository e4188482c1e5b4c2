// libpslfe: the KeyFrame-rate matchers of LocalMapping / LoopClosing. Product code.
// Reference behaviour reproduced, from the point where the host has projected its map points / lines:
//   ORBmatcher::Fuse (both)                     src/ORBmatcher.cc:825-966, 968-1100
//   ORBmatcher::SearchBySim3                    src/ORBmatcher.cc:1102-1326
//   ORBmatcher::SearchForTriangulation          src/ORBmatcher.cc:657-823 (+ CheckDistEpipolarLine :140-157)
//   KeyFrame::GetFeaturesInArea / GetLinesInArea src/KeyFrame.cc:685-724, 857-891
//   LSDmatcher::Fuse                            add_src/LSDmatcher.cpp:847-984
//   MapPoint / MapLine::ComputeDistinctiveDescriptors  src/MapPoint.cc:242-304, add_src/MapLine.cpp:250-310
//
// None of these has the first-come-first-served coupling of the Tracking matchers (vbMatched2 of SearchForTriangulation
// is never written, src/ORBmatcher.cc:686), so every query is one wave: lanes take the candidates, a key
// (distance << 16 | visiting order) is min-reduced across the wave, and the reference's tie rule is the key's low half.
#include <limits.h>
#include <string.h>

#include "match_kernels.h"

#define PSL_TH_LOW 50         // ORBmatcher::TH_LOW src/ORBmatcher.cc:38, LSDmatcher::TH_LOW
#define PSL_KF_LEVELS 16
#define PSL_DISTINCT_MAX 1024  // observations of one map point handled (36 KB of descriptors in LDS)

__device__ __forceinline__ uint32_t psl_wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o));
    return v;
}

__device__ __forceinline__ int psl_hamming_regs(const uint32_t (&qd)[8], const uint32_t* __restrict__ d) {
    const uint4 d0 = *reinterpret_cast<const uint4*>(d);
    const uint4 d1 = *reinterpret_cast<const uint4*>(d + 4);
    return __popc(qd[0] ^ d0.x) + __popc(qd[1] ^ d0.y) + __popc(qd[2] ^ d0.z) + __popc(qd[3] ^ d0.w) + __popc(qd[4] ^ d1.x) +
           __popc(qd[5] ^ d1.y) + __popc(qd[6] ^ d1.z) + __popc(qd[7] ^ d1.w);
}

struct BestArgs {
    FrameStore S;
    int slot;
    const PslProjQuery* q;
    const uint8_t* qdesc;
    int nq, chi2;
    float inv_sigma2[PSL_KF_LEVELS];
    int* best_idx;
    int* best_dist;
};

// Fuse / SearchBySim3 candidate loop: one wave per projected map point.
__global__ __launch_bounds__(256) void k_window_best(BestArgs A) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (qi >= A.nq) return;
    const PslProjQuery q = A.q[qi];
    if (!(q.radius >= 0)) {
        if (lane == 0) { A.best_idx[qi] = -1; A.best_dist[qi] = INT_MAX; }
        return;
    }
    const FrameView V = psl_frame_view(A.S, A.slot);
    const uint32_t* QD = reinterpret_cast<const uint32_t*>(A.qdesc + (size_t)qi * 32);
    uint32_t qd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qd[k] = QD[k];
    const WindowCols W = psl_window_cols(V, q, nullptr);
    const int lvl = q.max_level;
    const float r = q.radius;
    uint32_t best = PSL_KEY_INF;
    for (int base = 0; base < W.T; base += 64) {
        const int p = psl_window_pos(W, base + lane);
        if (p < 0) continue;  // (shuffles are done: the rest is lane-local)
        const int i2 = V.gidx[p];
        if (i2 < 0 || i2 >= V.n) continue;
        const float2 xy = *reinterpret_cast<const float2*>(&V.kps[i2].x);
        const int octave = V.kps[i2].octave;
        bool ok = __builtin_fabsf(PSL_FSUB(xy.x, q.u)) < r && __builtin_fabsf(PSL_FSUB(xy.y, q.v)) < r;
        ok = ok && !(octave < lvl - 1 || octave > lvl);
        if (ok && A.chi2) {  // src/ORBmatcher.cc:907-934
            const float kr = V.uright[i2];
            const float ex = PSL_FSUB(q.u, xy.x), ey = PSL_FSUB(q.v, xy.y);
            float e2 = PSL_FADD(PSL_FMUL(ex, ex), PSL_FMUL(ey, ey));
            const float is2 = A.inv_sigma2[octave & (PSL_KF_LEVELS - 1)];
            if (kr >= 0) {
                const float er = PSL_FSUB(q.ur, kr);
                e2 = PSL_FADD(e2, PSL_FMUL(er, er));
                ok = !((double)PSL_FMUL(e2, is2) > 7.8);
            } else {
                ok = !((double)PSL_FMUL(e2, is2) > 5.99);
            }
        }
        if (!ok) continue;
        const int dist = psl_hamming_regs(qd, V.desc + (size_t)i2 * 8);
        best = min(best, ((uint32_t)dist << 16) | (uint32_t)p);
    }
    best = psl_wave_min_u32(best);
    if (lane == 0) {
        A.best_idx[qi] = best == PSL_KEY_INF ? -1 : V.gidx[best & 0xffffu];
        A.best_dist[qi] = best == PSL_KEY_INF ? INT_MAX : (int)(best >> 16);
    }
}

// SearchBySim3 agreement check (src/ORBmatcher.cc:1225-1232, 1296-1323)
__global__ __launch_bounds__(256) void k_sim3_agree(const int* __restrict__ b1, const int* __restrict__ d1, int n1, const int* __restrict__ b2,
                                                     const int* __restrict__ d2, int n2, int* __restrict__ match12, int* __restrict__ nfound) {
    const int i1 = blockIdx.x * 256 + threadIdx.x;
    bool found = false;
    if (i1 < n1) {
        const int idx2 = d1[i1] <= PSL_TH_HIGH ? b1[i1] : -1;
        int m = -1;
        if (idx2 >= 0 && idx2 < n2) {
            const int idx1 = d2[idx2] <= PSL_TH_HIGH ? b2[idx2] : -1;
            if (idx1 == i1) m = idx2;
        }
        match12[i1] = m;
        found = m >= 0;
    }
    const int c = __popcll(__ballot(found));
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(nfound, c);
}

struct TriArgs {
    FrameStore S;
    int slot;
    const int* fidx;
    int nfidx;
    const uint8_t* taken;
    const PslTriQuery* q;
    const uint8_t* qdesc;
    int nq;
    float F[9];
    float ex, ey;
    int only_stereo, check_ori;
    float sf[PSL_KF_LEVELS], sig2[PSL_KF_LEVELS];
    int* choice;
    int* match;
    int* nmatches;
};

// SearchForTriangulation candidate loop (src/ORBmatcher.cc:713-756): one wave per feature of KF1.
__global__ __launch_bounds__(256) void k_triangulation(TriArgs A) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (qi >= A.nq) return;
    const PslTriQuery q = A.q[qi];
    const FrameView V = psl_frame_view(A.S, A.slot);
    const uint32_t* QD = reinterpret_cast<const uint32_t*>(A.qdesc + (size_t)qi * 32);
    uint32_t qd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qd[k] = QD[k];
    // epipolar line in the second image l = x1' F12 = [a b c]
    const float a = PSL_FADD(PSL_FADD(PSL_FMUL(q.x, A.F[0]), PSL_FMUL(q.y, A.F[3])), A.F[6]);
    const float b = PSL_FADD(PSL_FADD(PSL_FMUL(q.x, A.F[1]), PSL_FMUL(q.y, A.F[4])), A.F[7]);
    const float c = PSL_FADD(PSL_FADD(PSL_FMUL(q.x, A.F[2]), PSL_FMUL(q.y, A.F[5])), A.F[8]);
    const float den = PSL_FADD(PSL_FMUL(a, a), PSL_FMUL(b, b));
    const int start = max(q.start, 0), len = min(q.len, A.nfidx - start);
    uint32_t best = PSL_KEY_INF;
    for (int j = lane; j < len; j += 64) {
        const int idx2 = A.fidx[start + j];
        if (idx2 < 0 || idx2 >= V.n) continue;
        if (A.taken[idx2]) continue;
        const bool stereo2 = V.uright[idx2] >= 0;
        if (A.only_stereo && !stereo2) continue;
        const int dist = psl_hamming_regs(qd, V.desc + (size_t)idx2 * 8);
        if (dist > PSL_TH_LOW) continue;
        const float2 xy = *reinterpret_cast<const float2*>(&V.kps[idx2].x);
        const int octave = V.kps[idx2].octave & (PSL_KF_LEVELS - 1);
        if (!q.stereo && !stereo2) {
            const float dx = PSL_FSUB(A.ex, xy.x), dy = PSL_FSUB(A.ey, xy.y);
            if (PSL_FADD(PSL_FMUL(dx, dx), PSL_FMUL(dy, dy)) < PSL_FMUL(100.0f, A.sf[octave])) continue;
        }
        const float num = PSL_FADD(PSL_FADD(PSL_FMUL(a, xy.x), PSL_FMUL(b, xy.y)), c);
        if (den == 0) continue;
        const float dsqr = PSL_FDIV(PSL_FMUL(num, num), den);
        if (!((double)dsqr < PSL_DMUL(3.84, (double)A.sig2[octave]))) continue;
        // `dist > bestDist -> continue`: among equal distances the LAST visited candidate wins
        best = min(best, ((uint32_t)dist << 16) | (uint32_t)(0xffff - j));
    }
    best = psl_wave_min_u32(best);
    if (lane == 0) A.choice[qi] = best == PSL_KEY_INF ? -1 : A.fidx[start + (0xffff - (int)(best & 0xffffu))];
}

// rotation histogram + ComputeThreeMaxima + outputs (src/ORBmatcher.cc:764-811)
__global__ __launch_bounds__(1024) void k_triangulation_finish(TriArgs A) {
    __shared__ int s_hist[PSL_HISTO];
    __shared__ int s_ind[3];
    __shared__ int s_nm;
    __shared__ uint8_t s_bin[PSL_QMAX];
    const int tid = threadIdx.x;
    const FrameView V = psl_frame_view(A.S, A.slot);
    if (tid < PSL_HISTO) s_hist[tid] = 0;
    if (tid == 0) { s_ind[0] = s_ind[1] = s_ind[2] = -1; s_nm = 0; }
    __syncthreads();
    if (A.check_ori) {
        const float factor = 1.0f / PSL_HISTO;
        for (int qi = tid; qi < A.nq; qi += 1024) {
            const int c2 = A.choice[qi];
            if (c2 < 0) continue;
            float rot = PSL_FSUB(A.q[qi].angle, V.kps[c2].angle);
            if (rot < 0.0f) rot = PSL_FADD(rot, 360.0f);
            int bin = (int)__builtin_roundf(PSL_FMUL(rot, factor));
            if (bin == PSL_HISTO) bin = 0;
            bin = bin < 0 ? 0 : (bin >= PSL_HISTO ? PSL_HISTO - 1 : bin);
            s_bin[qi] = (uint8_t)bin;
            atomicAdd(&s_hist[bin], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < PSL_HISTO; ++i) {
                const int sz = s_hist[i];
                if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
                else if (sz > max3) { max3 = sz; ind3 = i; }
            }
            if ((float)max2 < PSL_FMUL(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < PSL_FMUL(0.1f, (float)max1)) { ind3 = -1; }
            s_ind[0] = ind1; s_ind[1] = ind2; s_ind[2] = ind3;
        }
        __syncthreads();
    }
    int local = 0;
    for (int qi = tid; qi < A.nq; qi += 1024) {
        const int c2 = A.choice[qi];
        bool good = c2 >= 0;
        if (good && A.check_ori) { const int bn = s_bin[qi]; good = (bn == s_ind[0] || bn == s_ind[1] || bn == s_ind[2]); }
        A.match[qi] = good ? c2 : -1;
        local += good;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((tid & 63) == 0 && local) atomicAdd(&s_nm, local);
    __syncthreads();
    if (tid == 0) *A.nmatches = s_nm;
}

// LSDmatcher::Fuse search: KeyFrame::GetLinesInArea is a linear scan of the keyframe's keylines; one wave per map line.
__global__ __launch_bounds__(256) void k_line_fuse_best(const PslKeyLine* __restrict__ kls, int n, const uint8_t* __restrict__ desc, int ndesc,
                                                         const PslLineFuseQuery* __restrict__ Q, const uint8_t* __restrict__ qdesc, int nq,
                                                         int* __restrict__ best_idx, int* __restrict__ best_dist) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (qi >= nq) return;
    const PslLineFuseQuery q = Q[qi];
    if (!(q.radius >= 0)) {
        if (lane == 0) { best_idx[qi] = -1; best_dist[qi] = 256; }
        return;
    }
    const uint32_t* QD = reinterpret_cast<const uint32_t*>(qdesc + (size_t)qi * 32);
    uint32_t qd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qd[k] = QD[k];
    float d1x = PSL_FSUB(q.x1, q.x2), d1y = PSL_FSUB(q.y1, q.y2);
    const float n1 = sqrtf(PSL_FADD(PSL_FMUL(d1x, d1x), PSL_FMUL(d1y, d1y)));
    d1x = PSL_FDIV(d1x, n1);
    d1y = PSL_FDIV(d1y, n1);
    const double mx = PSL_DMUL(0.5, (double)PSL_FADD(q.x1, q.x2)), my = PSL_DMUL(0.5, (double)PSL_FADD(q.y1, q.y2));
    const float rr = PSL_FMUL(q.radius, q.radius);
    uint32_t best = PSL_KEY_INF;
    for (int k = lane; k < n; k += 64) {
        const PslKeyLine kl = kls[k];
        const double ddx = PSL_DSUB(mx, (double)kl.pt_x), ddy = PSL_DSUB(my, (double)kl.pt_y);
        const float distance = (float)PSL_DADD(PSL_DMUL(ddx, ddx), PSL_DMUL(ddy, ddy));
        if (distance > rr) continue;
        float d2x = PSL_FSUB(kl.startPointX, kl.endPointX), d2y = PSL_FSUB(kl.startPointY, kl.endPointY);
        const float n2 = sqrtf(PSL_FADD(PSL_FMUL(d2x, d2x), PSL_FMUL(d2y, d2y)));
        d2x = PSL_FDIV(d2x, n2);
        d2y = PSL_FDIV(d2y, n2);
        const float cs = __builtin_fabsf(PSL_FADD(PSL_FMUL(d1x, d2x), PSL_FMUL(d1y, d2y)));
        if (cs < 0.998f) continue;
        if (kl.octave < q.level - 1 || kl.octave > q.level) continue;
        if (k >= ndesc) continue;
        const int dist = psl_hamming_regs(qd, reinterpret_cast<const uint32_t*>(desc) + (size_t)k * 8);
        best = min(best, ((uint32_t)dist << 16) | (uint32_t)k);
    }
    best = psl_wave_min_u32(best);
    if (lane == 0) {
        best_idx[qi] = best == PSL_KEY_INF ? -1 : (int)(best & 0xffffu);
        best_dist[qi] = best == PSL_KEY_INF ? 256 : (int)(best >> 16);
    }
}

// ComputeDistinctiveDescriptors: one wave per map point.  Row i of the distance matrix lives in the lanes' registers
// (lane l holds columns l, l + 64, ...); its median sorted[k], k = floor(0.5 (N - 1)), is the smallest value v with
// #{d <= v} >= k + 1, found by bisection on v in [0, 256] with ballot-free popcount sums.
__global__ __launch_bounds__(64) void k_distinctive(const uint8_t* __restrict__ desc, const int* __restrict__ offsets, int npts, int* __restrict__ best) {
    __shared__ uint32_t s_d[PSL_DISTINCT_MAX * 9];  // rows padded to 9 words: conflict-free column reads
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= npts) return;
    const int o0 = offsets[p];
    int N = offsets[p + 1] - o0;
    if (N <= 0) {
        if (lane == 0) best[p] = -1;
        return;
    }
    N = min(N, PSL_DISTINCT_MAX);
    const uint32_t* D = reinterpret_cast<const uint32_t*>(desc) + (size_t)o0 * 8;
    for (int i = lane; i < N * 8; i += 64) s_d[(i >> 3) * 9 + (i & 7)] = D[i];
    __syncthreads();
    const int k = (int)(0.5 * (double)(N - 1));
    constexpr int PER = PSL_DISTINCT_MAX / 64;
    uint32_t bestKey = PSL_KEY_INF;  // median << 16 | row
    for (int i = 0; i < N; ++i) {
        uint32_t qd[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) qd[w] = s_d[i * 9 + w];
        int d[PER];
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            const int j = lane + 64 * t;
            d[t] = 0x7fff;
            if (t * 64 < N && j < N) {
                int s = 0;
#pragma unroll
                for (int w = 0; w < 8; ++w) s += __popc(qd[w] ^ s_d[j * 9 + w]);
                d[t] = s;
            }
        }
        int lo = 0, hi = 256;  // smallest v with count(d <= v) >= k + 1
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
#pragma unroll
            for (int t = 0; t < PER; ++t)
                if (t * 64 < N) cnt += d[t] <= mid;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
            if (cnt >= k + 1) hi = mid; else lo = mid + 1;
        }
        bestKey = min(bestKey, ((uint32_t)lo << 16) | (uint32_t)i);
    }
    if (lane == 0) best[p] = (int)(bestKey & 0xffffu);
}

// ---------------------------------------------------------------------------------------------
struct pslfe_kf {
    pslfe_ctx* ctx = nullptr;
    char* arena = nullptr;
    size_t arena_cap = 0, arena_top = 0;

    int reserve(size_t bytes) {
        arena_top = 0;
        if (bytes <= arena_cap) return PSLFE_OK;
        PSL_HIP(hipStreamSynchronize(ctx->stream));
        if (arena) PSL_HIP(hipFree(arena));
        arena = nullptr;
        arena_cap = 0;
        const size_t want = psl_align_up(bytes + bytes / 2, 1 << 16);
        PSL_HIP(hipMalloc(&arena, want));
        arena_cap = want;
        return PSLFE_OK;
    }
    template <typename T>
    T* take(size_t count) {
        T* p = reinterpret_cast<T*>(arena + arena_top);
        arena_top += psl_align_up(count * sizeof(T) + 1, 256);
        return p;
    }
};

namespace {
size_t padded(size_t bytes) { return psl_align_up(bytes + 1, 256); }

int check_slot(pslfe_frame* f, int slot, const char* who) {
    PSL_REQUIRE(f, PSLFE_E_INVALID, "%s: NULL frame", who);
    PSL_REQUIRE(slot >= 0 && slot < f->max_frames && f->slot_set[slot], PSLFE_E_STATE, "%s: slot %d not set", who, slot);
    return PSLFE_OK;
}

int launch_window_best(pslfe_kf* k, pslfe_frame* f, int slot, const PslProjQuery* d_q, const uint8_t* d_qdesc, int nq, int chi2,
                       const float* inv_sigma2, int nlevels, int* d_idx, int* d_dist) {
    BestArgs A;
    A.S = f->S; A.slot = slot; A.q = d_q; A.qdesc = d_qdesc; A.nq = nq; A.chi2 = chi2;
    for (int i = 0; i < PSL_KF_LEVELS; ++i) A.inv_sigma2[i] = (inv_sigma2 && i < nlevels) ? inv_sigma2[i] : 0.f;
    A.best_idx = d_idx; A.best_dist = d_dist;
    k_window_best<<<(nq + 3) / 4, 256, 0, k->ctx->stream>>>(A);
    PSL_HIP(hipGetLastError());
    return PSLFE_OK;
}
}  // namespace

extern "C" {

int pslfe_kf_create(pslfe_ctx* ctx, pslfe_kf** out) {
    PSL_REQUIRE(ctx && out, PSLFE_E_INVALID, "pslfe_kf_create: NULL argument");
    pslfe_kf* k = new pslfe_kf();
    k->ctx = ctx;
    *out = k;
    return PSLFE_OK;
}

void pslfe_kf_destroy(pslfe_kf* k) {
    if (!k) return;
    if (k->arena) {
        (void)hipSetDevice(k->ctx->device);
        (void)hipStreamSynchronize(k->ctx->stream);
        (void)hipFree(k->arena);
    }
    delete k;
}

int pslfe_kf_window_best(pslfe_kf* k, pslfe_frame* f, int slot, const PslProjQuery* queries, const uint8_t* qdesc, int nq, int chi2,
                         const float* inv_level_sigma2, int nlevels, int32_t* best_idx, int32_t* best_dist) {
    PSL_REQUIRE(k && (nq == 0 || (queries && qdesc && best_idx && best_dist)), PSLFE_E_INVALID, "pslfe_kf_window_best: NULL argument");
    PSL_REQUIRE(nq >= 0, PSLFE_E_INVALID, "pslfe_kf_window_best: nq = %d", nq);
    PSL_REQUIRE(!chi2 || (inv_level_sigma2 && nlevels > 0 && nlevels <= PSL_KF_LEVELS), PSLFE_E_INVALID,
                "pslfe_kf_window_best: chi2 gates need mvInvLevelSigma2 with 1..%d levels", PSL_KF_LEVELS);
    if (int rc = check_slot(f, slot, "pslfe_kf_window_best")) return rc;
    if (nq == 0) return PSLFE_OK;
    PSL_HIP(hipSetDevice(k->ctx->device));
    hipStream_t st = k->ctx->stream;
    if (int rc = k->reserve(padded((size_t)nq * sizeof(PslProjQuery)) + padded((size_t)nq * 32) + 2 * padded((size_t)nq * 4))) return rc;
    PslProjQuery* d_q = k->take<PslProjQuery>(nq);
    uint8_t* d_qd = k->take<uint8_t>((size_t)nq * 32);
    int* d_i = k->take<int>(nq);
    int* d_d = k->take<int>(nq);
    PSL_HIP(hipMemcpyAsync(d_q, queries, (size_t)nq * sizeof(PslProjQuery), hipMemcpyHostToDevice, st));
    PSL_HIP(hipMemcpyAsync(d_qd, qdesc, (size_t)nq * 32, hipMemcpyHostToDevice, st));
    {
        PSL_STAGE_BEGIN(k->ctx, "kf.window_best");
        if (int rc = launch_window_best(k, f, slot, d_q, d_qd, nq, chi2, inv_level_sigma2, nlevels, d_i, d_d)) return rc;
        PSL_STAGE_END(k->ctx, "kf.window_best");
    }
    PSL_HIP(hipMemcpyAsync(best_idx, d_i, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    PSL_HIP(hipMemcpyAsync(best_dist, d_d, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    return PSLFE_OK;
}

int pslfe_kf_search_by_sim3(pslfe_kf* k, pslfe_frame* f1, int slot1, pslfe_frame* f2, int slot2, const PslProjQuery* q12,
                            const uint8_t* qdesc1, int n1, const PslProjQuery* q21, const uint8_t* qdesc2, int n2, int32_t* match12,
                            int* nfound) {
    PSL_REQUIRE(k && nfound && (n1 == 0 || (q12 && qdesc1 && match12)) && (n2 == 0 || (q21 && qdesc2)), PSLFE_E_INVALID,
                "pslfe_kf_search_by_sim3: NULL argument");
    PSL_REQUIRE(n1 >= 0 && n2 >= 0, PSLFE_E_INVALID, "pslfe_kf_search_by_sim3: negative count");
    if (int rc = check_slot(f1, slot1, "pslfe_kf_search_by_sim3")) return rc;
    if (int rc = check_slot(f2, slot2, "pslfe_kf_search_by_sim3")) return rc;
    *nfound = 0;
    if (n1 == 0) return PSLFE_OK;
    PSL_HIP(hipSetDevice(k->ctx->device));
    hipStream_t st = k->ctx->stream;
    const size_t m1 = (size_t)n1, m2 = (size_t)(n2 > 0 ? n2 : 1);
    if (int rc = k->reserve(padded(m1 * sizeof(PslProjQuery)) + padded(m1 * 32) + padded(m2 * sizeof(PslProjQuery)) + padded(m2 * 32) +
                            3 * padded(m1 * 4) + 2 * padded(m2 * 4) + padded(4)))
        return rc;
    PslProjQuery* d_q1 = k->take<PslProjQuery>(m1);
    uint8_t* d_qd1 = k->take<uint8_t>(m1 * 32);
    PslProjQuery* d_q2 = k->take<PslProjQuery>(m2);
    uint8_t* d_qd2 = k->take<uint8_t>(m2 * 32);
    int* d_b1 = k->take<int>(m1);
    int* d_d1 = k->take<int>(m1);
    int* d_m = k->take<int>(m1);
    int* d_b2 = k->take<int>(m2);
    int* d_d2 = k->take<int>(m2);
    int* d_nf = k->take<int>(1);
    PSL_HIP(hipMemcpyAsync(d_q1, q12, m1 * sizeof(PslProjQuery), hipMemcpyHostToDevice, st));
    PSL_HIP(hipMemcpyAsync(d_qd1, qdesc1, m1 * 32, hipMemcpyHostToDevice, st));
    if (n2 > 0) {
        PSL_HIP(hipMemcpyAsync(d_q2, q21, m2 * sizeof(PslProjQuery), hipMemcpyHostToDevice, st));
        PSL_HIP(hipMemcpyAsync(d_qd2, qdesc2, m2 * 32, hipMemcpyHostToDevice, st));
    }
    PSL_HIP(hipMemsetAsync(d_nf, 0, 4, st));
    {
        PSL_STAGE_BEGIN(k->ctx, "kf.sim3");
        if (int rc = launch_window_best(k, f2, slot2, d_q1, d_qd1, n1, 0, nullptr, 0, d_b1, d_d1)) return rc;
        if (n2 > 0)
            if (int rc = launch_window_best(k, f1, slot1, d_q2, d_qd2, n2, 0, nullptr, 0, d_b2, d_d2)) return rc;
        k_sim3_agree<<<(n1 + 255) / 256, 256, 0, st>>>(d_b1, d_d1, n1, d_b2, d_d2, n2, d_m, d_nf);
        PSL_STAGE_END(k->ctx, "kf.sim3");
    }
    PSL_HIP(hipGetLastError());
    PSL_HIP(hipMemcpyAsync(match12, d_m, m1 * 4, hipMemcpyDeviceToHost, st));
    PSL_HIP(hipMemcpyAsync(nfound, d_nf, 4, hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    return PSLFE_OK;
}

int pslfe_kf_search_for_triangulation(pslfe_kf* k, pslfe_frame* f2, int slot2, const int32_t* fidx2, int nfidx2, const uint8_t* taken2,
                                      const PslTriQuery* queries, const uint8_t* qdesc, int nq, const float* F12, float ex, float ey,
                                      int only_stereo, int check_orientation, const float* scale_factors, const float* level_sigma2,
                                      int nlevels, int32_t* match, int* nmatches) {
    PSL_REQUIRE(k && nmatches && F12 && scale_factors && level_sigma2 && (nq == 0 || (queries && qdesc && match)) &&
                    (nfidx2 == 0 || (fidx2 && taken2)),
                PSLFE_E_INVALID, "pslfe_kf_search_for_triangulation: NULL argument");
    PSL_REQUIRE(nq >= 0 && nq <= PSL_QMAX, PSLFE_E_INVALID, "pslfe_kf_search_for_triangulation: %d queries (max %d)", nq, PSL_QMAX);
    PSL_REQUIRE(nlevels > 0 && nlevels <= PSL_KF_LEVELS, PSLFE_E_INVALID, "pslfe_kf_search_for_triangulation: %d levels (max %d)", nlevels,
                PSL_KF_LEVELS);
    PSL_REQUIRE(nfidx2 >= 0 && nfidx2 <= 0xffff, PSLFE_E_CAPACITY, "pslfe_kf_search_for_triangulation: %d feature-vector entries", nfidx2);
    if (int rc = check_slot(f2, slot2, "pslfe_kf_search_for_triangulation")) return rc;
    *nmatches = 0;
    if (nq == 0) return PSLFE_OK;
    PSL_HIP(hipSetDevice(k->ctx->device));
    hipStream_t st = k->ctx->stream;
    FrameMeta m;
    PSL_HIP(hipMemcpyAsync(&m, f2->S.meta + slot2, sizeof(m), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    const size_t nf = (size_t)(nfidx2 > 0 ? nfidx2 : 1), nk = (size_t)(m.n > 0 ? m.n : 1);
    if (int rc = k->reserve(padded((size_t)nq * sizeof(PslTriQuery)) + padded((size_t)nq * 32) + padded(nf * 4) + padded(nk) +
                            2 * padded((size_t)nq * 4) + padded(4)))
        return rc;
    TriArgs A;
    PslTriQuery* d_q = k->take<PslTriQuery>(nq);
    uint8_t* d_qd = k->take<uint8_t>((size_t)nq * 32);
    int* d_fidx = k->take<int>(nf);
    uint8_t* d_taken = k->take<uint8_t>(nk);
    A.choice = k->take<int>(nq);
    A.match = k->take<int>(nq);
    A.nmatches = k->take<int>(1);
    PSL_HIP(hipMemcpyAsync(d_q, queries, (size_t)nq * sizeof(PslTriQuery), hipMemcpyHostToDevice, st));
    PSL_HIP(hipMemcpyAsync(d_qd, qdesc, (size_t)nq * 32, hipMemcpyHostToDevice, st));
    if (nfidx2 > 0) PSL_HIP(hipMemcpyAsync(d_fidx, fidx2, (size_t)nfidx2 * 4, hipMemcpyHostToDevice, st));
    if (m.n > 0 && taken2) PSL_HIP(hipMemcpyAsync(d_taken, taken2, (size_t)m.n, hipMemcpyHostToDevice, st));
    else PSL_HIP(hipMemsetAsync(d_taken, 0, nk, st));
    A.S = f2->S; A.slot = slot2; A.fidx = d_fidx; A.nfidx = nfidx2; A.taken = d_taken; A.q = d_q; A.qdesc = d_qd; A.nq = nq;
    memcpy(A.F, F12, sizeof(A.F));
    A.ex = ex; A.ey = ey; A.only_stereo = only_stereo; A.check_ori = check_orientation;
    for (int i = 0; i < PSL_KF_LEVELS; ++i) { A.sf[i] = i < nlevels ? scale_factors[i] : 0.f; A.sig2[i] = i < nlevels ? level_sigma2[i] : 0.f; }
    {
        PSL_STAGE_BEGIN(k->ctx, "kf.triangulation");
        k_triangulation<<<(nq + 3) / 4, 256, 0, st>>>(A);
        k_triangulation_finish<<<1, 1024, 0, st>>>(A);
        PSL_STAGE_END(k->ctx, "kf.triangulation");
    }
    PSL_HIP(hipGetLastError());
    PSL_HIP(hipMemcpyAsync(match, A.match, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    PSL_HIP(hipMemcpyAsync(nmatches, A.nmatches, 4, hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    return PSLFE_OK;
}

int pslfe_kf_line_fuse_best(pslfe_kf* k, const PslKeyLine* kls, int n, const uint8_t* desc, int ndesc, const PslLineFuseQuery* queries,
                            const uint8_t* qdesc, int nq, int32_t* best_idx, int32_t* best_dist) {
    PSL_REQUIRE(k && (nq == 0 || (queries && qdesc && best_idx && best_dist)) && (n == 0 || kls) && (ndesc == 0 || desc), PSLFE_E_INVALID,
                "pslfe_kf_line_fuse_best: NULL argument");
    PSL_REQUIRE(nq >= 0 && n >= 0 && n <= 0xffff && ndesc >= 0, PSLFE_E_INVALID, "pslfe_kf_line_fuse_best: bad count");
    if (nq == 0) return PSLFE_OK;
    PSL_HIP(hipSetDevice(k->ctx->device));
    hipStream_t st = k->ctx->stream;
    const size_t nn = (size_t)(n > 0 ? n : 1), nd = (size_t)(ndesc > 0 ? ndesc : 1);
    if (int rc = k->reserve(padded(nn * sizeof(PslKeyLine)) + padded(nd * 32) + padded((size_t)nq * sizeof(PslLineFuseQuery)) +
                            padded((size_t)nq * 32) + 2 * padded((size_t)nq * 4)))
        return rc;
    PslKeyLine* d_kl = k->take<PslKeyLine>(nn);
    uint8_t* d_desc = k->take<uint8_t>(nd * 32);
    PslLineFuseQuery* d_q = k->take<PslLineFuseQuery>(nq);
    uint8_t* d_qd = k->take<uint8_t>((size_t)nq * 32);
    int* d_i = k->take<int>(nq);
    int* d_d = k->take<int>(nq);
    if (n > 0) PSL_HIP(hipMemcpyAsync(d_kl, kls, (size_t)n * sizeof(PslKeyLine), hipMemcpyHostToDevice, st));
    if (ndesc > 0) PSL_HIP(hipMemcpyAsync(d_desc, desc, (size_t)ndesc * 32, hipMemcpyHostToDevice, st));
    PSL_HIP(hipMemcpyAsync(d_q, queries, (size_t)nq * sizeof(PslLineFuseQuery), hipMemcpyHostToDevice, st));
    PSL_HIP(hipMemcpyAsync(d_qd, qdesc, (size_t)nq * 32, hipMemcpyHostToDevice, st));
    {
        PSL_STAGE_BEGIN(k->ctx, "kf.line_fuse");
        k_line_fuse_best<<<(nq + 3) / 4, 256, 0, st>>>(d_kl, n, d_desc, ndesc, d_q, d_qd, nq, d_i, d_d);
        PSL_STAGE_END(k->ctx, "kf.line_fuse");
    }
    PSL_HIP(hipGetLastError());
    PSL_HIP(hipMemcpyAsync(best_idx, d_i, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    PSL_HIP(hipMemcpyAsync(best_dist, d_d, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    return PSLFE_OK;
}

int pslfe_kf_distinctive_descriptors(pslfe_kf* k, const uint8_t* desc, const int32_t* offsets, int npts, int32_t* best) {
    PSL_REQUIRE(k && (npts == 0 || (offsets && best)), PSLFE_E_INVALID, "pslfe_kf_distinctive_descriptors: NULL argument");
    PSL_REQUIRE(npts >= 0, PSLFE_E_INVALID, "pslfe_kf_distinctive_descriptors: npts = %d", npts);
    if (npts == 0) return PSLFE_OK;
    PSL_REQUIRE(offsets[0] == 0, PSLFE_E_INVALID, "pslfe_kf_distinctive_descriptors: offsets[0] must be 0");
    for (int p = 0; p < npts; ++p) {
        const int len = offsets[p + 1] - offsets[p];
        PSL_REQUIRE(len >= 0, PSLFE_E_INVALID, "pslfe_kf_distinctive_descriptors: offsets not ascending at %d", p);
        PSL_REQUIRE(len <= PSL_DISTINCT_MAX, PSLFE_E_CAPACITY, "pslfe_kf_distinctive_descriptors: point %d has %d observations (max %d)", p, len,
                    PSL_DISTINCT_MAX);
    }
    const size_t total = (size_t)offsets[npts];
    PSL_REQUIRE(total == 0 || desc, PSLFE_E_INVALID, "pslfe_kf_distinctive_descriptors: NULL descriptors");
    PSL_HIP(hipSetDevice(k->ctx->device));
    hipStream_t st = k->ctx->stream;
    if (int rc = k->reserve(padded((total ? total : 1) * 32) + padded((size_t)(npts + 1) * 4) + padded((size_t)npts * 4))) return rc;
    uint8_t* d_desc = k->take<uint8_t>((total ? total : 1) * 32);
    int* d_off = k->take<int>(npts + 1);
    int* d_best = k->take<int>(npts);
    if (total) PSL_HIP(hipMemcpyAsync(d_desc, desc, total * 32, hipMemcpyHostToDevice, st));
    PSL_HIP(hipMemcpyAsync(d_off, offsets, (size_t)(npts + 1) * 4, hipMemcpyHostToDevice, st));
    {
        PSL_STAGE_BEGIN(k->ctx, "kf.distinctive");
        k_distinctive<<<npts, 64, 0, st>>>(d_desc, d_off, npts, d_best);
        PSL_STAGE_END(k->ctx, "kf.distinctive");
    }
    PSL_HIP(hipGetLastError());
    PSL_HIP(hipMemcpyAsync(best, d_best, (size_t)npts * 4, hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    return PSLFE_OK;
}

}  // extern "C"
