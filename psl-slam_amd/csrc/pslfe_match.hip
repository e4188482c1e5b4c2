// libpslfe: descriptor matching on the 64x48 frame grid and brute-force Hamming kNN-2. Product code.
// Reference behaviour reproduced:
//   Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea   src/Frame.cc:269-284, 1040-1050, 985-1038
//   ORBmatcher::SearchByProjection(cur,last)                      src/ORBmatcher.cc:1328-1470
//   ORBmatcher::SearchByProjection(F,MPs)                         src/ORBmatcher.cc:45-129
//   ORBmatcher::ComputeThreeMaxima / DescriptorDistance           src/ORBmatcher.cc:1601-1663
//   BFMatcher(NORM_HAMMING).knnMatch(k=2) in LSDmatcher::matchNNR  add_src/LSDmatcher.cpp:354-376
//
// The reference matchers are sequential: a keypoint taken by a map point with observations is
// skipped by every LATER query (src/ORBmatcher.cc:1401-1403).  k_window_eval + k_window_resolve reproduce that
// exactly with a fixpoint: every query picks its best candidate among those not taken by an
// EARLIER query; "taken by" is recomputed from the current picks until nothing changes.  By
// induction over the query index the fixpoint is the sequential result.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "pslfe_internal.h"
#include "psl_device_math.h"

#include "match_kernels.h"

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_frame_import(FrameStore S, const PslKeyPoint* __restrict__ okps,
                                                       const uint8_t* __restrict__ odesc, const int* __restrict__ ocounts,
                                                       int ocap, float minX, float minY, float invW, float invH) {
    const int slot = blockIdx.x;
    int n = ocounts[slot];
    n = n < S.cap ? n : S.cap;
    const uint32_t* sk = reinterpret_cast<const uint32_t*>(okps + (size_t)slot * ocap);
    uint32_t* dk = reinterpret_cast<uint32_t*>(S.kps + (size_t)slot * S.cap);
    for (int i = threadIdx.x; i < n * 7; i += 256) dk[i] = sk[i];
    const uint32_t* sd = reinterpret_cast<const uint32_t*>(odesc + (size_t)slot * ocap * 32);
    uint32_t* dd = reinterpret_cast<uint32_t*>(S.desc + (size_t)slot * S.cap * 32);
    for (int i = threadIdx.x; i < n * 8; i += 256) dd[i] = sd[i];
    float* ur = S.uright + (size_t)slot * S.cap;
    for (int i = threadIdx.x; i < n; i += 256) ur[i] = -1.f;
    if (threadIdx.x == 0) {
        FrameMeta m;
        m.n = n; m.minX = minX; m.minY = minY; m.invW = invW; m.invH = invH;
        S.meta[slot] = m;
    }
}


// ---------------------------------------------------------------------------------------------
// RGB-D post-processing of a frame's keypoints: Frame::UndistortKeyPoints src/Frame.cc:1062-1092
// (cv::undistortPoints with P = K: five fixed iterations in double, OpenCV 3.2 Appendix A),
// Frame::ComputeStereoFromRGBD :1342-1363 and Frame::ComputeImageBounds :1135-1168.
__device__ __forceinline__ void psl_undistort_point(double u, double v, const PslCamera& C, float* ox, float* oy) {
    const double fx = C.fx, fy = C.fy, cx = C.cx, cy = C.cy, k1 = C.k1, k2 = C.k2, p1 = C.p1, p2 = C.p2, k3 = C.k3;
    const double ifx = __ddiv_rn(1., fx), ify = __ddiv_rn(1., fy);
    double x = __dmul_rn(__dsub_rn(u, cx), ifx), y = __dmul_rn(__dsub_rn(v, cy), ify);
    const double x0 = x, y0 = y;
#pragma unroll 1
    for (int j = 0; j < 5; ++j) {
        const double xx = __dmul_rn(x, x), yy = __dmul_rn(y, y);
        const double r2 = __dadd_rn(xx, yy);
        // (1 + ((k[7]*r2 + k[6])*r2 + k[5])*r2) with k5..k7 = 0 is exactly 1 (r2 is finite and >= 0)
        const double den = __dadd_rn(1., __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(k3, r2), k2), r2), k1), r2));
        const double icdist = __ddiv_rn(1., den);
        // 2*k[2]*x*y + k[3]*(r2 + 2*x*x) + k[8]*r2 + k[9]*r2*r2 with k8 = k9 = 0 (adding +0 changes nothing but the
        // sign of a -0 sum, which the subtraction from x0 below cannot observe)
        const double dX = __dadd_rn(__dmul_rn(__dmul_rn(__dmul_rn(2., p1), x), y), __dmul_rn(p2, __dadd_rn(r2, __dmul_rn(__dmul_rn(2., x), x))));
        const double dY = __dadd_rn(__dmul_rn(p1, __dadd_rn(r2, __dmul_rn(__dmul_rn(2., y), y))), __dmul_rn(__dmul_rn(__dmul_rn(2., p2), x), y));
        x = __dmul_rn(__dsub_rn(x0, dX), icdist);
        y = __dmul_rn(__dsub_rn(y0, dY), icdist);
    }
    // RR = K * I; xx = RR00*x + RR01*y + RR02 with RR01 = 0: (fx*x + 0*y) + cx; ww = 1/(0*x + 0*y + 1) = 1
    const double X = __dadd_rn(__dadd_rn(__dmul_rn(fx, x), __dmul_rn(0., y)), cx);
    const double Y = __dadd_rn(__dadd_rn(__dmul_rn(0., x), __dmul_rn(fy, y)), cy);
    *ox = (float)X;
    *oy = (float)Y;
}

__global__ void k_image_bounds(PslCamera C, int cols, int rows, float* __restrict__ bounds) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (C.k1 != 0.0f) {
        float cx[4], cy[4];
        psl_undistort_point(0., 0., C, &cx[0], &cy[0]);
        psl_undistort_point((double)(float)cols, 0., C, &cx[1], &cy[1]);
        psl_undistort_point(0., (double)(float)rows, C, &cx[2], &cy[2]);
        psl_undistort_point((double)(float)cols, (double)(float)rows, C, &cx[3], &cy[3]);
        bounds[0] = fminf(cx[0], cx[2]); bounds[2] = fmaxf(cx[1], cx[3]);
        bounds[1] = fminf(cy[0], cy[1]); bounds[3] = fmaxf(cy[2], cy[3]);
    } else { bounds[0] = 0.f; bounds[2] = (float)cols; bounds[1] = 0.f; bounds[3] = (float)rows; }
}

// grid (ceil(cap/256), nframes): keypoints of slot0+blockIdx.y are undistorted in place, depth/uright filled.
__global__ __launch_bounds__(256) void k_frame_post_rgbd(FrameStore S, float* __restrict__ mvDepth, int slot0, const float* __restrict__ depth,
                                                          int w, int h, int dstride, size_t dframe, PslCamera C) {
    const int slot = slot0 + blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S.meta[slot].n) return;
    PslKeyPoint* kp = S.kps + (size_t)slot * S.cap + i;
    const float x = kp->x, y = kp->y;
    float xu = x, yu = y;
    if (C.k1 != 0.0f) psl_undistort_point((double)x, (double)y, C, &xu, &yu);
    const int u = (int)x, v = (int)y;  // imDepth.at<float>(v,u) with float arguments truncates
    float d = 0.f;
    if (u >= 0 && u < w && v >= 0 && v < h) d = depth[(size_t)blockIdx.y * dframe + (size_t)v * dstride + u];
    float dep = -1.f, ur = -1.f;
    if (d > 0) { dep = d; ur = PSL_FSUB(xu, PSL_FDIV(C.bf, d)); }
    kp->x = xu; kp->y = yu;
    mvDepth[(size_t)slot * S.cap + i] = dep;
    S.uright[(size_t)slot * S.cap + i] = ur;
}

// meta of slots slot0..slot0+nslots-1 <- grid geometry from the device-side bounds (src/Frame.cc:163-164)
__global__ void k_frame_meta_bounds(FrameStore S, int slot0, int nslots, const float* __restrict__ bounds) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nslots) return;
    FrameMeta m = S.meta[slot0 + k];
    m.minX = bounds[0]; m.minY = bounds[1];
    m.invW = PSL_FDIV((float)PSL_GRID_COLS, PSL_FSUB(bounds[2], bounds[0]));
    m.invH = PSL_FDIV((float)PSL_GRID_ROWS, PSL_FSUB(bounds[3], bounds[1]));
    S.meta[slot0 + k] = m;
}

// mGrid[ix][iy] as CSR with cell = ix*48+iy (the order GetFeaturesInArea walks), indices ascending
// inside a cell (push_back order of AssignFeaturesToGrid).
__global__ __launch_bounds__(1024) void k_build_grid(FrameStore S, int slot0) {
    __shared__ int s_cnt[PSL_GRID_CELLS];
    __shared__ int s_w[17];
    const int slot = slot0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const FrameMeta M = S.meta[slot];
    const PslKeyPoint* kps = S.kps + (size_t)slot * S.cap;
    uint16_t* cellof = S.cellof + (size_t)slot * S.cap;
    int* gstart = S.gstart + (size_t)slot * (PSL_GRID_CELLS + 1);
    int* gidx = S.gidx + (size_t)slot * S.cap;
    for (int c = tid; c < PSL_GRID_CELLS; c += 1024) s_cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < M.n; i += 1024) {
        const int posX = (int)__builtin_roundf(PSL_FMUL(PSL_FSUB(kps[i].x, M.minX), M.invW));  // PosInGrid :1042-1043
        const int posY = (int)__builtin_roundf(PSL_FMUL(PSL_FSUB(kps[i].y, M.minY), M.invH));
        const bool ok = posX >= 0 && posX < PSL_GRID_COLS && posY >= 0 && posY < PSL_GRID_ROWS;
        const int c = ok ? posX * PSL_GRID_ROWS + posY : 0xffff;
        cellof[i] = (uint16_t)c;
        if (ok) atomicAdd(&s_cnt[c], 1);
    }
    __syncthreads();
    const int c0 = tid * 3;
    const int a = s_cnt[c0], b = s_cnt[c0 + 1], c = s_cnt[c0 + 2];
    int inc = a + b + c;
    const int mine = inc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int k = 0; k < 16; ++k) { const int t = s_w[k]; s_w[k] = acc; acc += t; }
        s_w[16] = acc;
    }
    __syncthreads();
    const int st = inc - mine + s_w[wave];
    gstart[c0] = st; gstart[c0 + 1] = st + a; gstart[c0 + 2] = st + a + b;
    if (tid == 1023) gstart[PSL_GRID_CELLS] = s_w[16];
    s_cnt[c0] = st; s_cnt[c0 + 1] = st + a; s_cnt[c0 + 2] = st + a + b;  // fill cursors
    __syncthreads();
    for (int i = tid; i < M.n; i += 1024) {
        const int cc = cellof[i];
        if (cc != 0xffff) gidx[atomicAdd(&s_cnt[cc], 1)] = i;
    }
    __syncthreads();
    // ascending index inside each (tiny) cell run
    int lo = st;
    const int len[3] = {a, b, c};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        for (int i = lo + 1; i < lo + len[k]; ++i) {
            const int v = gidx[i];
            int j = i - 1;
            while (j >= lo && gidx[j] > v) { gidx[j + 1] = gidx[j]; --j; }
            gidx[j + 1] = v;
        }
        lo += len[k];
    }
}

// ---------------------------------------------------------------------------------------------
#define PSL_TOPK 8   // cached best candidates per query

// Pass 1 (wide): every query of every pair in parallel, one wave each; ignores first-come blocking.  Writes the
// PSL_TOPK smallest keys of the query in ascending order and whether the window holds more gated candidates.
__global__ __launch_bounds__(256) void k_window_eval(MatchArgs A) {
    const int pair = blockIdx.y, qi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    int nq = A.nq_arr ? A.nq_arr[pair] : A.nq_single;
    nq = min(min(nq, A.qstride), PSL_QMAX);
    if (qi >= nq) return;
    const FrameView V = psl_frame_view(A.S, A.slot0 + pair);
    const PslProjQuery q = A.q[(size_t)pair * A.qstride + qi];
    const uint32_t* QD = reinterpret_cast<const uint32_t*>(A.qdesc + ((size_t)pair * A.qstride + qi) * 32);
    uint32_t qd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qd[k] = QD[k];
    const uint8_t* taken = A.taken ? A.taken + (size_t)pair * A.S.cap : nullptr;
    const WindowCols W = psl_window_cols(V, q, A.fidx);
    uint32_t best = PSL_KEY_INF;  // lanes 0..PSL_TOPK-1: running smallest keys, ascending
    int cnt = 0;
    for (int base = 0; base < W.T; base += 64) {
        uint32_t key = psl_window_key(V, q, qd, taken, nullptr, qi, W, base + lane, A.fidx, A.no_stereo);
        cnt += __popcll(__ballot(key != PSL_KEY_INF));
        key = psl_wave_sort(key);
        if (base > 0) {  // merge this round's smallest with the running ones
            const uint32_t o = __shfl(key, (lane - PSL_TOPK) & 63);
            key = psl_wave_sort(lane < PSL_TOPK ? best : (lane < 2 * PSL_TOPK ? o : PSL_KEY_INF));
        }
        best = key;
    }
    if (lane < PSL_TOPK) A.topk[((size_t)pair * A.qstride + qi) * PSL_TOPK + lane] = best;
    if (lane == 0) A.more[(size_t)pair * A.qstride + qi] = cnt > PSL_TOPK;
}

// Pass 1 for the many-frames launches, STAGED: one workgroup per frame copies what the window search reads of the frame - the CSR
// grid, the keypoints' (x, y), octave and mvuRight, the descriptors: 72 KB for up to PSL_WS_CAP keypoints - into LDS once, coalesced,
// and its 16 waves then work through the frame's queries (wave = query, as above) without touching HBM again except for the query
// itself.  The wave-per-query kernel above issues three DEPENDENT global fetches per candidate (run bounds -> keypoint index ->
// keypoint + descriptor) from 12 M independent waves per launch: 8.7 GB fetched for 0.9 GB of frames, two thirds of a wave's life
// spent waiting (profiles/r03e_*).  Same arithmetic, same keys, same order.
#define PSL_WS_CAP 1280
__global__ __launch_bounds__(1024, 8) void k_window_eval_staged(MatchArgs A) {   // 64 VGPRs: two workgroups (72 KB of LDS each) per CU
    __shared__ int s_gstart[PSL_GRID_CELLS + 1];
    __shared__ uint16_t s_gidx[PSL_WS_CAP];
    __shared__ float2 s_xy[PSL_WS_CAP];
    __shared__ float s_ur[PSL_WS_CAP];
    __shared__ uint8_t s_oct[PSL_WS_CAP];
    __shared__ uint4 s_desc[PSL_WS_CAP * 2];
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nq = A.nq_arr ? A.nq_arr[pair] : A.nq_single;
    nq = min(min(nq, A.qstride), PSL_QMAX);
    const FrameView V = psl_frame_view(A.S, A.slot0 + pair);
    const int n = min(V.n, PSL_WS_CAP);
    for (int i = tid; i < n; i += 1024) {
        s_xy[i] = *reinterpret_cast<const float2*>(&V.kps[i].x);
        s_oct[i] = (uint8_t)V.kps[i].octave;
        s_ur[i] = V.uright[i];
        s_gidx[i] = (uint16_t)V.gidx[i];
    }
    for (int i = tid; i < 2 * n; i += 1024) s_desc[i] = reinterpret_cast<const uint4*>(V.desc)[i];
    for (int c = tid; c <= PSL_GRID_CELLS; c += 1024) s_gstart[c] = V.gstart[c];
    __syncthreads();
    const FrameMeta& M = V.M;
    const PslProjQuery* Q = A.q + (size_t)pair * A.qstride;
    const uint4* QD = reinterpret_cast<const uint4*>(A.qdesc + (size_t)pair * A.qstride * 32);
    const uint8_t* taken = A.taken ? A.taken + (size_t)pair * A.S.cap : nullptr;
    // the next query is in flight while this one is evaluated
    int qi = wave;
    PslProjQuery qn = {};
    uint4 qa = {}, qb = {};
    if (qi < nq) { qn = Q[qi]; qa = QD[2 * (size_t)qi]; qb = QD[2 * (size_t)qi + 1]; }
    for (; qi < nq; qi += 16) {
        const PslProjQuery q = qn;
        const uint32_t qd[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
        if (qi + 16 < nq) { qn = Q[qi + 16]; qa = QD[2 * (size_t)(qi + 16)]; qb = QD[2 * (size_t)(qi + 16) + 1]; }
        // psl_window_cols on the staged grid
        const float r = q.radius;
        const int minCX = max(0, (int)__builtin_floorf(PSL_FMUL(PSL_FSUB(PSL_FSUB(q.u, M.minX), r), M.invW)));
        const int maxCX = min(PSL_GRID_COLS - 1, (int)__builtin_ceilf(PSL_FMUL(PSL_FADD(PSL_FSUB(q.u, M.minX), r), M.invW)));
        const int minCY = max(0, (int)__builtin_floorf(PSL_FMUL(PSL_FSUB(PSL_FSUB(q.v, M.minY), r), M.invH)));
        const int maxCY = min(PSL_GRID_ROWS - 1, (int)__builtin_ceilf(PSL_FMUL(PSL_FADD(PSL_FSUB(q.v, M.minY), r), M.invH)));
        const bool window = minCX < PSL_GRID_COLS && maxCX >= 0 && minCY < PSL_GRID_ROWS && maxCY >= 0;
        WindowCols W;
        W.start = 0;
        int len = 0;
        if (window && minCX + lane <= maxCX) {
            const int ix = minCX + lane;
            W.start = s_gstart[ix * PSL_GRID_ROWS + minCY];
            len = s_gstart[ix * PSL_GRID_ROWS + maxCY + 1] - W.start;
        }
        int incl = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
        W.incl = incl;
        W.excl = incl - len;
        W.T = __shfl(incl, 63);
        W.checkLevels = (q.min_level > 0) || (q.max_level >= 0);
        uint32_t best = PSL_KEY_INF;  // lanes 0..PSL_TOPK-1: running smallest keys, ascending
        int cnt = 0;
        for (int base = 0; base < W.T; base += 64) {
            // psl_window_key on the staged frame
            const int p = psl_window_pos(W, base + lane);
            uint32_t key = PSL_KEY_INF;
            if (p >= 0) {
                const int i2 = s_gidx[p];
                const float2 xy = s_xy[i2];
                const int octave = s_oct[i2];
                const float ur = s_ur[i2];
                const uint4 d0 = s_desc[2 * i2], d1 = s_desc[2 * i2 + 1];
                bool ok = i2 < n;
                if (W.checkLevels) ok = ok && !(octave < q.min_level) && !(q.max_level >= 0 && octave > q.max_level);
                ok = ok && (__builtin_fabsf(PSL_FSUB(xy.x, q.u)) < r && __builtin_fabsf(PSL_FSUB(xy.y, q.v)) < r);
                if (taken) ok = ok && !taken[i2];
                if (!A.no_stereo) ok = ok && !(ur > 0 && __builtin_fabsf(PSL_FSUB(q.ur, ur)) > r);
                const int dist = __popc(qd[0] ^ d0.x) + __popc(qd[1] ^ d0.y) + __popc(qd[2] ^ d0.z) + __popc(qd[3] ^ d0.w) +
                                 __popc(qd[4] ^ d1.x) + __popc(qd[5] ^ d1.y) + __popc(qd[6] ^ d1.z) + __popc(qd[7] ^ d1.w);
                if (ok) key = ((uint32_t)dist << 16) | (uint32_t)p;
            }
            cnt += __popcll(__ballot(key != PSL_KEY_INF));
            key = psl_wave_sort(key);
            if (base > 0) {  // merge this round's smallest with the running ones
                const uint32_t o = __shfl(key, (lane - PSL_TOPK) & 63);
                key = psl_wave_sort(lane < PSL_TOPK ? best : (lane < 2 * PSL_TOPK ? o : PSL_KEY_INF));
            }
            best = key;
        }
        if (lane < PSL_TOPK) A.topk[((size_t)pair * A.qstride + qi) * PSL_TOPK + lane] = best;
        if (lane == 0) A.more[(size_t)pair * A.qstride + qi] = cnt > PSL_TOPK;
    }
}

// Pass 2: one workgroup per frame resolves the sequential semantics by a fixpoint on "taken by an earlier
// query".  A thread owns QM/BS queries and keeps their cached candidate lists (key, keypoint, octave) in
// registers, so an iteration touches only LDS: phase A picks every query's best candidate not taken by an
// earlier query (blockers of the previous iteration), phase B rebuilds the blockers from the picks.  A query
// whose whole list is blocked while its window holds more candidates is re-scanned in full by a wave (rare).
// Then the rotation histogram and the outputs.
// MODE 0: SearchByProjection(cur,last) (:1328-1470); MODE 1: SearchByProjection(F, MapPoints) (:45-129).
template <int MODE, int QM, int BS>
__global__ __launch_bounds__(BS) void k_window_resolve(MatchArgs A) {
    constexpr int R = QM / BS;
    __shared__ int s_choice[QM];
    __shared__ int s_blk[2][QM];
    __shared__ uint8_t s_bin[QM];
    __shared__ uint8_t s_slow[QM];
    __shared__ int s_hist[PSL_HISTO];
    __shared__ int s_ind[3];
    __shared__ int s_changed[3], s_anyslow[3], s_nm;
    int* s_blocker = s_blk[0];

    const int pair = blockIdx.x, slot = A.slot0 + pair, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const FrameStore& S = A.S;
    const FrameView V = psl_frame_view(S, slot);
    const int n = V.n < QM ? V.n : QM;
    int nq = A.nq_arr ? A.nq_arr[pair] : A.nq_single;
    nq = min(min(nq, A.qstride), QM);
    const PslProjQuery* Q = A.q + (size_t)pair * A.qstride;
    const uint32_t* QD = reinterpret_cast<const uint32_t*>(A.qdesc + (size_t)pair * A.qstride * 32);
    const uint8_t* taken = A.taken ? A.taken + (size_t)pair * S.cap : nullptr;
    const uint32_t* TK = A.topk + (size_t)pair * A.qstride * PSL_TOPK;
    const uint8_t* MORE = A.more + (size_t)pair * A.qstride;
    const int* posmap = A.fidx ? A.fidx : V.gidx;  // candidate position -> keypoint of the frame

    // candidate lists of this thread's queries: keys (distance << 16 | CSR position, ascending, INF-terminated) and
    // per candidate keypoint index | octave << 12
    uint32_t K[R][PSL_TOPK];
    uint32_t CP[R][PSL_TOPK / 2];  // two 16-bit entries per register
    uint32_t flags = 0;  // bit r: query blocks, bit 8 + r: window holds more candidates than the list
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int qi = tid + r * BS;
        if (qi < nq) {
            const uint4 k0 = *reinterpret_cast<const uint4*>(TK + (size_t)qi * PSL_TOPK);
            const uint4 k1 = *reinterpret_cast<const uint4*>(TK + (size_t)qi * PSL_TOPK + 4);
            K[r][0] = k0.x; K[r][1] = k0.y; K[r][2] = k0.z; K[r][3] = k0.w; K[r][4] = k1.x; K[r][5] = k1.y; K[r][6] = k1.z; K[r][7] = k1.w;
            flags |= (Q[qi].blocks ? 1u : 0u) << r;
            flags |= (MORE[qi] ? 1u : 0u) << (8 + r);
        } else {
#pragma unroll
            for (int e = 0; e < PSL_TOPK; ++e) K[r][e] = PSL_KEY_INF;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int h = 0; h < PSL_TOPK / 2; ++h) {
            uint32_t c0 = 0, c1 = 0;
            if (K[r][2 * h] != PSL_KEY_INF) { c0 = (uint32_t)posmap[K[r][2 * h] & 0xffff]; if (MODE == 1) c0 |= (uint32_t)V.kps[c0].octave << 12; }
            if (K[r][2 * h + 1] != PSL_KEY_INF) { c1 = (uint32_t)posmap[K[r][2 * h + 1] & 0xffff]; if (MODE == 1) c1 |= (uint32_t)V.kps[c1].octave << 12; }
            CP[r][h] = c0 | (c1 << 16);
        }
#define PSL_CI(r, e) ((CP[r][(e) >> 1] >> (16 * ((e) & 1))) & 0xffffu)

    for (int i = tid; i < nq; i += BS) s_choice[i] = -2;
    for (int i = tid; i < n; i += BS) s_blk[0][i] = 0x7fffffff;
    if (tid < 3) { s_changed[tid] = 0; s_anyslow[tid] = 0; }
    __syncthreads();

    // decide a query from its two best non-blocked candidates (key, keypoint | octave << 12)
    auto decide = [&](uint32_t k1, uint32_t c1, uint32_t k2, uint32_t c2) -> int {
        if (k1 == PSL_KEY_INF) return -1;
        const int bestDist = (int)(k1 >> 16);
        bool ok = bestDist <= A.th;
        if (MODE == 1 && ok && k2 != PSL_KEY_INF) {
            const int bestDist2 = (int)(k2 >> 16);
            if ((c1 >> 12) == (c2 >> 12) && (float)bestDist > PSL_FMUL(A.nnratio, (float)bestDist2)) ok = false;  // (:118-121)
        }
        if (MODE == 2 && ok) {  // SearchByBoW (:229-231): ratio against the second best, 256 when there is none
            const int bestDist2 = k2 != PSL_KEY_INF ? (int)(k2 >> 16) : 256;
            if (!((float)bestDist < PSL_FMUL(A.nnratio, (float)bestDist2))) ok = false;
        }
        return ok ? (int)(c1 & 0xfff) : -1;
    };

    int cur = 0;
    for (int iter = 0; iter <= nq; ++iter) {
        const int f = iter % 3;
        const int* blk = s_blk[cur];
        // phase A: picks from the cached lists
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int qi = tid + r * BS;
            if (qi < nq) {
                uint32_t k1 = PSL_KEY_INF, k2 = PSL_KEY_INF, c1 = 0, c2 = 0;
                int found = 0;
                bool exhausted = true;  // the list ended before we had what we need
#pragma unroll
                for (int e = 0; e < PSL_TOPK; ++e) {
                    const uint32_t key = K[r][e];
                    const bool live = !(found == (MODE == 0 ? 1 : 2)) && key != PSL_KEY_INF;
                    const uint32_t ci = PSL_CI(r, e);
                    if (live && !(blk[ci & 0xfff] < qi)) {
                        if (found == 0) { k1 = key; c1 = ci; found = 1; }
                        else { k2 = key; c2 = ci; found = 2; }
                    }
                }
                exhausted = found < (MODE == 0 ? 1 : 2);
                const bool slow = exhausted && ((flags >> (8 + r)) & 1);  // the list ran out but the window holds more
                s_slow[qi] = slow;
                if (slow) s_anyslow[f] = 1;
                else {
                    const int pick = decide(k1, c1, k2, c2);
                    if (s_choice[qi] != pick) { s_choice[qi] = pick; s_changed[f] = 1; }
                }
            }
        }
        for (int i = tid; i < n; i += BS) s_blk[cur ^ 1][i] = 0x7fffffff;
        if (tid == 0) { s_changed[(iter + 1) % 3] = 0; s_anyslow[(iter + 1) % 3] = 0; }
        __syncthreads();
        if (s_anyslow[f]) {  // slow path: full window scan with the current blockers, one wave per query
            for (int qi = wave; qi < nq; qi += BS / 64) {
                if (!s_slow[qi]) continue;
                const PslProjQuery q = Q[qi];
                uint32_t qd[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) qd[k] = QD[(size_t)qi * 8 + k];
                const WindowCols W = psl_window_cols(V, q, A.fidx);
                uint32_t t0 = PSL_KEY_INF, t1 = PSL_KEY_INF;  // two smallest keys
                for (int base = 0; base < W.T; base += 64) {
                    const uint32_t key = psl_window_key(V, q, qd, taken, blk, qi, W, base + lane, A.fidx, A.no_stereo);
                    if (key < t0) { t1 = t0; t0 = key; } else if (key < t1) t1 = key;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const uint32_t a0 = __shfl_xor(t0, o), a1 = __shfl_xor(t1, o);
                    const uint32_t lo = min(t0, a0), hi = max(t0, a0);
                    t1 = min(hi, min(t1, a1));
                    t0 = lo;
                }
                uint32_t c1 = 0, c2 = 0;
                if (t0 != PSL_KEY_INF) { c1 = posmap[t0 & 0xffff]; if (MODE == 1) c1 |= (uint32_t)V.kps[c1].octave << 12; }
                if (t1 != PSL_KEY_INF) { c2 = posmap[t1 & 0xffff]; if (MODE == 1) c2 |= (uint32_t)V.kps[c2].octave << 12; }
                const int pick = decide(t0, c1, t1, c2);
                if (lane == 0 && s_choice[qi] != pick) { s_choice[qi] = pick; s_changed[f] = 1; }
            }
            __syncthreads();
        }
        if (!s_changed[f]) break;
        // phase B: blockers of the next iteration
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int qi = tid + r * BS;
            if (qi < nq && ((flags >> r) & 1)) {
                const int c = s_choice[qi];
                if (c >= 0) atomicMin(&s_blk[cur ^ 1][c], qi);
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    __syncthreads();

    // rotation consistency (:1431-1467)
    if (tid < PSL_HISTO) s_hist[tid] = 0;
    if (tid == 0) { s_ind[0] = s_ind[1] = s_ind[2] = -1; s_nm = 0; }
    __syncthreads();
    const bool ori = (MODE == 0 || MODE == 2) && A.check_ori;
    if (ori) {
        const float factor = 1.0f / PSL_HISTO;
        for (int qi = tid; qi < nq; qi += BS) {
            const int c = s_choice[qi];
            if (c < 0) continue;
            float rot = PSL_FSUB(Q[qi].angle, V.kps[c].angle);
            if (rot < 0.0f) rot = PSL_FADD(rot, 360.0f);
            int bin = (int)__builtin_roundf(PSL_FMUL(rot, factor));
            if (bin == PSL_HISTO) bin = 0;
            bin = bin < 0 ? 0 : (bin >= PSL_HISTO ? PSL_HISTO - 1 : bin);
            s_bin[qi] = (uint8_t)bin;
            atomicAdd(&s_hist[bin], 1);
        }
        __syncthreads();
        if (tid == 0) {  // ComputeThreeMaxima (:1601-1645)
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
    // owners: the last query that assigned a keypoint, unless one of its assignments was filtered
    for (int i = tid; i < n; i += BS) s_blocker[i] = -1;
    __syncthreads();
    int local = 0;
    int* match = A.match + (size_t)pair * A.qstride;
    for (int qi = tid; qi < nq; qi += BS) {
        const int c = s_choice[qi];
        bool good = c >= 0;
        if (good) atomicMax(&s_blocker[c], qi);
        if (good && ori) { const int b = s_bin[qi]; good = (b == s_ind[0] || b == s_ind[1] || b == s_ind[2]); }
        match[qi] = good ? c : -1;
        local += good;
    }
    __syncthreads();
    if (ori)
        for (int qi = tid; qi < nq; qi += BS) {
            const int c = s_choice[qi];
            if (c < 0) continue;
            const int b = s_bin[qi];
            if (!(b == s_ind[0] || b == s_ind[1] || b == s_ind[2])) s_blocker[c] = -1;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if (lane == 0 && local) atomicAdd(&s_nm, local);
    __syncthreads();
    if (A.assigned) {
        int* asg = A.assigned + (size_t)pair * S.cap;
        for (int i = tid; i < V.M.n; i += BS) asg[i] = i < n ? s_blocker[i] : -1;
    }
    if (tid == 0) A.nmatches[pair] = s_nm;
}

// BFMatcher(NORM_HAMMING).knnMatch(k = 2): one wave per query row, lanes stride the train rows.
__global__ __launch_bounds__(256) void k_hamming_knn2(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t, int nt,
                                                       int* __restrict__ idx, int* __restrict__ dist) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (qi >= nq) return;
    uint32_t qd[8];
    const uint32_t* Q = reinterpret_cast<const uint32_t*>(q) + (size_t)qi * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) qd[k] = Q[k];
    const uint32_t* T = reinterpret_cast<const uint32_t*>(t);
    uint32_t k1 = 0xffffffffu, k2 = 0xffffffffu;
    for (int j = lane; j < nt; j += 64) {
        const uint32_t key = ((uint32_t)psl_hamming256(qd, T + (size_t)j * 8) << 20) | (uint32_t)j;  // nt < 2^20
        if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) psl_merge2(k1, k2, __shfl_xor(k1, o), __shfl_xor(k2, o));
    if (lane == 0) {
        idx[2 * qi] = k1 == 0xffffffffu ? -1 : (int)(k1 & 0xfffff);
        dist[2 * qi] = k1 == 0xffffffffu ? 0x7fffffff : (int)(k1 >> 20);
        idx[2 * qi + 1] = k2 == 0xffffffffu ? -1 : (int)(k2 & 0xfffff);
        dist[2 * qi + 1] = k2 == 0xffffffffu ? 0x7fffffff : (int)(k2 >> 20);
    }
}

// ---------------------------------------------------------------------------------------------

int pslfe_orb_internal_last(pslfe_orb* orb, const PslKeyPoint** kps, const uint8_t** desc, const int** counts, int* cap,
                            int* nframes, pslfe_ctx** ctx);

namespace {
int host_search(pslfe_frame* f, int slot, const PslProjQuery* queries, const uint8_t* qdesc, int nq, const uint8_t* taken,
                int mode, int check_ori, float nnratio, int32_t* match, int32_t* assigned, int* nmatches, int th = PSL_TH_HIGH,
                int no_stereo = 0, const int32_t* fidx = nullptr, int nfidx = 0) {
    PSL_REQUIRE(f && nmatches && (nq == 0 || (queries && qdesc && match)), PSLFE_E_INVALID, "search_by_projection: NULL argument");
    PSL_REQUIRE(slot >= 0 && slot < f->max_frames && f->slot_set[slot], PSLFE_E_STATE, "search_by_projection: slot %d not set", slot);
    PSL_REQUIRE(nq >= 0 && nq <= PSL_QMAX, PSLFE_E_INVALID, "search_by_projection: %d queries (max %d)", nq, PSL_QMAX);
    *nmatches = 0;
    if (nq == 0) return PSLFE_OK;
    PSL_HIP(hipSetDevice(f->ctx->device));
    hipStream_t st = f->ctx->stream;
    PSL_HIP(hipMemcpyAsync(f->d_q, queries, (size_t)nq * sizeof(PslProjQuery), hipMemcpyHostToDevice, st));
    PSL_HIP(hipMemcpyAsync(f->d_qdesc, qdesc, (size_t)nq * 32, hipMemcpyHostToDevice, st));
    FrameMeta m;
    PSL_HIP(hipMemcpyAsync(&m, f->S.meta + slot, sizeof(m), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    if (taken) PSL_HIP(hipMemcpyAsync(f->d_taken, taken, (size_t)m.n, hipMemcpyHostToDevice, st));
    if (fidx) {  // the frame's FeatureVector: staged in the "assigned" scratch's sibling buffer
        PSL_REQUIRE(nfidx >= 0 && nfidx <= f->cap, PSLFE_E_CAPACITY, "search_by_bow: %d feature-vector entries, capacity %d", nfidx, f->cap);
        if (nfidx) PSL_HIP(hipMemcpyAsync(f->d_fidx, fidx, (size_t)nfidx * sizeof(int), hipMemcpyHostToDevice, st));
    }
    MatchArgs A;
    A.S = f->S;
    A.th = th; A.no_stereo = no_stereo; A.fidx = fidx ? f->d_fidx : nullptr;
    // host calls address exactly one slot; scratch arrays are indexed as "pair 0"
    A.slot0 = slot; A.q = f->d_q; A.qdesc = f->d_qdesc; A.nq_arr = nullptr; A.nq_single = nq; A.qstride = PSL_QMAX;
    A.taken = taken ? f->d_taken : nullptr; A.check_ori = check_ori; A.nnratio = nnratio;
    A.match = f->d_match; A.assigned = f->d_assigned; A.nmatches = f->d_nm;
    A.topk = f->d_topk1; A.more = f->d_more1;
    {
        PSL_STAGE_BEGIN(f->ctx, "match.window");
        k_window_eval<<<dim3((nq + 3) / 4, 1), 256, 0, st>>>(A);
        if (mode == 0) k_window_resolve<0, PSL_QMAX, 1024><<<1, 1024, 0, st>>>(A);
        else if (mode == 1) k_window_resolve<1, PSL_QMAX, 1024><<<1, 1024, 0, st>>>(A);
        else k_window_resolve<2, PSL_QMAX, 1024><<<1, 1024, 0, st>>>(A);
        PSL_STAGE_END(f->ctx, "match.window");
    }
    PSL_HIP(hipGetLastError());
    PSL_HIP(hipMemcpyAsync(match, f->d_match, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, st));
    if (assigned && m.n > 0) PSL_HIP(hipMemcpyAsync(assigned, f->d_assigned, (size_t)m.n * sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipMemcpyAsync(nmatches, f->d_nm, sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    return PSLFE_OK;
}
}  // namespace

extern "C" {

int pslfe_frame_create(pslfe_ctx* ctx, int max_keypoints, int max_frames, pslfe_frame** out) {
    PSL_REQUIRE(ctx && out, PSLFE_E_INVALID, "pslfe_frame_create: NULL argument");
    *out = nullptr;
    PSL_REQUIRE(max_keypoints >= 1 && max_keypoints <= PSL_QMAX && max_frames >= 1, PSLFE_E_INVALID,
                "pslfe_frame_create: max_keypoints %d (1..%d), max_frames %d", max_keypoints, PSL_QMAX, max_frames);
    PSL_HIP(hipSetDevice(ctx->device));
    pslfe_frame* f = new pslfe_frame();
    f->ctx = ctx; f->cap = max_keypoints; f->max_frames = max_frames;
    f->slot_set.assign(max_frames, 0);
    const size_t F = (size_t)max_frames, K = (size_t)max_keypoints;
    f->S.cap = max_keypoints;
    hipError_t e = hipSuccess;
    auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 1); };
    A((void**)&f->S.kps, F * K * sizeof(PslKeyPoint));
    A((void**)&f->S.desc, F * K * 32);
    A((void**)&f->S.uright, F * K * sizeof(float));
    A((void**)&f->S.cellof, F * K * sizeof(uint16_t));
    A((void**)&f->S.gstart, F * (PSL_GRID_CELLS + 1) * sizeof(int));
    A((void**)&f->S.gidx, F * K * sizeof(int));
    A((void**)&f->S.meta, F * sizeof(FrameMeta));
    A((void**)&f->d_q, PSL_QMAX * sizeof(PslProjQuery));
    A((void**)&f->d_qdesc, PSL_QMAX * 32);
    A((void**)&f->d_taken, K);
    A((void**)&f->d_match, PSL_QMAX * sizeof(int));
    A((void**)&f->d_assigned, K * sizeof(int));
    A((void**)&f->d_nm, sizeof(int));
    A((void**)&f->d_topk, F * K * PSL_TOPK * sizeof(uint32_t));
    A((void**)&f->d_more, F * K);
    A((void**)&f->d_topk1, (size_t)PSL_QMAX * PSL_TOPK * sizeof(uint32_t));
    A((void**)&f->d_more1, PSL_QMAX);
    A((void**)&f->d_fidx, K * sizeof(int));
    A((void**)&f->d_depth, F * K * sizeof(float));
    A((void**)&f->d_bounds, 4 * sizeof(float));
    if (e != hipSuccess) {
        pslfe_set_error("pslfe_frame_create: hipMalloc failed: %s", hipGetErrorString(e));
        pslfe_frame_destroy(f);
        return PSLFE_E_HIP;
    }
    e = hipMemset(f->S.meta, 0, F * sizeof(FrameMeta));
    *out = f;
    return PSLFE_OK;
}

void pslfe_frame_destroy(pslfe_frame* f) {
    if (!f) return;
    hipSetDevice(f->ctx->device);
    hipStreamSynchronize(f->ctx->stream);
    hipFree(f->S.kps); hipFree(f->S.desc); hipFree(f->S.uright); hipFree(f->S.cellof); hipFree(f->S.gstart);
    hipFree(f->S.gidx); hipFree(f->S.meta); hipFree(f->d_q); hipFree(f->d_qdesc); hipFree(f->d_taken);
    hipFree(f->d_match); hipFree(f->d_assigned); hipFree(f->d_nm);
    hipFree(f->d_topk); hipFree(f->d_more); hipFree(f->d_topk1); hipFree(f->d_more1);
    hipFree(f->d_depth); hipFree(f->d_bounds); hipFree(f->d_fidx);
    delete f;
}

int pslfe_frame_set(pslfe_frame* f, int slot, const PslKeyPoint* kps, const uint8_t* desc, const float* uright, int n,
                    float min_x, float min_y, float max_x, float max_y) {
    PSL_REQUIRE(f && (n == 0 || (kps && desc)), PSLFE_E_INVALID, "pslfe_frame_set: NULL argument");
    PSL_REQUIRE(slot >= 0 && slot < f->max_frames, PSLFE_E_INVALID, "pslfe_frame_set: slot %d of %d", slot, f->max_frames);
    PSL_REQUIRE(n >= 0 && n <= f->cap, PSLFE_E_CAPACITY, "pslfe_frame_set: %d keypoints, capacity %d", n, f->cap);
    PSL_REQUIRE(max_x > min_x && max_y > min_y, PSLFE_E_INVALID, "pslfe_frame_set: empty image bounds");
    PSL_HIP(hipSetDevice(f->ctx->device));
    hipStream_t st = f->ctx->stream;
    const size_t o = (size_t)slot * f->cap;
    std::vector<float> ur(n > 0 ? n : 1, -1.f);
    if (n > 0) {
        PSL_HIP(hipMemcpyAsync(f->S.kps + o, kps, (size_t)n * sizeof(PslKeyPoint), hipMemcpyHostToDevice, st));
        PSL_HIP(hipMemcpyAsync(f->S.desc + o * 32, desc, (size_t)n * 32, hipMemcpyHostToDevice, st));
        PSL_HIP(hipMemcpyAsync(f->S.uright + o, uright ? uright : ur.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
    }
    FrameMeta m;
    m.n = n; m.minX = min_x; m.minY = min_y;
    m.invW = (float)PSL_GRID_COLS / (float)(max_x - min_x);  // src/Frame.cc:163-164
    m.invH = (float)PSL_GRID_ROWS / (float)(max_y - min_y);
    PSL_HIP(hipMemcpyAsync(f->S.meta + slot, &m, sizeof(m), hipMemcpyHostToDevice, st));
    PSL_HIP(hipStreamSynchronize(st));  // host staging buffers go out of scope
    {
        PSL_STAGE_BEGIN(f->ctx, "match.grid");
        k_build_grid<<<1, 1024, 0, st>>>(f->S, slot);
        PSL_STAGE_END(f->ctx, "match.grid");
    }
    PSL_HIP(hipGetLastError());
    f->slot_set[slot] = 1;
    return PSLFE_OK;
}

int pslfe_frame_set_from_orb(pslfe_frame* f, pslfe_orb* orb, float min_x, float min_y, float max_x, float max_y) {
    PSL_REQUIRE(f && orb, PSLFE_E_INVALID, "pslfe_frame_set_from_orb: NULL argument");
    PSL_REQUIRE(max_x > min_x && max_y > min_y, PSLFE_E_INVALID, "pslfe_frame_set_from_orb: empty image bounds");
    const PslKeyPoint* okps; const uint8_t* odesc; const int* ocnt; int ocap, nframes; pslfe_ctx* octx;
    int rc = pslfe_orb_internal_last(orb, &okps, &odesc, &ocnt, &ocap, &nframes, &octx);
    if (rc) return rc;
    PSL_REQUIRE(octx == f->ctx, PSLFE_E_INVALID, "pslfe_frame_set_from_orb: handles belong to different contexts");
    PSL_REQUIRE(nframes <= f->max_frames, PSLFE_E_CAPACITY, "pslfe_frame_set_from_orb: %d frames, %d slots", nframes, f->max_frames);
    PSL_REQUIRE(ocap <= f->cap, PSLFE_E_CAPACITY, "pslfe_frame_set_from_orb: extractor capacity %d > frame capacity %d", ocap, f->cap);
    PSL_HIP(hipSetDevice(f->ctx->device));
    hipStream_t st = f->ctx->stream;
    const float invW = (float)PSL_GRID_COLS / (float)(max_x - min_x), invH = (float)PSL_GRID_ROWS / (float)(max_y - min_y);
    {
        PSL_STAGE_BEGIN(f->ctx, "match.grid");
        k_frame_import<<<nframes, 256, 0, st>>>(f->S, okps, odesc, ocnt, ocap, min_x, min_y, invW, invH);
        k_build_grid<<<nframes, 1024, 0, st>>>(f->S, 0);
        PSL_STAGE_END(f->ctx, "match.grid");
    }
    PSL_HIP(hipGetLastError());
    for (int s = 0; s < nframes; ++s) f->slot_set[s] = 1;
    return PSLFE_OK;
}


int pslfe_image_bounds(pslfe_frame* f, const PslCamera* cam, int cols, int rows, float* bounds) {
    PSL_REQUIRE(f && cam && bounds, PSLFE_E_INVALID, "pslfe_image_bounds: NULL argument");
    PSL_REQUIRE(cols > 0 && rows > 0, PSLFE_E_INVALID, "pslfe_image_bounds: %dx%d", cols, rows);
    PSL_HIP(hipSetDevice(f->ctx->device));
    k_image_bounds<<<1, 64, 0, f->ctx->stream>>>(*cam, cols, rows, f->d_bounds);
    PSL_HIP(hipMemcpyAsync(bounds, f->d_bounds, 4 * sizeof(float), hipMemcpyDeviceToHost, f->ctx->stream));
    PSL_HIP(hipStreamSynchronize(f->ctx->stream));
    return PSLFE_OK;
}

static int frame_post_rgbd(pslfe_frame* f, int slot0, int nslots, const float* d_depth, int w, int h, int dstride, size_t dframe,
                           const PslCamera* cam) {
    hipStream_t st = f->ctx->stream;
    PSL_STAGE_BEGIN(f->ctx, "frame.rgbd");
    k_image_bounds<<<1, 64, 0, st>>>(*cam, w, h, f->d_bounds);
    k_frame_meta_bounds<<<(nslots + 255) / 256, 256, 0, st>>>(f->S, slot0, nslots, f->d_bounds);
    k_frame_post_rgbd<<<dim3((f->cap + 255) / 256, nslots), 256, 0, st>>>(f->S, f->d_depth, slot0, d_depth, w, h, dstride, dframe, *cam);
    PSL_STAGE_END(f->ctx, "frame.rgbd");
    {
        PSL_STAGE_BEGIN(f->ctx, "match.grid");
        k_build_grid<<<nslots, 1024, 0, st>>>(f->S, slot0);
        PSL_STAGE_END(f->ctx, "match.grid");
    }
    PSL_HIP(hipGetLastError());
    for (int s = slot0; s < slot0 + nslots; ++s) f->slot_set[s] = 1;
    return PSLFE_OK;
}

int pslfe_frame_set_rgbd(pslfe_frame* f, int slot, const PslKeyPoint* kps, const uint8_t* desc, int n, const float* depth, int width,
                         int height, int depth_stride, const PslCamera* cam) {
    PSL_REQUIRE(f && cam && depth && (n == 0 || (kps && desc)), PSLFE_E_INVALID, "pslfe_frame_set_rgbd: NULL argument");
    PSL_REQUIRE(slot >= 0 && slot < f->max_frames, PSLFE_E_INVALID, "pslfe_frame_set_rgbd: slot %d of %d", slot, f->max_frames);
    PSL_REQUIRE(n >= 0 && n <= f->cap, PSLFE_E_CAPACITY, "pslfe_frame_set_rgbd: %d keypoints, capacity %d", n, f->cap);
    PSL_REQUIRE(width > 0 && height > 0 && depth_stride >= width, PSLFE_E_INVALID, "pslfe_frame_set_rgbd: depth %dx%d stride %d", width, height, depth_stride);
    PSL_HIP(hipSetDevice(f->ctx->device));
    hipStream_t st = f->ctx->stream;
    const size_t o = (size_t)slot * f->cap;
    { const int rc_ = psl_scratch_begin(f->ctx); if (rc_) return rc_; }
    float* d_img = static_cast<float*>(psl_scratch(f->ctx, (size_t)height * depth_stride * sizeof(float)));   // the context's scratch arena: no hipMalloc / hipFree per frame
    PSL_REQUIRE(d_img, PSLFE_E_HIP, "pslfe_frame_set_rgbd: out of device memory");
    hipError_t e = hipMemcpyAsync(d_img, depth, (size_t)height * depth_stride * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && n > 0) e = hipMemcpyAsync(f->S.kps + o, kps, (size_t)n * sizeof(PslKeyPoint), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && n > 0) e = hipMemcpyAsync(f->S.desc + o * 32, desc, (size_t)n * 32, hipMemcpyHostToDevice, st);
    FrameMeta m = {};
    m.n = n;
    if (e == hipSuccess) e = hipMemcpyAsync(f->S.meta + slot, &m, sizeof(m), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    int rc = PSLFE_OK;
    if (e != hipSuccess) { pslfe_set_error("pslfe_frame_set_rgbd: H2D: %s", hipGetErrorString(e)); rc = PSLFE_E_HIP; }
    if (!rc) rc = frame_post_rgbd(f, slot, 1, d_img, width, height, depth_stride, 0, cam);
    hipStreamSynchronize(st);
    return rc;
}

int pslfe_frame_set_from_orb_rgbd(pslfe_frame* f, pslfe_orb* orb, const float* d_depth, int width, int height, const PslCamera* cam) {
    PSL_REQUIRE(f && orb && d_depth && cam, PSLFE_E_INVALID, "pslfe_frame_set_from_orb_rgbd: NULL argument");
    PSL_REQUIRE(width > 0 && height > 0, PSLFE_E_INVALID, "pslfe_frame_set_from_orb_rgbd: depth %dx%d", width, height);
    const PslKeyPoint* okps; const uint8_t* odesc; const int* ocnt; int ocap, nframes; pslfe_ctx* octx;
    int rc = pslfe_orb_internal_last(orb, &okps, &odesc, &ocnt, &ocap, &nframes, &octx);
    if (rc) return rc;
    PSL_REQUIRE(octx == f->ctx, PSLFE_E_INVALID, "pslfe_frame_set_from_orb_rgbd: handles belong to different contexts");
    PSL_REQUIRE(nframes <= f->max_frames, PSLFE_E_CAPACITY, "pslfe_frame_set_from_orb_rgbd: %d frames, %d slots", nframes, f->max_frames);
    PSL_REQUIRE(ocap <= f->cap, PSLFE_E_CAPACITY, "pslfe_frame_set_from_orb_rgbd: extractor capacity %d > frame capacity %d", ocap, f->cap);
    PSL_HIP(hipSetDevice(f->ctx->device));
    {
        PSL_STAGE_BEGIN(f->ctx, "match.grid");
        k_frame_import<<<nframes, 256, 0, f->ctx->stream>>>(f->S, okps, odesc, ocnt, ocap, 0.f, 0.f, 1.f, 1.f);
        PSL_STAGE_END(f->ctx, "match.grid");
    }
    return frame_post_rgbd(f, 0, nframes, d_depth, width, height, width, (size_t)width * height, cam);
}

int pslfe_frame_fetch(pslfe_frame* f, int slot, PslKeyPoint* kps_un, float* depth, float* uright, int cap, int* n) {
    PSL_REQUIRE(f && n, PSLFE_E_INVALID, "pslfe_frame_fetch: NULL argument");
    PSL_REQUIRE(slot >= 0 && slot < f->max_frames && f->slot_set[slot], PSLFE_E_STATE, "pslfe_frame_fetch: slot %d not set", slot);
    PSL_HIP(hipSetDevice(f->ctx->device));
    hipStream_t st = f->ctx->stream;
    // through the context's pinned staging buffer (pslfe_internal.h): the slot's meta record, then the three arrays with one wait
    FrameMeta m;
    char* hs = psl_host_stage(f->ctx, sizeof(m));
    PSL_REQUIRE(hs, PSLFE_E_HIP, "pslfe_frame_fetch: no pinned staging memory (hipHostMalloc)");
    PSL_HIP(hipMemcpyAsync(hs, f->S.meta + slot, sizeof(m), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    memcpy(&m, hs, sizeof(m));
    *n = m.n;
    PSL_REQUIRE(m.n <= cap, PSLFE_E_CAPACITY, "pslfe_frame_fetch: %d keypoints, capacity %d", m.n, cap);
    const size_t o = (size_t)slot * f->cap;
    if (m.n > 0) {
        const size_t bk = psl_align_up((size_t)m.n * sizeof(PslKeyPoint), 16), bf = psl_align_up((size_t)m.n * sizeof(float), 16);
        hs = psl_host_stage(f->ctx, bk + 2 * bf);
        PSL_REQUIRE(hs, PSLFE_E_HIP, "pslfe_frame_fetch: no pinned staging memory (hipHostMalloc)");
        if (kps_un) PSL_HIP(hipMemcpyAsync(hs, f->S.kps + o, (size_t)m.n * sizeof(PslKeyPoint), hipMemcpyDeviceToHost, st));
        if (depth) PSL_HIP(hipMemcpyAsync(hs + bk, f->d_depth + o, (size_t)m.n * sizeof(float), hipMemcpyDeviceToHost, st));
        if (uright) PSL_HIP(hipMemcpyAsync(hs + bk + bf, f->S.uright + o, (size_t)m.n * sizeof(float), hipMemcpyDeviceToHost, st));
        PSL_HIP(hipStreamSynchronize(st));
        if (kps_un) memcpy(kps_un, hs, (size_t)m.n * sizeof(PslKeyPoint));
        if (depth) memcpy(depth, hs + bk, (size_t)m.n * sizeof(float));
        if (uright) memcpy(uright, hs + bk + bf, (size_t)m.n * sizeof(float));
    }
    return PSLFE_OK;
}

int pslfe_frame_debug_grid(pslfe_frame* f, int slot, int32_t* start, int32_t* idx, int cap, int* n) {
    PSL_REQUIRE(f && start && n, PSLFE_E_INVALID, "pslfe_frame_debug_grid: NULL argument");
    PSL_REQUIRE(slot >= 0 && slot < f->max_frames && f->slot_set[slot], PSLFE_E_STATE, "pslfe_frame_debug_grid: slot %d not set", slot);
    PSL_HIP(hipSetDevice(f->ctx->device));
    PSL_HIP(hipStreamSynchronize(f->ctx->stream));
    PSL_HIP(hipMemcpy(start, f->S.gstart + (size_t)slot * (PSL_GRID_CELLS + 1), (PSL_GRID_CELLS + 1) * sizeof(int), hipMemcpyDeviceToHost));
    *n = start[PSL_GRID_CELLS];
    PSL_REQUIRE(*n <= cap, PSLFE_E_CAPACITY, "pslfe_frame_debug_grid: %d entries, capacity %d", *n, cap);
    if (idx && *n > 0) PSL_HIP(hipMemcpy(idx, f->S.gidx + (size_t)slot * f->cap, (size_t)*n * sizeof(int), hipMemcpyDeviceToHost));
    return PSLFE_OK;
}

int pslfe_orb_search_by_projection_last(pslfe_frame* cur, int slot, const PslProjQuery* queries, const uint8_t* qdesc, int nq,
                                        const uint8_t* taken, int check_orientation, int32_t* match, int32_t* assigned, int* nmatches) {
    return host_search(cur, slot, queries, qdesc, nq, taken, 0, check_orientation, 0.f, match, assigned, nmatches);
}

int pslfe_orb_search_by_projection_map(pslfe_frame* cur, int slot, const PslProjQuery* queries, const uint8_t* qdesc, int nq,
                                       const uint8_t* taken, float nnratio, int32_t* match, int32_t* assigned, int* nmatches) {
    return host_search(cur, slot, queries, qdesc, nq, taken, 1, 0, nnratio, match, assigned, nmatches);
}

int pslfe_orb_search_by_projection_kf(pslfe_frame* cur, int slot, const PslProjQuery* queries, const uint8_t* qdesc, int nq,
                                      const uint8_t* taken, int orb_dist, int check_orientation, int32_t* match, int32_t* assigned,
                                      int* nmatches) {
    PSL_REQUIRE(orb_dist >= 0 && orb_dist <= 256, PSLFE_E_INVALID, "pslfe_orb_search_by_projection_kf: ORBdist %d", orb_dist);
    // every match occupies its keypoint (CurrentFrame.mvpMapPoints[bestIdx2] = pMP, :1566), so `blocks` is forced to 1
    std::vector<PslProjQuery> q(queries, queries + (nq > 0 && queries ? nq : 0));
    for (auto& e : q) e.blocks = 1;
    return host_search(cur, slot, q.data(), qdesc, nq, taken, 0, check_orientation, 0.f, match, assigned, nmatches, orb_dist, 1);
}

int pslfe_orb_search_by_bow(pslfe_frame* f, int slot, const int32_t* fidx, int nfidx, const PslBowQuery* queries, const uint8_t* qdesc,
                            int nq, float nnratio, int check_orientation, int32_t* match, int32_t* assigned, int* nmatches) {
    PSL_REQUIRE(f && (nq == 0 || queries) && (nfidx == 0 || fidx), PSLFE_E_INVALID, "pslfe_orb_search_by_bow: NULL argument");
    std::vector<PslProjQuery> q((size_t)(nq > 0 ? nq : 0));
    for (int i = 0; i < nq; ++i) {
        PSL_REQUIRE(queries[i].start >= 0 && queries[i].len >= 0 && queries[i].start + queries[i].len <= nfidx, PSLFE_E_INVALID,
                    "pslfe_orb_search_by_bow: query %d refers to entries %d..%d of %d", i, queries[i].start, queries[i].start + queries[i].len, nfidx);
        PslProjQuery e;
        memset(&e, 0, sizeof(e));
        e.min_level = queries[i].start; e.max_level = queries[i].len;  // the run, see MatchArgs::fidx
        e.angle = queries[i].angle;
        e.blocks = 1;  // vpMapPointMatches[bestIdxF] = pMP occupies the frame feature (:233)
        q[i] = e;
    }
    for (int i = 0; i < nfidx; ++i)
        PSL_REQUIRE(fidx[i] >= 0 && fidx[i] < f->cap, PSLFE_E_INVALID, "pslfe_orb_search_by_bow: feature index %d out of range", fidx[i]);
    return host_search(f, slot, q.data(), qdesc, nq, nullptr, 2, check_orientation, nnratio, match, assigned, nmatches, 50 /* TH_LOW :38 */, 1,
                       fidx, nfidx);
}

int pslfe_orb_search_by_projection_last_device(pslfe_frame* cur, int slot0, int npairs, const PslProjQuery* d_queries,
                                               const uint8_t* d_qdesc, const int32_t* d_nq, int qstride, int check_orientation,
                                               int32_t* d_match, int32_t* d_nmatches) {
    PSL_REQUIRE(cur && d_queries && d_qdesc && d_nq && d_match && d_nmatches, PSLFE_E_INVALID, "search_by_projection_last_device: NULL argument");
    PSL_REQUIRE(npairs >= 1 && slot0 >= 0 && slot0 + npairs <= cur->max_frames && qstride >= 1, PSLFE_E_INVALID,
                "search_by_projection_last_device: slots %d..%d of %d", slot0, slot0 + npairs - 1, cur->max_frames);
    for (int s = slot0; s < slot0 + npairs; ++s)
        PSL_REQUIRE(cur->slot_set[s], PSLFE_E_STATE, "search_by_projection_last_device: slot %d not set", s);
    PSL_HIP(hipSetDevice(cur->ctx->device));
    MatchArgs A;
    A.S = cur->S; A.slot0 = slot0; A.q = d_queries; A.qdesc = d_qdesc; A.nq_arr = d_nq; A.nq_single = 0; A.qstride = qstride;
    A.taken = nullptr; A.check_ori = check_orientation; A.nnratio = 0.f; A.match = d_match; A.assigned = nullptr; A.nmatches = d_nmatches;
    A.th = PSL_TH_HIGH; A.no_stereo = 0; A.fidx = nullptr;
    PSL_REQUIRE(qstride <= cur->cap && npairs <= cur->max_frames, PSLFE_E_CAPACITY, "search_by_projection_last_device: qstride %d > capacity %d", qstride, cur->cap);
    A.topk = cur->d_topk; A.more = cur->d_more;
    {
        PSL_STAGE_BEGIN(cur->ctx, "match.window");
        if (cur->cap <= PSL_WS_CAP && npairs >= 64)   // many frames of at most 1280 keypoints: the frame staged in LDS, one workgroup per frame
            k_window_eval_staged<<<npairs, 1024, 0, cur->ctx->stream>>>(A);
        else
            k_window_eval<<<dim3((std::min(qstride, PSL_QMAX) + 3) / 4, npairs), 256, 0, cur->ctx->stream>>>(A);
        // one 1024-thread workgroup per frame: measured faster than 512-thread workgroups at 256 and at 4096 frames
        k_window_resolve<0, PSL_QMAX, 1024><<<npairs, 1024, 0, cur->ctx->stream>>>(A);
        PSL_STAGE_END(cur->ctx, "match.window");
    }
    PSL_HIP(hipGetLastError());
    return PSLFE_OK;
}

int pslfe_hamming_knn2_device(pslfe_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_idx, int32_t* d_dist) {
    PSL_REQUIRE(ctx && d_idx && d_dist, PSLFE_E_INVALID, "pslfe_hamming_knn2_device: NULL argument");
    PSL_REQUIRE(nq >= 0 && nt >= 0 && nt < (1 << 20), PSLFE_E_INVALID, "pslfe_hamming_knn2_device: nq %d nt %d", nq, nt);
    if (nq == 0) return PSLFE_OK;
    PSL_REQUIRE(d_q && (nt == 0 || d_t), PSLFE_E_INVALID, "pslfe_hamming_knn2_device: NULL descriptors");
    PSL_HIP(hipSetDevice(ctx->device));
    {
        PSL_STAGE_BEGIN(ctx, "match.knn2");
        k_hamming_knn2<<<(nq + 3) / 4, 256, 0, ctx->stream>>>(d_q, nq, d_t, nt, d_idx, d_dist);
        PSL_STAGE_END(ctx, "match.knn2");
    }
    PSL_HIP(hipGetLastError());
    return PSLFE_OK;
}

int pslfe_hamming_knn2(pslfe_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist) {
    PSL_REQUIRE(ctx && idx && dist, PSLFE_E_INVALID, "pslfe_hamming_knn2: NULL argument");
    PSL_REQUIRE(nq >= 0 && nt >= 0 && nt < (1 << 20), PSLFE_E_INVALID, "pslfe_hamming_knn2: nq %d nt %d", nq, nt);
    if (nq == 0) return PSLFE_OK;
    PSL_REQUIRE(q && (nt == 0 || t), PSLFE_E_INVALID, "pslfe_hamming_knn2: NULL descriptors");
    PSL_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    uint8_t *dq = nullptr, *dt = nullptr;
    int *di = nullptr, *dd = nullptr;
    hipError_t e = hipMalloc((void**)&dq, (size_t)nq * 32);
    if (e == hipSuccess) e = hipMalloc((void**)&dt, nt ? (size_t)nt * 32 : 1);
    if (e == hipSuccess) e = hipMalloc((void**)&di, (size_t)nq * 2 * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&dd, (size_t)nq * 2 * sizeof(int));
    int rc = PSLFE_OK;
    if (e != hipSuccess) { pslfe_set_error("pslfe_hamming_knn2: hipMalloc: %s", hipGetErrorString(e)); rc = PSLFE_E_HIP; }
    if (!rc) {
        e = hipMemcpyAsync(dq, q, (size_t)nq * 32, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && nt) e = hipMemcpyAsync(dt, t, (size_t)nt * 32, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { pslfe_set_error("pslfe_hamming_knn2: H2D: %s", hipGetErrorString(e)); rc = PSLFE_E_HIP; }
    }
    if (!rc) rc = pslfe_hamming_knn2_device(ctx, dq, nq, dt, nt, di, dd);
    if (!rc) {
        e = hipMemcpyAsync(idx, di, (size_t)nq * 2 * sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(dist, dd, (size_t)nq * 2 * sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { pslfe_set_error("pslfe_hamming_knn2: D2H: %s", hipGetErrorString(e)); rc = PSLFE_E_HIP; }
    }
    hipFree(dq); hipFree(dt); hipFree(di); hipFree(dd);
    return rc;
}

int pslfe_line_match_nnr(pslfe_ctx* ctx, const uint8_t* desc1, int n1, const uint8_t* desc2, int n2, float nnr,
                         int32_t* matches12, int* nmatches) {
    PSL_REQUIRE(ctx && nmatches && (n1 == 0 || matches12), PSLFE_E_INVALID, "pslfe_line_match_nnr: NULL argument");
    *nmatches = 0;
    if (n1 <= 0) return PSLFE_OK;
    std::vector<int> idx((size_t)n1 * 2), dist((size_t)n1 * 2);
    int rc = pslfe_hamming_knn2(ctx, desc1, n1, desc2, n2, idx.data(), dist.data());
    if (rc) return rc;
    int m = 0;
    for (int i = 0; i < n1; ++i) {
        matches12[i] = -1;
        if (n2 < 2) continue;  // reference reads matches_[idx][1] out of bounds (:369): defined as no match
        if ((float)dist[2 * i] < (float)dist[2 * i + 1] * nnr) { matches12[i] = idx[2 * i]; ++m; }  // (:369)
    }
    *nmatches = m;
    return PSLFE_OK;
}

}  // extern "C"
