// Shared by pslfe_match.hip and pslfe_kf.hip: the frame store (keypoints bucketed on the 64x48 grid of include/Frame.h:45-46),
// the GetFeaturesInArea window as CSR runs, candidate keys and wave helpers.  Product code.
#ifndef PSL_MATCH_KERNELS_H
#define PSL_MATCH_KERNELS_H
#include <vector>

#include "pslfe_internal.h"
#include "psl_device_math.h"

#define PSL_GRID_COLS 64
#define PSL_GRID_ROWS 48
#define PSL_GRID_CELLS (PSL_GRID_COLS * PSL_GRID_ROWS)
#define PSL_QMAX 4096      // most queries / keypoints one workgroup handles
#define PSL_TH_HIGH 100    // ORBmatcher::TH_HIGH src/ORBmatcher.cc:37
#define PSL_HISTO 30       // ORBmatcher::HISTO_LENGTH :39

struct FrameMeta {
    int n;
    float minX, minY, invW, invH;
};

struct FrameStore {  // slot s lives at [s * cap] of every array
    PslKeyPoint* kps;
    uint8_t* desc;
    float* uright;
    uint16_t* cellof;
    int* gstart;  // [slot][PSL_GRID_CELLS + 1]
    int* gidx;    // [slot][cap]
    FrameMeta* meta;
    int cap;
};


__device__ __forceinline__ int psl_hamming256(const uint32_t* q, const uint32_t* __restrict__ d) {
    int s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += __popc(q[k] ^ d[k]);
    return s;
}

__device__ __forceinline__ void psl_merge2(uint32_t& k1, uint32_t& k2, uint32_t o1, uint32_t o2) {
    const uint32_t lo = min(k1, o1), hi = max(k1, o1);
    k2 = min(hi, min(k2, o2));
    k1 = lo;
}

struct MatchArgs {
    FrameStore S;
    int slot0;
    const PslProjQuery* q;
    const uint8_t* qdesc;
    const int* nq_arr;
    int nq_single, qstride;
    const uint8_t* taken;
    int check_ori;
    float nnratio;
    int* match;
    int* assigned;
    int* nmatches;
    uint32_t* topk;   // [pair][qstride][PSL_TOPK]: the smallest keys (dist << 16 | CSR position) of every query, ascending
    uint8_t* more;    // [pair][qstride]: the query has more than PSL_TOPK gated candidates
    int th;           // descriptor-distance gate of the decision: TH_HIGH, ORBdist or TH_LOW
    int no_stereo;    // skip the mvuRight gate (SearchByProjection(cur,KF), SearchByBoW)
    const int* fidx;  // != NULL: candidates of a query are the run [min_level, min_level + max_level) of this index list
                      // (the frame's DBoW2 FeatureVector flattened in node order) instead of a grid window
};

#define PSL_KEY_INF 0xffffffffu

__device__ __forceinline__ void psl_top4_insert(uint32_t (&t)[4], uint32_t key) {
    if (key < t[3]) {
        t[3] = key;
        if (t[3] < t[2]) { const uint32_t u = t[2]; t[2] = t[3]; t[3] = u; }
        if (t[2] < t[1]) { const uint32_t u = t[1]; t[1] = t[2]; t[2] = u; }
        if (t[1] < t[0]) { const uint32_t u = t[0]; t[0] = t[1]; t[1] = u; }
    }
}

struct FrameView {
    const PslKeyPoint* kps;
    const uint32_t* desc;
    const float* uright;
    const int* gstart;
    const int* gidx;
    FrameMeta M;
    int n;
};

__device__ __forceinline__ FrameView psl_frame_view(const FrameStore& S, int slot) {
    FrameView V;
    V.M = S.meta[slot];
    V.kps = S.kps + (size_t)slot * S.cap;
    V.desc = reinterpret_cast<const uint32_t*>(S.desc + (size_t)slot * S.cap * 32);
    V.uright = S.uright + (size_t)slot * S.cap;
    V.gstart = S.gstart + (size_t)slot * (PSL_GRID_CELLS + 1);
    V.gidx = S.gidx + (size_t)slot * S.cap;
    V.n = V.M.n < PSL_QMAX ? V.M.n : PSL_QMAX;
    return V;
}

// GetFeaturesInArea window of one query (src/Frame.cc:985-1038), one wave per query.  Grid column ix of the window
// (cells nMinCellY..nMaxCellY) is one contiguous CSR run; lane l fetches the run of column nMinCellX + l and a wave
// scan flattens the runs into candidate numbers 0..T-1 in the reference's visiting order.  Lane l then evaluates
// candidates l, l + 64, ...: every candidate costs the same three dependent fetches (run bounds -> keypoint index
// -> keypoint, descriptor, gates) no matter how many share its column, and all lanes work even for narrow windows.
struct WindowCols {
    int start, excl, incl, T;
    bool checkLevels;
};

__device__ __forceinline__ WindowCols psl_window_cols(const FrameView& V, const PslProjQuery& q, const int* fidx) {
    const int lane = threadIdx.x & 63;
    if (fidx) {  // one run: the frame's features under the query's vocabulary node
        WindowCols W;
        W.start = lane == 0 ? q.min_level : 0;
        const int len = lane == 0 ? (q.max_level > 0 ? q.max_level : 0) : 0;
        W.incl = len > 0 ? len : 0;
        W.incl = __shfl(W.incl, 0);  // inclusive counts: every lane >= 0 holds the total
        W.excl = lane == 0 ? 0 : W.incl;
        W.T = W.incl;
        W.checkLevels = false;
        return W;
    }
    const FrameMeta& M = V.M;
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
        W.start = V.gstart[ix * PSL_GRID_ROWS + minCY];
        len = V.gstart[ix * PSL_GRID_ROWS + maxCY + 1] - W.start;
    }
    int incl = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    W.incl = incl;
    W.excl = incl - len;
    W.T = __shfl(incl, 63);
    W.checkLevels = (q.min_level > 0) || (q.max_level >= 0);
    return W;
}

// CSR position of candidate number j of the window, -1 if j >= T.  Called by all 64 lanes (shuffles inside).
__device__ __forceinline__ int psl_window_pos(const WindowCols& W, int j) {
    int c = 0;  // number of columns whose inclusive count is <= j == the column of candidate j
#pragma unroll
    for (int step = 32; step > 0; step >>= 1) {
        const int v = __shfl(W.incl, c + step - 1);
        if (v <= j) c += step;
    }
    c = c < 63 ? c : 63;
    const int cs = __shfl(W.start, c), ce = __shfl(W.excl, c);
    return j < W.T ? cs + (j - ce) : -1;
}

// Key (distance << 16 | CSR position) of candidate number j, PSL_KEY_INF if j >= T or a gate rejects it: level band,
// window, stereo (:1405-1411), taken initially (:1401-1403), taken by an earlier query of this call (blocker != NULL).
// Called by all 64 lanes (shuffles inside).
__device__ __forceinline__ uint32_t psl_window_key(const FrameView& V, const PslProjQuery& q, const uint32_t* qd, const uint8_t* taken,
                                                   const int* blocker, int qi, const WindowCols& W, int j, const int* fidx, int no_stereo) {
    const int p = psl_window_pos(W, j);
    uint32_t key = PSL_KEY_INF;
    if (p >= 0) {
        const float r = q.radius;
        const int i2 = fidx ? fidx[p] : V.gidx[p];
        const float2 xy = *reinterpret_cast<const float2*>(&V.kps[i2].x);
        const int octave = V.kps[i2].octave;
        const float ur = V.uright[i2];
        const uint4 d0 = *reinterpret_cast<const uint4*>(V.desc + (size_t)i2 * 8);
        const uint4 d1 = *reinterpret_cast<const uint4*>(V.desc + (size_t)i2 * 8 + 4);
        bool ok = i2 >= 0 && i2 < V.n;
        if (!fidx) {
            if (W.checkLevels) ok = ok && !(octave < q.min_level) && !(q.max_level >= 0 && octave > q.max_level);
            ok = ok && (__builtin_fabsf(PSL_FSUB(xy.x, q.u)) < r && __builtin_fabsf(PSL_FSUB(xy.y, q.v)) < r);
        }
        if (taken) ok = ok && !taken[i2];
        if (blocker) ok = ok && !(blocker[i2] < qi);
        if (!no_stereo) ok = ok && !(ur > 0 && __builtin_fabsf(PSL_FSUB(q.ur, ur)) > r);
        const int dist = __popc(qd[0] ^ d0.x) + __popc(qd[1] ^ d0.y) + __popc(qd[2] ^ d0.z) + __popc(qd[3] ^ d0.w) +
                         __popc(qd[4] ^ d1.x) + __popc(qd[5] ^ d1.y) + __popc(qd[6] ^ d1.z) + __popc(qd[7] ^ d1.w);
        if (ok) key = ((uint32_t)dist << 16) | (uint32_t)p;
    }
    return key;
}

// Ascending bitonic sort of one 32-bit key per lane across the wave.
__device__ __forceinline__ uint32_t psl_wave_sort(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const uint32_t o = __shfl_xor(v, j);
            const bool up = (lane & k) == 0, lower = (lane & j) == 0;
            v = (up == lower) ? min(v, o) : max(v, o);
        }
    }
    return v;
}


struct pslfe_frame {
    pslfe_ctx* ctx = nullptr;
    int cap = 0, max_frames = 0;
    FrameStore S = {};
    // scratch for the host-pointer entry points
    PslProjQuery* d_q = nullptr;
    uint8_t* d_qdesc = nullptr;
    uint8_t* d_taken = nullptr;
    int* d_match = nullptr;
    int* d_assigned = nullptr;
    int* d_nm = nullptr;
    uint32_t* d_topk = nullptr;  // [max_frames][cap][PSL_TOPK]
    uint8_t* d_more = nullptr;   // [max_frames][cap]
    uint32_t* d_topk1 = nullptr; // [PSL_QMAX][PSL_TOPK] for the host-pointer entry points
    uint8_t* d_more1 = nullptr;
    int* d_fidx = nullptr;       // [cap] the frame's FeatureVector (SearchByBoW, host-pointer entry point)
    float* d_depth = nullptr;    // [max_frames][cap] mvDepth (RGB-D post-processing)
    float* d_bounds = nullptr;   // [4] scratch for k_image_bounds
    std::vector<char> slot_set;
};

#endif
