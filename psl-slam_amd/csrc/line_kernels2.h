// HIP kernels of the line front-end, part 2: line merging + KeyLine construction, LBD descriptor,
// structural-line (LIL/LJL) pairing.  Product code.  See line_kernels.h for the reference map.
// Convention shared with the oracle: unqualified libm calls on float arguments in the reference
// (atan, atan2, tan, sin, cos) are the float overloads; sinf/cosf are psl_sincosf (bit-identical to
// glibc on [-2pi, 2pi]), atanf / atan2f are psl_atanf / psl_atan2f (glibc's float algorithms restated, pinned against libm);
// tanf is psl_tanf (glibc's float tanf restated, bit-identical for every float in [0, 120), psl_f64math.h); the double sin / cos
// of MergeTwoLines are glibc's table-driven algorithm restated with a regenerated table (psl_sincos_glibc.h, bit-identical to
// libm on 6e7 arguments).  Nothing here calls the device math library.
#ifndef PSL_LINE_KERNELS2_H
#define PSL_LINE_KERNELS2_H

#include "line_kernels.h"

#define PSL_MERGE_NMAX 4096    // most segments MergeLines handles per frame
#define PSL_MERGE_CLMAX 65536  // capacity of the concatenated (sub-)cluster lists
#define PSL_FAN_CAP 4096       // rows of the fans matrix before de-duplication, per frame

struct MergeScratch {  // base pointers; frame f uses base + f * (per-frame size)
    float* lines0;     // [NMAX][4]
    float* lines1;     // [NMAX][4]
    float* merged;     // [NMAX][4]
    float* angles;     // [NMAX]
    float* length;     // [NMAX]
    int* order;        // sorted position -> index
    int* pos;          // index -> sorted position
    uint32_t* adj;     // [NMAX][NMAX/32] over sorted positions
    int* code;         // [NMAX]
    int* clist;        // [CLMAX]
    int* coff;         // [2*NMAX+2]
    int* work;         // [4*NMAX]
    uint32_t* bits;    // [NMAX/32]
    PslKeyLine* stage; // [NMAX]
};

__device__ __forceinline__ float psl_point_line_distance(const float* l, float x0, float y0) {  // uselongline.cpp:5-15
    const float x1 = l[0], y1 = l[1], x2 = l[2], y2 = l[3];
    const float numf = __builtin_fabsf(PSL_FADD(PSL_FADD(PSL_FMUL(PSL_FSUB(y2, y1), x0), PSL_FMUL(PSL_FSUB(x1, x2), y0)),
                                                PSL_FSUB(PSL_FMUL(x2, y1), PSL_FMUL(x1, y2))));
    const double a = (double)PSL_FSUB(y2, y1), b = (double)PSL_FSUB(x1, x2);  // std::pow(float, 2) promotes to double
    const double den = __dsqrt_rn(PSL_DADD(PSL_DMUL(a, a), PSL_DMUL(b, b)));
    return (float)((double)numf / den);
}

__device__ __forceinline__ float psl_angle_diff_f(float a1, float a2) {  // :17-22
    const float c1 = __builtin_fabsf(PSL_FSUB(a2, a1));
    const float c2 = (float)PSL_DSUB(PSL_DADD(PSL_PI, (double)fminf(a1, a2)), (double)fmaxf(a1, a2));
    return fminf(c1, c2);
}

// pair test of MergeLines (:61-156); idx1 is the line at the earlier sorted position
__device__ bool psl_merge_pair(const float* src, const float* angles, int idx1, int idx2, float angle_thr, float distance_thr, float ep_thr) {
    float x11 = src[4 * idx1], y11 = src[4 * idx1 + 1], x12 = src[4 * idx1 + 2], y12 = src[4 * idx1 + 3];
    const float angle1 = angles[idx1];
    const bool to_sort_x = __builtin_fabsf(angle1) < (float)(PSL_PI / 4.0);
    if ((to_sort_x && (x12 < x11)) || ((!to_sort_x) && y12 < y11)) { float t = x11; x11 = x12; x12 = t; t = y11; y11 = y12; y12 = t; }
    float x21 = src[4 * idx2], y21 = src[4 * idx2 + 1], x22 = src[4 * idx2 + 2], y22 = src[4 * idx2 + 3];
    if ((to_sort_x && (x22 < x21)) || ((!to_sort_x) && y22 < y21)) { float t = x21; x21 = x22; x22 = t; t = y21; y21 = y22; y22 = t; }
    if (psl_angle_diff_f(angle1, angles[idx2]) > angle_thr) return false;
    const float mid_x1 = (float)PSL_DMUL(0.5, (double)PSL_FADD(src[4 * idx1], src[4 * idx1 + 2]));
    const float mid_y1 = (float)PSL_DMUL(0.5, (double)PSL_FADD(src[4 * idx1 + 1], src[4 * idx1 + 3]));
    const float mid_x2 = (float)PSL_DMUL(0.5, (double)PSL_FADD(src[4 * idx2], src[4 * idx2 + 2]));
    const float mid_y2 = (float)PSL_DMUL(0.5, (double)PSL_FADD(src[4 * idx2 + 1], src[4 * idx2 + 3]));
    const float m1 = psl_point_line_distance(&src[4 * idx2], mid_x1, mid_y1);
    const float m2 = psl_point_line_distance(&src[4 * idx1], mid_x2, mid_y2);
    if (m1 > distance_thr && m2 > distance_thr) return false;
    float cx12, cy12, cx21, cy21;
    if ((to_sort_x && x12 > x22) || (!to_sort_x && y12 > y22)) { cx12 = x22; cy12 = y22; cx21 = x11; cy21 = y11; }
    else { cx12 = x12; cy12 = y12; cx21 = x21; cy21 = y21; }
    bool to_merge = ((to_sort_x && cx12 >= cx21) || (!to_sort_x && cy12 >= cy21));
    if (!to_merge) {
        const float ex = PSL_FSUB(cx21, cx12), ey = PSL_FSUB(cy21, cy12);
        to_merge = PSL_FADD(PSL_FMUL(ex, ex), PSL_FMUL(ey, ey)) < ep_thr;
    }
    return to_merge;
}

__device__ void psl_merge_two_lines(const float* l1, const float* l2, float* out, const double* sctab) {  // :266-334
    const float ax = l1[0], ay = l1[1], bx = l1[2], by = l1[3], cx = l2[0], cy = l2[1], dx = l2[2], dy = l2[3];
    const float dlix = PSL_FSUB(bx, ax), dliy = PSL_FSUB(by, ay), dljx = PSL_FSUB(dx, cx), dljy = PSL_FSUB(dy, cy);
    const double li = __dsqrt_rn(PSL_DADD((double)PSL_FMUL(dlix, dlix), (double)PSL_FMUL(dliy, dliy)));
    const double lj = __dsqrt_rn(PSL_DADD((double)PSL_FMUL(dljx, dljx), (double)PSL_FMUL(dljy, dljy)));
    const double den = PSL_DMUL(2.0, PSL_DADD(li, lj));
    const double xg = PSL_DADD(PSL_DMUL(li, (double)PSL_FADD(ax, bx)), PSL_DMUL(lj, (double)PSL_FADD(cx, dx))) / den;
    const double yg = PSL_DADD(PSL_DMUL(li, (double)PSL_FADD(ay, by)), PSL_DMUL(lj, (double)PSL_FADD(cy, dy))) / den;
    const double thi = dlix == 0.0f ? PSL_PI / 2.0 : (double)psl_atanf(PSL_FDIV(dliy, dlix));
    const double thj = dljx == 0.0f ? PSL_PI / 2.0 : (double)psl_atanf(PSL_FDIV(dljy, dljx));
    double thr;
    if (fabs(PSL_DSUB(thi, thj)) <= PSL_PI / 2.0) thr = PSL_DADD(PSL_DMUL(li, thi), PSL_DMUL(lj, thj)) / PSL_DADD(li, lj);
    else {
        const double tmp = PSL_DSUB(thj, PSL_DMUL(PSL_PI, thj / fabs(thj)));
        thr = PSL_DADD(PSL_DMUL(li, thi), PSL_DMUL(lj, tmp));
        thr = thr / PSL_DADD(li, lj);
    }
    // glibc's double sin / cos, bit for bit (psl_sincos_glibc.h): d1 * s + yg can cancel (an end point on the image border)
    const double s = psl_glibc_sin(thr, sctab), c = psl_glibc_cos(thr, sctab);
    const double axg = PSL_DADD(PSL_DMUL(PSL_DSUB((double)ay, yg), s), PSL_DMUL(PSL_DSUB((double)ax, xg), c));
    const double bxg = PSL_DADD(PSL_DMUL(PSL_DSUB((double)by, yg), s), PSL_DMUL(PSL_DSUB((double)bx, xg), c));
    const double cxg = PSL_DADD(PSL_DMUL(PSL_DSUB((double)cy, yg), s), PSL_DMUL(PSL_DSUB((double)cx, xg), c));
    const double dxg = PSL_DADD(PSL_DMUL(PSL_DSUB((double)dy, yg), s), PSL_DMUL(PSL_DSUB((double)dx, xg), c));
    const double d1 = fmin(axg, fmin(bxg, fmin(cxg, dxg))), d2 = fmax(axg, fmax(bxg, fmax(cxg, dxg)));
    out[0] = (float)PSL_DADD(PSL_DMUL(d1, c), xg); out[1] = (float)PSL_DADD(PSL_DMUL(d1, s), yg);
    out[2] = (float)PSL_DADD(PSL_DMUL(d2, c), xg); out[3] = (float)PSL_DADD(PSL_DMUL(d2, s), yg);
}

// One MergeLines pass (src[0..n) -> M.merged) followed by FilterShortLines(length_thr) into dst.
// Called by all 256 threads; returns the new count (uniform).  s_i[1] accumulates the overflow flag.
// The clustering below is the reference's serial algorithm, run by one thread; with its working set in HBM every step
// is a dependent ~1 us access (85 % of the kernel's wave time was s_waitcnt).  For n <= PSL_MERGE_LDSN (always, at
// 640x480) the per-line arrays it chases live in LDS: cluster code, sort order / position, length, the two BFS
// frontiers, the sub-cluster marks.  Larger inputs use the HBM arrays (same code through flat pointers).
// The kernel exists for two sizes of that LDS working set: 512 lines (15 KB: 8 workgroups per CU; what a 640x480 frame
// usually has) and 1024 lines (30 KB: 5 per CU).  Both are launched; a workgroup whose frame belongs to the other one exits.
#define PSL_MERGE_LDSN 1024
#define PSL_MERGE_LDSN_SMALL 512
__device__ __forceinline__ int psl_block_excl_scan256(int v, int* s_w, int* total);
template <int LN>
struct MergeLds {
    int code[LN], order[LN], pos[LN], tc[LN], nx[LN], loc[LN];
    float length[LN];
    uint32_t bits[LN / 32];
    uint8_t clustered[LN];
    __attribute__((aligned(16))) uint32_t row[PSL_MERGE_NMAX / 32];  // the adjacency row being scanned by the serial clustering
};

// One adjacency row HBM -> LDS with the 16-byte loads of up to 8 quads in flight: the serial scan below would otherwise
// pay one dependent round trip per 32-pair word.
template <int LN>
__device__ __forceinline__ const uint32_t* psl_merge_row(MergeLds<LN>& LD, const uint32_t* row, int words) {
    const uint4* r4 = reinterpret_cast<const uint4*>(row);
    uint4* d4 = reinterpret_cast<uint4*>(LD.row);
    const int nq = (words + 3) >> 2;
    for (int base = 0; base < nq; base += 8) {
        uint4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = base + u < nq ? r4[base + u] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (base + u < nq) d4[base + u] = t[u];
    }
    return LD.row;
}

template <int LN>
__device__ int psl_merge_pass(const MergeScratch& M, MergeLds<LN>& LD, const float* src, float* dst, int n, float angle_threshold,
                              float distance_threshold, float endpoint_threshold, float length_thr, int* s_i, const double* sctab) {
    const int tid = threadIdx.x, BS = 256;
    if (n <= 0) return 0;
    const bool small = n <= LN;
    const int words = (n + 31) >> 5;
    const int ROW = PSL_MERGE_NMAX / 32;
    for (int i = tid; i < n; i += BS) {
        const float dx = PSL_FSUB(src[4 * i + 2], src[4 * i]), dy = PSL_FSUB(src[4 * i + 3], src[4 * i + 1]);
        M.angles[i] = psl_atanf(PSL_FDIV(dy, dx));  // Eigen ArrayXf::atan()
        const float len = sqrtf(PSL_FADD(PSL_FMUL(dx, dx), PSL_FMUL(dy, dy)));
        M.length[i] = len;
        M.code[i] = -1;
        if (small) { LD.length[i] = len; LD.code[i] = -1; }
    }
    __syncthreads();
    for (int i = tid; i < n; i += BS) {  // std::sort of indices by angle (stable convention H16): rank by counting
        const float a = M.angles[i];
        int r = 0;
        for (int j = 0; j < n; ++j) { const float b = M.angles[j]; r += (b < a) || (b == a && j < i); }
        M.pos[i] = r;
        M.order[r] = i;
        if (small) { LD.pos[i] = r; LD.order[r] = i; }
    }
    __syncthreads();
    const float ep_thr = PSL_FMUL(endpoint_threshold, endpoint_threshold);
    if (small) {
        // Adjacency over sorted positions, banded.  psl_merge_pair starts with the angle test, and the lines are sorted by
        // angle, so for a row only the positions whose angle lies within the threshold - around the row's own angle, or
        // across the +-pi/2 seam - can be set: three contiguous runs.  A row's candidate 32-pair words are listed in a
        // bit mask, the (row, word) items are flattened by a prefix sum and evaluated densely; all other words are zero.
        // (All n^2 pair tests, 97 % of them failing the angle test one lane at a time, were the kernel's main cost.)
        float* sa = reinterpret_cast<float*>(LD.loc);  // angle by sorted position
        int* wm = LD.tc;                               // candidate words of a row (bit w), n <= 1024 -> <= 32 words
        int* pw = LD.nx;                               // exclusive prefix of popc(wm) over the rows
        for (int p = tid; p < n; p += BS) sa[p] = M.angles[LD.order[p]];
        for (int t = tid; t < n * words; t += BS) M.adj[(size_t)(t / words) * ROW + (t % words)] = 0;
        __syncthreads();
        const float tband = angle_threshold + 1e-3f;  // superset of the exact test
        auto lower = [&](float v) { int lo = 0, hi = n; while (lo < hi) { const int mid = (lo + hi) >> 1; if (sa[mid] < v) lo = mid + 1; else hi = mid; } return lo; };   // first p with sa[p] >= v
        auto upper = [&](float v) { int lo = 0, hi = n; while (lo < hi) { const int mid = (lo + hi) >> 1; if (sa[mid] <= v) lo = mid + 1; else hi = mid; } return lo; };  // first p with sa[p] > v
        auto span = [](int lo, int hi) -> uint32_t {  // bits lo >> 5 .. hi >> 5
            if (lo > hi) return 0u;
            const int a = lo >> 5, b = hi >> 5;
            const uint32_t upto = b >= 31 ? 0xffffffffu : ((1u << (b + 1)) - 1u);
            return upto & ~((1u << a) - 1u);
        };
        __shared__ int s_tot, s_scan[4];
        if (tid == 0) s_tot = 0;
        __syncthreads();
        for (int base = 0; base < n; base += BS) {
            const int pi = base + tid;
            uint32_t m = 0;
            if (pi < n) {
                const float a = sa[pi];
                m = span(lower(a - tband), upper(a + tband) - 1);
                m |= span(lower(a + (float)PSL_PI - tband), n - 1);
                m |= span(0, upper(a - (float)PSL_PI + tband) - 1);
                wm[pi] = (int)m;
            }
            int tot;
            const int ex = psl_block_excl_scan256(__popc(m), s_scan, &tot);
            if (pi < n) pw[pi] = s_tot + ex;
            __syncthreads();
            if (tid == 0) s_tot += tot;
            __syncthreads();
        }
        const int T = s_tot;
        for (int k = tid; k < T; k += BS) {
            int lo = 0, hi = n;  // row = last pi with pw[pi] <= k
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (pw[mid] <= k) lo = mid + 1; else hi = mid; }
            const int pi = lo - 1;
            uint32_t m = (uint32_t)wm[pi];
            for (int r = k - pw[pi]; r > 0; --r) m &= m - 1;
            const int wj = __ffs(m) - 1;
            uint32_t bw = 0;
            const int idxi = LD.order[pi];
            for (int b = 0; b < 32; ++b) {
                const int pj = wj * 32 + b;
                if (pj >= n || pj == pi) continue;
                const int idxj = LD.order[pj];
                const bool mm = pi < pj ? psl_merge_pair(src, M.angles, idxi, idxj, angle_threshold, distance_threshold, ep_thr)
                                        : psl_merge_pair(src, M.angles, idxj, idxi, angle_threshold, distance_threshold, ep_thr);
                bw |= (uint32_t)mm << b;
            }
            M.adj[(size_t)pi * ROW + wj] = bw;
        }
    } else {
        for (int t = tid; t < n * words; t += BS) {  // adjacency over sorted positions, one 32-pair word per step
            const int pi = t / words, wj = t - pi * words;
            uint32_t bw = 0;
            const int idxi = M.order[pi];
            for (int b = 0; b < 32; ++b) {
                const int pj = wj * 32 + b;
                if (pj >= n || pj == pi) continue;
                const int idxj = M.order[pj];
                const bool m = pi < pj ? psl_merge_pair(src, M.angles, idxi, idxj, angle_threshold, distance_threshold, ep_thr)
                                       : psl_merge_pair(src, M.angles, idxj, idxi, angle_threshold, distance_threshold, ep_thr);
                bw |= (uint32_t)m << b;
            }
            M.adj[(size_t)pi * ROW + wj] = bw;
        }
    }
    __syncthreads();
    // clustering (:159-188).  The reference walks, per cluster, a breadth-first frontier: members of `to_check` are coded and
    // appended in list order, their still-uncoded neighbours form the next frontier (a std::set: ascending index).  Small
    // inputs run it on one wave: lane = frontier member (or adjacency word); a neighbour k is "still uncoded" when member
    // q is processed iff it was uncoded at the start of the round and is not itself a member at a position <= q, which
    // makes every member independent of the others; appends are ordered compactions.  Larger inputs: serial, literally.
    __shared__ int s_cl[3];  // ncl, total, overflow
    if (small) {
        if (tid < 64) {
            const int lane = tid;
            const unsigned long long ltm = (1ull << lane) - 1ull;
            int* code = LD.code;
            const int* order = LD.order;
            const int* posv = LD.pos;
            uint32_t* bits = LD.bits;
            int* to_check = LD.tc;
            int* next = LD.nx;
            int* qpos = LD.loc;
            for (int k = lane; k < n; k += 64) qpos[k] = 0x7fffffff;
            if (lane == 0) M.coff[0] = 0;
            __builtin_amdgcn_wave_barrier();
            int ncl = 0, total = 0;
            bool overflow = false;
            for (int i = 0; i < n && !overflow; ++i) {
                if (code[i] >= 0) continue;
                const int new_code = ncl;
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) code[i] = new_code;
                int ntc;
                {   // neighbours of i in sorted-position order
                    uint32_t v = lane < words ? M.adj[(size_t)posv[i] * ROW + lane] : 0u;
                    const int c = __popc(v);
                    int inc = c;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
                    ntc = __shfl(inc, 63);
                    int o = inc - c;
                    while (v) { const int bb = __ffs(v) - 1; v &= v - 1; to_check[o++] = order[lane * 32 + bb]; }
                }
                if (total < PSL_MERGE_CLMAX) { if (lane == 0) M.clist[total] = i; ++total; } else overflow = true;
                __builtin_amdgcn_wave_barrier();
                while (ntc > 0 && !overflow) {
                    if (lane < words) bits[lane] = 0;
                    for (int q = lane; q < ntc; q += 64) qpos[to_check[q]] = q;
                    __builtin_amdgcn_wave_barrier();
                    for (int base = 0; base < ntc; base += 64) {
                        const int q = base + lane;
                        const int j = q < ntc ? to_check[q] : 0;
                        const bool unc = q < ntc && code[j] < 0;
                        const unsigned long long mu = __ballot(unc);
                        const int at = total + __popcll(mu & ltm);
                        if (unc && at < PSL_MERGE_CLMAX) M.clist[at] = j;
                        total += __popcll(mu);
                        if (total > PSL_MERGE_CLMAX) { total = PSL_MERGE_CLMAX; overflow = true; }
                        if (q < ntc) {
                            const uint32_t* row = M.adj + (size_t)posv[j] * ROW;
                            for (int w = 0; w < words; ++w) {
                                uint32_t v = row[w];
                                while (v) {
                                    const int bb = __ffs(v) - 1; v &= v - 1;
                                    const int k = order[w * 32 + bb];
                                    if (code[k] < 0 && !(qpos[k] <= q)) atomicOr(&bits[k >> 5], 1u << (k & 31));
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    for (int q = lane; q < ntc; q += 64) {
                        const int j = to_check[q];
                        if (code[j] < 0) code[j] = new_code;
                        qpos[j] = 0x7fffffff;
                    }
                    __builtin_amdgcn_wave_barrier();
                    {   // next frontier = set bits, ascending
                        uint32_t v = lane < words ? bits[lane] : 0u;
                        const int c = __popc(v);
                        int inc = c;
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
                        ntc = __shfl(inc, 63);
                        int o = inc - c;
                        while (v) { const int bb = __ffs(v) - 1; v &= v - 1; next[o++] = lane * 32 + bb; }
                    }
                    int* t = to_check; to_check = next; next = t;
                    __builtin_amdgcn_wave_barrier();
                }
                ++ncl;
                if (lane == 0) M.coff[ncl] = total;
            }
            if (lane == 0) { s_cl[0] = ncl; s_cl[1] = total; s_cl[2] = overflow ? 1 : 0; }
        }
    } else if (tid == 0) {
        int ncl = 0, total = 0;
        int* code = M.code;
        const int* order = M.order;
        const int* posv = M.pos;
        uint32_t* bits = M.bits;
        int* to_check = M.work;
        int* next = M.work + PSL_MERGE_NMAX;
        bool overflow = false;
        M.coff[0] = 0;
        for (int i = 0; i < n && !overflow; ++i) {
            if (code[i] >= 0) continue;
            const int new_code = ncl;
            code[i] = new_code;
            int ntc = 0;
            {
                const uint32_t* row = psl_merge_row(LD, M.adj + (size_t)posv[i] * ROW, words);
                for (int w = 0; w < words; ++w) { uint32_t v = row[w]; while (v) { const int b = __ffs(v) - 1; v &= v - 1; to_check[ntc++] = order[w * 32 + b]; } }
            }
            if (total < PSL_MERGE_CLMAX) M.clist[total++] = i; else overflow = true;
            while (ntc > 0 && !overflow) {
                for (int w = 0; w < words; ++w) bits[w] = 0;
                for (int q = 0; q < ntc; ++q) {
                    const int j = to_check[q];
                    if (code[j] < 0) { code[j] = new_code; if (total < PSL_MERGE_CLMAX) M.clist[total++] = j; else overflow = true; }
                    const uint32_t* row = psl_merge_row(LD, M.adj + (size_t)posv[j] * ROW, words);
                    for (int w = 0; w < words; ++w) {
                        uint32_t v = row[w];
                        while (v) { const int b = __ffs(v) - 1; v &= v - 1; const int k = order[w * 32 + b]; if (code[k] < 0) bits[k >> 5] |= 1u << (k & 31); }
                    }
                }
                ntc = 0;
                for (int w = 0; w < words; ++w) { uint32_t v = bits[w]; while (v) { const int b = __ffs(v) - 1; v &= v - 1; next[ntc++] = w * 32 + b; } }
                int* t = to_check; to_check = next; next = t;
            }
            M.coff[++ncl] = total;
        }
        s_cl[0] = ncl; s_cl[1] = total; s_cl[2] = overflow ? 1 : 0;
    }
    __syncthreads();
    if (tid == 0) {
        int ncl = s_cl[0], total = s_cl[1];
        bool overflow = s_cl[2] != 0;
        const int* order = small ? LD.order : M.order;
        const int* posv = small ? LD.pos : M.pos;
        const float* length = small ? LD.length : M.length;
        // sub-clusters (:190-228), appended behind the raw clusters
        const int raw_ncl = ncl;
        int* loc = small ? LD.loc : M.work + 2 * PSL_MERGE_NMAX;
        uint8_t* clustered = small ? LD.clustered : reinterpret_cast<uint8_t*>(M.work + 3 * PSL_MERGE_NMAX);
        int nfinal = 0;
        int* foff = M.coff + PSL_MERGE_NMAX + 1;
        foff[0] = total;
        for (int c = 0; c < raw_ncl && !overflow; ++c) {
            const int cs = M.coff[c + 1] - M.coff[c];
            int* cl = M.clist + M.coff[c];
            if (cs <= 2) {
                for (int q = 0; q < cs; ++q) { if (total < PSL_MERGE_CLMAX) M.clist[total++] = cl[q]; else overflow = true; }
                foff[++nfinal] = total;
                continue;
            }
            for (int a = 1; a < cs; ++a) {  // sort by length, descending (stable)
                const int v = cl[a];
                const float lv = length[v];
                int b = a - 1;
                while (b >= 0 && length[cl[b]] < lv) { cl[b + 1] = cl[b]; --b; }
                cl[b + 1] = v;
            }
            for (int q = 0; q < cs; ++q) { loc[cl[q]] = q; clustered[q] = 0; }
            for (int j = 0; j < cs && !overflow; ++j) {
                if (clustered[j]) continue;
                const int line_idx = cl[j];
                if (total < PSL_MERGE_CLMAX) M.clist[total++] = line_idx; else overflow = true;
                const uint32_t* row = psl_merge_row(LD, M.adj + (size_t)posv[line_idx] * ROW, words);
                for (int w = 0; w < words; ++w) {
                    uint32_t v = row[w];
                    while (v) {
                        const int b = __ffs(v) - 1; v &= v - 1;
                        const int k = order[w * 32 + b];
                        clustered[loc[k]] = 1;
                        if (total < PSL_MERGE_CLMAX) M.clist[total++] = k; else overflow = true;
                    }
                }
                foff[++nfinal] = total;
            }
        }
        s_i[0] = nfinal;
        if (overflow) s_i[1] = 1;
    }
    __syncthreads();
    const int nfinal = s_i[0];
    const int* foff = M.coff + PSL_MERGE_NMAX + 1;
    for (int c = tid; c < nfinal; c += BS) {  // merge chains (:230-262); the first line is merged with itself first
        const int* cl = M.clist + foff[c];
        const int cs = foff[c + 1] - foff[c];
        float nl[4] = {src[4 * cl[0]], src[4 * cl[0] + 1], src[4 * cl[0] + 2], src[4 * cl[0] + 3]};
        for (int q = 0; q < cs; ++q) {
            float o[4];
            psl_merge_two_lines(nl, &src[4 * cl[q]], o, sctab);
            nl[0] = o[0]; nl[1] = o[1]; nl[2] = o[2]; nl[3] = o[3];
        }
        M.merged[4 * c] = nl[0]; M.merged[4 * c + 1] = nl[1]; M.merged[4 * c + 2] = nl[2]; M.merged[4 * c + 3] = nl[3];
    }
    __syncthreads();
    if (tid == 0) {  // FilterShortLines (:338-351), order preserving
        const float thr2 = PSL_FMUL(length_thr, length_thr);
        int keep = 0;
        for (int c = 0; c < nfinal; ++c) {
            const float* m = M.merged + 4 * c;
            const float dx = PSL_FSUB(m[2], m[0]), dy = PSL_FSUB(m[3], m[1]);
            if (PSL_FADD(PSL_FMUL(dx, dx), PSL_FMUL(dy, dy)) > thr2) {
                dst[4 * keep] = m[0]; dst[4 * keep + 1] = m[1]; dst[4 * keep + 2] = m[2]; dst[4 * keep + 3] = m[3];
                ++keep;
            }
        }
        s_i[0] = keep;
    }
    __syncthreads();
    const int out_n = s_i[0];
    __syncthreads();
    return out_n;
}

// cv::clipLine (64-bit integer arithmetic) + LineIterator(8-connected).count (Appendix A.8)
__device__ int psl_line_iterator_count(int w, int h, float fx1, float fy1, float fx2, float fy2) {
    long long x1 = psl_cvround_f(fx1), y1 = psl_cvround_f(fy1), x2 = psl_cvround_f(fx2), y2 = psl_cvround_f(fy2);
    if ((unsigned long long)x1 >= (unsigned long long)w || (unsigned long long)x2 >= (unsigned long long)w ||
        (unsigned long long)y1 >= (unsigned long long)h || (unsigned long long)y2 >= (unsigned long long)h) {
        const long long right = w - 1, bottom = h - 1;
        int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
        int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            long long a;
            if (c1 & 12) { a = c1 < 8 ? 0 : bottom; x1 += (a - y1) * (x2 - x1) / (y2 - y1); y1 = a; c1 = (x1 < 0) + (x1 > right) * 2; }
            if (c2 & 12) { a = c2 < 8 ? 0 : bottom; x2 += (a - y2) * (x2 - x1) / (y2 - y1); y2 = a; c2 = (x2 < 0) + (x2 > right) * 2; }
            if ((c1 & c2) == 0 && (c1 | c2) != 0) {
                if (c1) { a = c1 == 1 ? 0 : right; y1 += (a - x1) * (y2 - y1) / (x2 - x1); x1 = a; c1 = 0; }
                if (c2) { a = c2 == 1 ? 0 : right; y2 += (a - x2) * (y2 - y1) / (x2 - x1); x2 = a; c2 = 0; }
            }
        }
        if ((c1 | c2) != 0) return 0;
    }
    long long dx = x2 - x1, dy = y2 - y1;
    dx = dx < 0 ? -dx : dx; dy = dy < 0 ? -dy : dy;
    return (int)(dx > dy ? dx : dy) + 1;
}

// optimizeAndMergeLines_lsd + top-N + line equations: one workgroup per frame.
// status bits: 1 = more than NMAX raw segments (truncated), 2 = cluster list overflow, 4 = more
// than maxkl merged lines (truncated).
template <int LN>
__global__ __launch_bounds__(256, LN <= PSL_MERGE_LDSN_SMALL ? 8 : 5) void k_line_merge(LineParams P, MergeScratch M0, const float* __restrict__ seg, const int* __restrict__ nseg,
                                                     PslKeyLine* __restrict__ kls, double* __restrict__ lineEq, int* __restrict__ nkl,
                                                     int* __restrict__ status) {
    __shared__ int s_i[4];
    __shared__ MergeLds<LN> LD;
    const int frame = blockIdx.x, tid = threadIdx.x;
    if ((nseg[frame] <= PSL_MERGE_LDSN_SMALL) != (LN == PSL_MERGE_LDSN_SMALL)) return;  // the other instance's frame
    const size_t f = (size_t)frame;
    MergeScratch M;
    M.lines0 = M0.lines0 + f * PSL_MERGE_NMAX * 4; M.lines1 = M0.lines1 + f * PSL_MERGE_NMAX * 4;
    M.merged = M0.merged + f * PSL_MERGE_NMAX * 4;
    M.angles = M0.angles + f * PSL_MERGE_NMAX; M.length = M0.length + f * PSL_MERGE_NMAX;
    M.order = M0.order + f * PSL_MERGE_NMAX; M.pos = M0.pos + f * PSL_MERGE_NMAX;
    M.adj = M0.adj + f * (size_t)PSL_MERGE_NMAX * (PSL_MERGE_NMAX / 32);
    M.code = M0.code + f * PSL_MERGE_NMAX; M.clist = M0.clist + f * PSL_MERGE_CLMAX;
    M.coff = M0.coff + f * (2 * PSL_MERGE_NMAX + 2); M.work = M0.work + f * 4 * PSL_MERGE_NMAX;
    M.bits = M0.bits + f * (PSL_MERGE_NMAX / 32); M.stage = M0.stage + f * PSL_MERGE_NMAX;
    if (tid == 0) s_i[1] = 0;
    int n = nseg[frame];
    int st = 0;
    if (n > PSL_MERGE_NMAX) { n = PSL_MERGE_NMAX; st |= 1; }
    const float* src = seg + f * P.maxseg * 4;
    for (int i = tid; i < n * 4; i += 256) M.lines0[i] = src[i];
    __syncthreads();
    n = psl_merge_pass(M, LD, M.lines0, M.lines1, n, 0.05f, 5.f, 15.f, 30.f, s_i, P.sctab);
    n = psl_merge_pass(M, LD, M.lines1, M.lines0, n, 0.03f, 3.f, 30.f, 50.f, s_i, P.sctab);
    st |= s_i[1] << 1;
    // convertVec4fToKeyLine (:411-447)
    const float* L = M.lines0;
    PslKeyLine* out = kls + f * P.maxkl;
    float* resp = M.angles;
    for (int i = tid; i < n; i += 256) {
        const float* l = L + 4 * i;
        PslKeyLine kl;
        kl.startPointX = l[0]; kl.startPointY = l[1]; kl.endPointX = l[2]; kl.endPointY = l[3];  // * octaveScale (1.0): exact
        kl.sPointInOctaveX = l[0]; kl.sPointInOctaveY = l[1]; kl.ePointInOctaveX = l[2]; kl.ePointInOctaveY = l[3];
        const double ex = (double)PSL_FSUB(l[0], l[2]), ey = (double)PSL_FSUB(l[1], l[3]);
        kl.lineLength = (float)__dsqrt_rn(PSL_DADD(PSL_DMUL(ex, ex), PSL_DMUL(ey, ey)));
        kl.angle = psl_atan2f(PSL_FSUB(kl.endPointY, kl.startPointY), PSL_FSUB(kl.endPointX, kl.startPointX));
        kl.class_id = i;
        kl.octave = 0;
        kl.size = PSL_FMUL(PSL_FSUB(kl.endPointX, kl.startPointX), PSL_FSUB(kl.endPointY, kl.startPointY));
        kl.pt_x = PSL_FADD(kl.endPointX, kl.startPointX) / 2;
        kl.pt_y = PSL_FADD(kl.endPointY, kl.startPointY) / 2;
        kl.response = PSL_FDIV(kl.lineLength, (float)(P.w > P.h ? P.w : P.h));
        kl.numOfPixels = psl_line_iterator_count(P.w, P.h, l[0], l[1], l[2], l[3]);
        resp[i] = kl.response;
        M.stage[i] = kl;
    }
    __syncthreads();
    int m = n;
    if (n > P.nfeatures) {  // sort by response, descending (stable), keep nLSDFeature, class_id = rank (LineExtractor.cpp:342-348)
        m = P.nfeatures;
        for (int i = tid; i < n; i += 256) {
            const float r = resp[i];
            int rank = 0;
            for (int j = 0; j < n; ++j) { const float q = resp[j]; rank += (q > r) || (q == r && j < i); }
            if (rank < m) { PslKeyLine kl = M.stage[i]; kl.class_id = rank; out[rank] = kl; }
        }
    } else {
        if (m > P.maxkl) { m = P.maxkl; st |= 4; }
        for (int i = tid; i < m; i += 256) out[i] = M.stage[i];
    }
    __syncthreads();
    double* eq = lineEq + f * P.maxkl * 3;
    for (int i = tid; i < m; i += 256) {  // lineV = sp x ep / |lineV.xy| (LineExtractor.cpp:352-363), double
        const double sx = out[i].startPointX, sy = out[i].startPointY, ex = out[i].endPointX, ey = out[i].endPointY;
        const double lx = PSL_DSUB(sy, ey), ly = PSL_DSUB(ex, sx), lz = PSL_DSUB(PSL_DMUL(sx, ey), PSL_DMUL(sy, ex));
        const double nrm = __dsqrt_rn(PSL_DADD(PSL_DMUL(lx, lx), PSL_DMUL(ly, ly)));
        eq[3 * i] = lx / nrm; eq[3 * i + 1] = ly / nrm; eq[3 * i + 2] = lz / nrm;
    }
    if (tid == 0) { nkl[frame] = m; status[frame] = st; }
}

// ---------------------------------------------------------------------------------------------
// LBD pre-processing (binary_descriptor_custom.cpp:351-399): GaussianBlur 5x5 sigma 1 on 8U (integer
// kernel, sum 257) then Sobel 3x3 -> s16 dx, dy, both BORDER_REFLECT_101.
// ---------------------------------------------------------------------------------------------
// Fused and tiled: a 64 x 32 output tile needs 66 x 34 blurred samples, i.e. 70 x 38 input pixels staged in
// LDS.  Blur values in the 1-px halo of the image border are computed from reflected input, which equals
// reflecting the blurred image because the kernel is symmetric.  Output: interleaved (dx, dy) as short2.
__global__ __launch_bounds__(256) void k_lbd_pre(LineParams P, const uint8_t* __restrict__ gray, int stride, size_t fstride,
                                                  short2* __restrict__ dxy, int nframes, int xcd) {
    __shared__ __attribute__((aligned(16))) uint8_t s_in[38 * 72 + 16];
    __shared__ __attribute__((aligned(16))) uint16_t s_row[38 * 68 + 8];
    __shared__ __attribute__((aligned(16))) uint8_t s_bl[34 * 68 + 8];
    const int tid = threadIdx.x;
    int tx, ty, frame;
    if (!psl_tile_frame((P.w + 63) / 64, nframes, xcd, &tx, &ty, &frame)) return;
    const int x0 = tx * 64, y0 = ty * 32;
    const uint8_t* img = gray + (size_t)frame * fstride;
    // s_in[r][c] = pixel (x0 - 4 + c, y0 - 3 + r), c in [0, 72): columns x0-3 .. x0+66 are needed (c = 1 .. 70)
    const bool fast = x0 >= 4 && x0 + 68 <= P.w && ((reinterpret_cast<uintptr_t>(img) | (uintptr_t)stride) & 3) == 0;
    if (fast) {
        for (int k = tid; k < 38 * 18; k += 256) {
            const int r = k / 18, c4 = k - r * 18;
            const int gy = psl_reflect101i(y0 + r - 3, P.h);
            reinterpret_cast<uint32_t*>(s_in)[r * 18 + c4] = *reinterpret_cast<const uint32_t*>(img + (size_t)gy * stride + x0 - 4 + c4 * 4);
        }
    } else {
        for (int k = tid; k < 38 * 72; k += 256) {
            const int r = k / 72, c = k - r * 72;
            s_in[r * 72 + c] = img[(size_t)psl_reflect101i(y0 + r - 3, P.h) * stride + psl_reflect101i(x0 - 4 + c, P.w)];
        }
    }
    __syncthreads();
    // The kernel was LDS-bound with one byte / u16 read per tap.  Same integer arithmetic, four adjacent outputs per
    // thread from aligned dwords: the row pass in packed 16-bit math (row sums fit 16 bits: 255 * 257), the column
    // pass with v_dot2_u32_u16 on vertically paired row sums, Sobel from six dwords per four pixels.
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const uint32_t K0 = (uint32_t)P.lbdK[0], K1 = (uint32_t)P.lbdK[1], K2 = (uint32_t)P.lbdK[2];
    {
        const u16x2 k0 = __builtin_bit_cast(u16x2, K0 * 0x10001u), k1 = __builtin_bit_cast(u16x2, K1 * 0x10001u), k2 = __builtin_bit_cast(u16x2, K2 * 0x10001u);
        // row pass at blurred columns x0-1 .. x0+64 (index j = 0..65; groups of 4, the last group's extra columns unused):
        // blurred column j reads input bytes (j+1)..(j+5) of the row
        for (int k = tid; k < 38 * 17; k += 256) {
            const int r = k / 17, g = k - r * 17;
            const uint32_t* in32 = reinterpret_cast<const uint32_t*>(&s_in[r * 72 + 4 * g]);
            const uint32_t w0 = in32[0], w1 = in32[1], w2 = in32[2];
#define PSL_PAIR(hi, lo, b0, b1) __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (uint32_t)(b0) | ((uint32_t)(b1) << 16)))
            const u16x2 p0 = PSL_PAIR(w1, w0, 1, 2), p1 = PSL_PAIR(w1, w0, 2, 3), p2 = PSL_PAIR(w1, w0, 3, 4), p3 = PSL_PAIR(w1, w0, 4, 5);
            const u16x2 p4 = PSL_PAIR(w1, w0, 5, 6), p5 = PSL_PAIR(w1, w0, 6, 7), p6 = PSL_PAIR(w2, w1, 3, 4);
#undef PSL_PAIR
            const u16x2 o01 = k0 * (p0 + p4) + k1 * (p1 + p3) + k2 * p2;
            const u16x2 o23 = k0 * (p2 + p6) + k1 * (p3 + p5) + k2 * p4;
            *reinterpret_cast<uint2*>(&s_row[r * 68 + 4 * g]) = make_uint2(__builtin_bit_cast(uint32_t, o01), __builtin_bit_cast(uint32_t, o23));
        }
    }
    __syncthreads();
    {
        // column pass at blurred rows y0-1 .. y0+32 (index r = 0..33): row r reads row sums r .. r+4.  A thread makes
        // 4 columns x 4 rows from 8 row-sum rows.
        const u16x2 c01 = __builtin_bit_cast(u16x2, K0 | (K1 << 16)), c23 = __builtin_bit_cast(u16x2, K2 | (K1 << 16));
        for (int k = tid; k < 9 * 17; k += 256) {
            const int st = k / 17, g = k - st * 17;
            const int r0 = st * 4;
            uint2 q[8];
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) q[rr] = *reinterpret_cast<const uint2*>(&s_row[min(r0 + rr, 37) * 68 + 4 * g]);
            uint32_t acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int x = 0; x < 4; ++x) acc[i][x] = 1u << 15;
#pragma unroll
            for (int rr = 0; rr < 6; ++rr) {  // pair of rows (rr, rr + 1)
                const u16x2 a0 = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(q[rr + 1].x, q[rr].x, 0x05040100u));
                const u16x2 a1 = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(q[rr + 1].x, q[rr].x, 0x07060302u));
                const u16x2 a2 = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(q[rr + 1].y, q[rr].y, 0x05040100u));
                const u16x2 a3 = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(q[rr + 1].y, q[rr].y, 0x07060302u));
#pragma unroll
                for (int i = 0; i < 4; ++i) {  // output row r0 + i uses the pairs starting at i (K0,K1) and i + 2 (K2,K1)
                    const int m = rr - i;
                    if (m == 0 || m == 2) {
                        const u16x2 c = m == 0 ? c01 : c23;
                        acc[i][0] = __builtin_amdgcn_udot2(a0, c, acc[i][0], false);
                        acc[i][1] = __builtin_amdgcn_udot2(a1, c, acc[i][1], false);
                        acc[i][2] = __builtin_amdgcn_udot2(a2, c, acc[i][2], false);
                        acc[i][3] = __builtin_amdgcn_udot2(a3, c, acc[i][3], false);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (r0 + i >= 34) break;
                const uint2 t = q[i + 4];  // 5th tap, coefficient K0
                acc[i][0] += K0 * (t.x & 0xffff); acc[i][1] += K0 * (t.x >> 16);
                acc[i][2] += K0 * (t.y & 0xffff); acc[i][3] += K0 * (t.y >> 16);
                const uint32_t m0 = min(acc[i][0], 0xffffffu), m1 = min(acc[i][1], 0xffffffu);
                const uint32_t m2 = min(acc[i][2], 0xffffffu), m3 = min(acc[i][3], 0xffffffu);
                const uint32_t lo = __builtin_amdgcn_perm(m1, m0, 0x0c0c0602u), hi = __builtin_amdgcn_perm(m3, m2, 0x06020c0cu);
                *reinterpret_cast<uint32_t*>(&s_bl[(r0 + i) * 68 + 4 * g]) = lo | hi;
            }
        }
    }
    __syncthreads();
    for (int k = tid; k < 32 * 16; k += 256) {  // Sobel 3x3, 4 pixels per thread: blurred bytes 4g .. 4g+5 of three rows
        const int oy = k >> 4, g = k & 15;
        const int x = x0 + 4 * g, y = y0 + oy;
        if (x >= P.w || y >= P.h) continue;
        int c[3][6];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const uint32_t* b32 = reinterpret_cast<const uint32_t*>(&s_bl[(oy + rr) * 68 + 4 * g]);
            const uint32_t w0 = b32[0], w1 = b32[1];
            c[rr][0] = w0 & 255; c[rr][1] = (w0 >> 8) & 255; c[rr][2] = (w0 >> 16) & 255; c[rr][3] = w0 >> 24;
            c[rr][4] = w1 & 255; c[rr][5] = (w1 >> 8) & 255;
        }
        short2 o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a00 = c[0][j], a01 = c[0][j + 1], a02 = c[0][j + 2], a10 = c[1][j], a12 = c[1][j + 2], a20 = c[2][j], a21 = c[2][j + 1], a22 = c[2][j + 2];
            o[j] = make_short2((short)((a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20)), (short)((a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02)));
        }
        short2* dst = dxy + (size_t)frame * P.w * P.h + (size_t)y * P.w + x;
        if ((P.w & 3) == 0 && x + 3 < P.w) {
            *reinterpret_cast<uint4*>(dst) = make_uint4(__builtin_bit_cast(uint32_t, o[0]), __builtin_bit_cast(uint32_t, o[1]),
                                                        __builtin_bit_cast(uint32_t, o[2]), __builtin_bit_cast(uint32_t, o[3]));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (x + j < P.w) dst[j] = o[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LBD (computeLBD :1027-1373): one wave per line, lane = row hID of the 63-row line support region.
// Every float accumulation runs in the reference's order: a lane walks its row left to right, band
// sums are taken over hID ascending by lanes 0..8, the 72-float normalisation by lane 0.
// ---------------------------------------------------------------------------------------------
__constant__ int c_lbd_combos[32][2] = {{0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {0, 6}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6},
                                        {2, 3}, {2, 4}, {2, 5}, {2, 6}, {2, 7}, {2, 8}, {3, 4}, {3, 5}, {3, 6}, {3, 7}, {3, 8},
                                        {4, 5}, {4, 6}, {4, 7}, {4, 8}, {5, 6}, {5, 7}, {5, 8}, {6, 7}, {6, 8}, {7, 8}};

#ifndef PSL_LBD_WAVES
#define PSL_LBD_WAVES 1
#endif
__global__ __launch_bounds__(256, PSL_LBD_WAVES) void k_lbd(LineParams P, const short2* __restrict__ dxyI,
                                              const PslKeyLine* __restrict__ kls, const int* __restrict__ nkl, uint8_t* __restrict__ desc,
                                              float* __restrict__ fdesc) {
    __shared__ float s_row[4][63][8];
    __shared__ float s_des[4][72];
    const int frame = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int line = blockIdx.x * 4 + wave;
    const bool active = line < nkl[frame];
    const PslKeyLine kl = kls[(size_t)frame * P.maxkl + (active ? line : 0)];
    const short2* pdxy = dxyI + (size_t)frame * P.w * P.h;
    const int NB = 9, WB = 7;
    const short realWidth = (short)P.w, imageWidth = (short)(realWidth - 1), imageHeight = (short)(P.h - 1);
    const short lengthOfLSP = (short)kl.numOfPixels;
    const short halfHeight = (short)((WB * NB - 1) / 2), halfWidth = (short)((lengthOfLSP - 1) / 2);
    const float midX = (float)PSL_DMUL(0.5, (double)PSL_FADD(kl.sPointInOctaveX, kl.ePointInOctaveX));
    const float midY = (float)PSL_DMUL(0.5, (double)PSL_FADD(kl.sPointInOctaveY, kl.ePointInOctaveY));
    float dL0, dL1;
    psl_sincosf(kl.angle, &dL1, &dL0);  // dL = (cos(direction), sin(direction)), float overloads
    const float dO0 = -dL1, dO1 = dL0;
    if (active && lane < 63) {
        float sCorX0 = PSL_FADD(PSL_FADD(PSL_FMUL(-dL0, (float)halfWidth), PSL_FMUL(dL1, (float)halfHeight)), midX);
        float sCorY0 = PSL_FADD(PSL_FSUB(PSL_FMUL(-dL1, (float)halfWidth), PSL_FMUL(dL0, (float)halfHeight)), midY);
        for (int hh = 0; hh < lane; ++hh) { sCorX0 = PSL_FSUB(sCorX0, dL1); sCorY0 = PSL_FADD(sCorY0, dL0); }
        float sCorX = sCorX0, sCorY = sCorY0;
        float pL = 0, nL = 0, pO = 0, nO = 0;
        // The gathers of a row do not depend on each other: fetch 8 samples ahead (the coordinate chain is plain
        // float adds), then accumulate them in the reference's order.  One sample per wait made the kernel
        // latency-bound (90 % of the wave cycles in s_waitcnt).
        // The same values with fewer instructions per sample (the kernel is bound by vector issue, ~55 per sample before): the rounded coordinate
        // is clamped as an int (v_med3; it is far inside the range of the reference's short), the line's length is wave-uniform (a scalar loop
        // bound: full blocks of 8 run without a guard), and "if (g > 0) p += g; else n -= g;" is p += max(g, 0); n -= min(g, 0): adding or
        // subtracting a zero of either sign leaves p and n - which start at +0 and only ever grow - bit for bit as they are.
        const int len = __builtin_amdgcn_readfirstlane((int)lengthOfLSP);
        const int iw = (int)imageWidth, ih = (int)imageHeight, rw = (int)realWidth;
        auto fetch8 = [&](short2* g8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int t = (int)__builtin_roundf(sCorX);
                const int xCor = t < 0 ? 0 : (t > iw ? iw : t);
                t = (int)__builtin_roundf(sCorY);
                const int yCor = t < 0 ? 0 : (t > ih ? ih : t);
                g8[u] = pdxy[(uint32_t)(__mul24(yCor, rw) + xCor)];  // coordinates are clamped: reads past the line's end are harmless (24-bit multiply: a full-rate instruction)
                sCorX = PSL_FADD(sCorX, dL0);
                sCorY = PSL_FADD(sCorY, dL1);
            }
        };
        auto sample = [&](const short2 g) {
            const float gx = (float)g.x, gy = (float)g.y;
            const float gDL = PSL_FADD(PSL_FMUL(gx, dL0), PSL_FMUL(gy, dL1));
            const float gDO = PSL_FADD(PSL_FMUL(gx, dO0), PSL_FMUL(gy, dO1));
            pL = PSL_FADD(pL, __builtin_fmaxf(gDL, 0.0f)); nL = PSL_FSUB(nL, __builtin_fminf(gDL, 0.0f));
            pO = PSL_FADD(pO, __builtin_fmaxf(gDO, 0.0f)); nO = PSL_FSUB(nO, __builtin_fminf(gDO, 0.0f));
        };
        // The gathers of a row do not depend on each other: fetch 8 samples ahead (the coordinate chain is plain
        // float adds), then accumulate them in the reference's order.  One sample per wait made the kernel
        // latency-bound (90 % of the wave cycles in s_waitcnt).
        int w0 = 0;
        for (; w0 + 8 <= len; w0 += 8) {
            short2 g8[8];
            fetch8(g8);
#pragma unroll
            for (int u = 0; u < 8; ++u) sample(g8[u]);
        }
        if (w0 < len) {
            short2 g8[8];
            fetch8(g8);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (w0 + u < len) sample(g8[u]);
        }
        const float coef = P.gaussG[lane];
        pL = PSL_FMUL(coef, pL); nL = PSL_FMUL(coef, nL); pO = PSL_FMUL(coef, pO); nO = PSL_FMUL(coef, nO);
        float* r = s_row[wave][lane];
        r[0] = pL; r[1] = nL; r[2] = PSL_FMUL(pL, pL); r[3] = PSL_FMUL(nL, nL);
        r[4] = pO; r[5] = nO; r[6] = PSL_FMUL(pO, pO); r[7] = PSL_FMUL(nO, nO);
    }
    __syncthreads();
    if (active && lane < NB) {
        const int b = lane;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // pL, nL, pL2, nL2, pO, nO, pO2, nO2
        const int h0 = max(0, WB * (b - 1)), h1 = min(WB * NB, WB * (b + 2));
        for (int hID = h0; hID < h1; ++hID) {
            const int own = hID / WB;
            const float c = own == b ? P.gaussL[hID % WB + WB] : (own == b + 1 ? P.gaussL[hID % WB + 2 * WB] : P.gaussL[hID % WB]);
            const float* r = s_row[wave][hID];
            const float cc = PSL_FMUL(c, c);
            acc[0] = PSL_FADD(acc[0], PSL_FMUL(c, r[0])); acc[1] = PSL_FADD(acc[1], PSL_FMUL(c, r[1]));
            acc[2] = PSL_FADD(acc[2], PSL_FMUL(cc, r[2])); acc[3] = PSL_FADD(acc[3], PSL_FMUL(cc, r[3]));
            acc[4] = PSL_FADD(acc[4], PSL_FMUL(c, r[4])); acc[5] = PSL_FADD(acc[5], PSL_FMUL(c, r[5]));
            acc[6] = PSL_FADD(acc[6], PSL_FMUL(cc, r[6])); acc[7] = PSL_FADD(acc[7], PSL_FMUL(cc, r[7]));
        }
        const float invN = (b == 0 || b == NB - 1) ? (float)(1.0 / (WB * 2.0)) : (float)(1.0 / (WB * 3.0));
        float* d = &s_des[wave][8 * b];
        float temp = PSL_FMUL(acc[0], invN);
        d[0] = temp; d[4] = sqrtf(PSL_FSUB(PSL_FMUL(acc[2], invN), PSL_FMUL(temp, temp)));
        temp = PSL_FMUL(acc[1], invN);
        d[1] = temp; d[5] = sqrtf(PSL_FSUB(PSL_FMUL(acc[3], invN), PSL_FMUL(temp, temp)));
        temp = PSL_FMUL(acc[4], invN);
        d[2] = temp; d[6] = sqrtf(PSL_FSUB(PSL_FMUL(acc[6], invN), PSL_FMUL(temp, temp)));
        temp = PSL_FMUL(acc[5], invN);
        d[3] = temp; d[7] = sqrtf(PSL_FSUB(PSL_FMUL(acc[7], invN), PSL_FMUL(temp, temp)));
    }
    __syncthreads();
    if (active && lane == 0) {
        float* v = s_des[wave];
        float tempM = 0, tempS = 0;
        for (int b = 0; b < NB; ++b) {
            const float* q = v + 8 * b;
            tempM = PSL_FADD(tempM, PSL_FMUL(q[0], q[0])); tempM = PSL_FADD(tempM, PSL_FMUL(q[1], q[1]));
            tempM = PSL_FADD(tempM, PSL_FMUL(q[2], q[2])); tempM = PSL_FADD(tempM, PSL_FMUL(q[3], q[3]));
            tempS = PSL_FADD(tempS, PSL_FMUL(q[4], q[4])); tempS = PSL_FADD(tempS, PSL_FMUL(q[5], q[5]));
            tempS = PSL_FADD(tempS, PSL_FMUL(q[6], q[6])); tempS = PSL_FADD(tempS, PSL_FMUL(q[7], q[7]));
        }
        tempM = PSL_FDIV(1.0f, sqrtf(tempM));
        tempS = PSL_FDIV(1.0f, sqrtf(tempS));
        for (int b = 0; b < NB; ++b) {
            float* q = v + 8 * b;
            q[0] = PSL_FMUL(q[0], tempM); q[1] = PSL_FMUL(q[1], tempM); q[2] = PSL_FMUL(q[2], tempM); q[3] = PSL_FMUL(q[3], tempM);
            q[4] = PSL_FMUL(q[4], tempS); q[5] = PSL_FMUL(q[5], tempS); q[6] = PSL_FMUL(q[6], tempS); q[7] = PSL_FMUL(q[7], tempS);
        }
        for (int i = 0; i < 72; ++i) if ((double)v[i] > 0.4) v[i] = (float)0.4;
        float temp = 0;
        for (int i = 0; i < 72; ++i) temp = PSL_FADD(temp, PSL_FMUL(v[i], v[i]));
        temp = PSL_FDIV(1.0f, sqrtf(temp));
        for (int i = 0; i < 72; ++i) v[i] = PSL_FMUL(v[i], temp);
    }
    __syncthreads();
    if (!active) return;
    const size_t o = (size_t)frame * P.maxkl + line;
    if (lane < 32) {  // binaryConversion (:402-413)
        const float* f1 = &s_des[wave][8 * c_lbd_combos[lane][0]];
        const float* f2 = &s_des[wave][8 * c_lbd_combos[lane][1]];
        uint32_t r = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) r |= (uint32_t)(f1[i] > f2[i]) << i;
        desc[o * 32 + lane] = (uint8_t)r;
    }
    if (fdesc) for (int i = lane; i < 72; i += 64) fdesc[o * 72 + i] = s_des[wave][i];
}

// ---------------------------------------------------------------------------------------------
// CPartiallyRecoverConnectivity (PartiallyRecoverConnectivity.cpp:14-133): one workgroup per frame.
// Rows are produced in the reference's order (line i outer; candidate points = all start points
// then all end points) by ordered compaction; then the unordered-pair de-duplication that keeps the
// LAST occurrence.  ptsDropInRotatedRect is evaluated as cv::addWeighted evaluates the folded
// MatExpr: x*dcos + y*dsin + (float)(-cx*dcos - cy*dsin).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double psl_det2(float a, float b, float c, float d) {  // cv::determinant, 2x2 CV_32F -> double
    return PSL_DSUB(PSL_DMUL((double)a, (double)d), PSL_DMUL((double)b, (double)c));
}

__device__ __forceinline__ int psl_block_excl_scan256(int v, int* s_w, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    const int t0 = s_w[0], t1 = s_w[1], t2 = s_w[2], t3 = s_w[3];
    const int base = w == 0 ? 0 : (w == 1 ? t0 : (w == 2 ? t0 + t1 : t0 + t1 + t2));
    *total = t0 + t1 + t2 + t3;
    __syncthreads();
    return inc - v + base;
}

#ifndef PSL_PAIR_WAVES
#define PSL_PAIR_WAVES 8   // 56 VGPRs instead of 58: 8 waves per SIMD instead of 7, 4.88 -> 4.37 ms per 12288 dense frames (profiles/r03z_ab_waves_misc.log; the same knob
                         // does nothing for k_lsd_grad and costs k_lbd and k_line_good their registers: left as they are)
#endif
__global__ __launch_bounds__(256, PSL_PAIR_WAVES) void k_lil_pair(const float* __restrict__ lines, size_t lstride, int lp, const int* __restrict__ nlines, int nlines_single,
                                                   float radius, float fanThr, int imgCols, int imgRows, float* __restrict__ raw,
                                                   float* __restrict__ fans, int fan_cap, int* __restrict__ nfans) {
    __shared__ int s_w[4];
    const int frame = blockIdx.x, tid = threadIdx.x;
    const float* L = lines + (size_t)frame * lstride;
    const int rows = nlines ? nlines[frame] : nlines_single;
    float* R = raw + (size_t)frame * PSL_FAN_CAP * 4;
    int count = 0;
    for (int i = 0; i < rows; ++i) {
        const float p0 = L[lp * i], p1 = L[lp * i + 1], p2 = L[lp * i + 2], p3 = L[lp * i + 3];
        const float cenx = PSL_FADD(p0, p2) / 2, ceny = PSL_FADD(p1, p3) / 2;
        const float dy = PSL_FSUB(p3, p1), dx = PSL_FSUB(p2, p0);
        const float degAng = psl_fast_atan2(dy, dx);
        const float arcAng = (float)PSL_DMUL((double)(degAng / 180), PSL_PI);
        const float length = __builtin_fabsf(psl_tanf(arcAng)) > 1 ? __builtin_fabsf(dy) : __builtin_fabsf(dx);
        const int th = (int)PSL_FMUL(radius, 2.f), tw = (int)PSL_FADD(length, PSL_FMUL(2.f, radius));  // CvSize is integer
        const float hafW = (float)tw / 2, hafH = (float)th / 2;
        const float angle = (float)(PSL_DMUL((double)degAng, PSL_PI) / 180);
        float dsin, dcos;
        psl_sincosf(angle, &dsin, &dcos);
        const float gx = (float)PSL_DSUB(PSL_DMUL(-(double)cenx, (double)dcos), PSL_DMUL((double)ceny, (double)dsin));
        const float gy = (float)PSL_DADD(PSL_DMUL(-(double)cenx, (double)dsin), PSL_DMUL((double)ceny, (double)dcos));
        const float ndcos = -dcos;
        for (int base = 0; base < 2 * rows; base += 256) {
            const int pj = base + tid;
            bool keep = false;
            float X = 0, Y = 0;
            int curSer = 0;
            if (pj < 2 * rows) {
                const float px = pj < rows ? L[lp * pj] : L[lp * (pj - rows) + 2];
                const float py = pj < rows ? L[lp * pj + 1] : L[lp * (pj - rows) + 3];
                const float fposx = PSL_FADD(PSL_FADD(PSL_FMUL(px, dcos), PSL_FMUL(py, dsin)), gx);
                const float fposy = PSL_FADD(PSL_FADD(PSL_FMUL(px, dsin), PSL_FMUL(py, ndcos)), gy);
                curSer = pj >= rows ? pj - rows : pj;
                if (-hafW <= fposx && fposx < hafW && -hafH <= fposy && fposy < hafH && curSer != i) {
                    const float q0 = L[lp * curSer], q1 = L[lp * curSer + 1], q2 = L[lp * curSer + 2], q3 = L[lp * curSer + 3];
                    const float degAng1 = psl_fast_atan2(PSL_FSUB(q3, q1), PSL_FSUB(q2, q0));
                    const float arcAng1 = (float)PSL_DMUL((double)(degAng1 / 180), PSL_PI);
                    const float tmpa = fmodf(__builtin_fabsf(PSL_FSUB(arcAng, arcAng1)), (float)PSL_PI);
                    if (!(tmpa < fanThr || PSL_DSUB(PSL_PI, (double)tmpa) < (double)fanThr)) {
                        // intersectionOfLines (:226-247)
                        const float A1 = PSL_FSUB(p1, p3), B1 = PSL_FSUB(p2, p0), C1 = PSL_FSUB(PSL_FMUL(p3, p0), PSL_FMUL(p1, p2));
                        const float A2 = PSL_FSUB(q1, q3), B2 = PSL_FSUB(q2, q0), C2 = PSL_FSUB(PSL_FMUL(q3, q0), PSL_FMUL(q1, q2));
                        const float D = (float)psl_det2(A1, B1, A2, B2);
                        X = (float)(psl_det2(-C1, B1, -C2, B2) / (double)D);
                        Y = (float)(psl_det2(A1, -C1, A2, -C2) / (double)D);
                        // isPtInRotatedRect (:135-149), scalar float arithmetic
                        const float fx = PSL_FADD(PSL_FMUL(dcos, PSL_FSUB(X, cenx)), PSL_FMUL(dsin, PSL_FSUB(Y, ceny)));
                        const float fy = PSL_FSUB(PSL_FMUL(dsin, PSL_FSUB(X, cenx)), PSL_FMUL(dcos, PSL_FSUB(Y, ceny)));
                        keep = (-hafW <= fx && fx < hafW && -hafH <= fy && fy < hafH) &&
                               (X >= 4 && X < imgCols - 4 && Y >= 4 && Y < imgRows - 4);
                    }
                }
            }
            int tot;
            const int ofs = psl_block_excl_scan256(keep ? 1 : 0, s_w, &tot);
            if (keep && count + ofs < PSL_FAN_CAP) {
                float* r = R + 4 * (size_t)(count + ofs);
                r[0] = X; r[1] = Y; r[2] = (float)i; r[3] = (float)curSer;
            }
            count += tot;
        }
    }
    if (count > PSL_FAN_CAP) count = PSL_FAN_CAP;
    __syncthreads();
    // keep the LAST occurrence of every unordered (i, j) pair (:109-131), order preserved
    float* F = fans + (size_t)frame * fan_cap * 4;
    int kept = 0;
    for (int base = 0; base < count; base += 256) {
        const int r = base + tid;
        bool flag = false;
        if (r < count) {
            const int s1 = (int)R[4 * r + 2], s2 = (int)R[4 * r + 3];
            flag = true;
            for (int j = r + 1; j < count; ++j) {
                const int s3 = (int)R[4 * j + 2], s4 = (int)R[4 * j + 3];
                if ((s1 == s3 && s2 == s4) || (s1 == s4 && s2 == s3)) { flag = false; break; }
            }
        }
        int tot;
        const int ofs = psl_block_excl_scan256(flag ? 1 : 0, s_w, &tot);
        if (flag && kept + ofs < fan_cap) {
            float* d = F + 4 * (size_t)(kept + ofs);
            d[0] = R[4 * r]; d[1] = R[4 * r + 1]; d[2] = R[4 * r + 2]; d[3] = R[4 * r + 3];
        }
        kept += tot;
    }
    if (tid == 0) nfans[frame] = kept < fan_cap ? kept : fan_cap;
}


// ---------------------------------------------------------------------------------------------
// LSDmatcher::matchNNR (add_src/LSDmatcher.cpp:354-376) for every frame of a batch against the frame
// `shift` positions earlier (cyclic), i.e. lmatcher.match(mLastFrame.mLdesc, mCurrentFrame.mLdesc, nnr)
// of src/Tracking.cc:901 for a whole stream at once.  One wave per query row; kNN-2 by Hamming
// distance with "lower train index first"; accept d0 < d1 * nnr (float compare).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_line_match_batch(const uint8_t* __restrict__ desc, const int* __restrict__ counts, int cap, int nframes,
                                                           int shift, float nnr, int* __restrict__ matches12, int* __restrict__ nmatches) {
    const int cur = blockIdx.x, lane = threadIdx.x & 63;
    const int last = ((cur - shift) % nframes + nframes) % nframes;
    const int n1 = min(counts[last], cap), n2 = min(counts[cur], cap);
    const int qi = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (qi >= n1) return;
    const uint32_t* Q = reinterpret_cast<const uint32_t*>(desc + ((size_t)last * cap + qi) * 32);
    const uint32_t* T = reinterpret_cast<const uint32_t*>(desc + (size_t)cur * cap * 32);
    uint32_t qd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qd[k] = Q[k];
    uint32_t k1 = 0xffffffffu, k2 = 0xffffffffu;
    for (int j = lane; j < n2; j += 64) {
        int d = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) d += __popc(qd[k] ^ T[(size_t)j * 8 + k]);
        const uint32_t key = ((uint32_t)d << 20) | (uint32_t)j;
        if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t o1 = __shfl_xor(k1, o), o2 = __shfl_xor(k2, o);
        const uint32_t lo = min(k1, o1), hi = max(k1, o1);
        k2 = min(hi, min(k2, o2));
        k1 = lo;
    }
    if (lane == 0) {
        int m = -1;
        if (n2 >= 2 && (float)(k1 >> 20) < PSL_FMUL((float)(k2 >> 20), nnr)) m = (int)(k1 & 0xfffff);
        matches12[(size_t)cur * cap + qi] = m;
        if (m >= 0) atomicAdd(&nmatches[cur], 1);
    }
}

#endif
