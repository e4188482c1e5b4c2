// libpslfe: the RGB-D line glue of the Frame constructor. Product code.
// Reference behaviour reproduced:
//   Frame::isLineGood                         src/Frame.cc:662-750        -> k_line_good
//   LINEextractor::compPt3dCov / verify3dLine / computeLine3d_svd / mah_dist3d_pt_line / extract3dline_mahdist
//                                             add_src/LineExtractor.cpp:40-322 ; random_unique add_inc/LineExtractor.h:23-37
//   Frame::convertFansToKeyLines + Frame_shortestDistance   src/Frame.cc:381-472   -> k_fans_planes (phase 1)
//   plane from a pair of 3-D lines + Frame::OldPlane         src/Frame.cc:474-660   -> k_fans_planes (phase 2)
// Third-party arithmetic restated (same statements as oracle/glue_oracle.cpp, see there): glibc rand() (TYPE_3, srand(seed)
// per frame: convention H7), OpenCV's one-sided Jacobi SVD (f64), Cramer's rule for the 2x2 system.
//
// Work decomposition: the RANSAC of isLineGood consumes rand() values in line order and the number consumed depends on
// the outcome, so the lines of a frame form a serial chain: one wave per frame, lines in order.  Inside a line the <= 21
// depth samples are the lanes: back-projection, the 3x3 covariance SVD and every Mahalanobis distance run in parallel;
// inlier sets are ballots; arg-min/arg-max with the reference's first-occurrence rule are wave reductions; only the
// order-sensitive sums (mean, Jacobi sweeps of the n x 3 matrix) run on one lane.  Frames fill the GPU.
#include <string.h>

#include <vector>

#include "pslfe_internal.h"

#define PSL_GLUE_MAXPTS 21

struct GlueP3 { double x, y, z; };
__device__ __forceinline__ GlueP3 operator+(const GlueP3& a, const GlueP3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ GlueP3 operator-(const GlueP3& a, const GlueP3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ GlueP3 operator*(const GlueP3& a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ double gdot(const GlueP3& a, const GlueP3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double gnorm(const GlueP3& a) { return __dsqrt_rn(a.x * a.x + a.y * a.y + a.z * a.z); }

// OpenCV JacobiSVDImpl_<double> on N rows of length M held by one lane (compile-time sizes: everything stays in
// registers).  Rows of At come back normalised, W descending, Vt = accumulated rotations (N x N).
template <int M, int N>
__device__ __forceinline__ void psl_jacobi_rows(double (&At)[N][M], double (&W)[N], double (&Vt)[N][N]) {
    const double eps = 2.220446049250313e-16 * 10;
    const int max_iter = M > 30 ? M : 30;
    double Wd[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; ++k) sd += At[i][k] * At[i][k];
        Wd[i] = sd;
#pragma unroll
        for (int k = 0; k < N; ++k) Vt[i][k] = 0;
        Vt[i][i] = 1;
    }
    for (int iter = 0; iter < max_iter; ++iter) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < N - 1; ++i)
#pragma unroll
            for (int j = i + 1; j < N; ++j) {
                double a = Wd[i], p = 0, b = Wd[j];
#pragma unroll
                for (int k = 0; k < M; ++k) p += At[i][k] * At[j][k];
                if (!(fabs(p) <= eps * __dsqrt_rn(a * b))) {
                    p *= 2;
                    const double beta = a - b, gamma = __dsqrt_rn(p * p + beta * beta);
                    double c, s;
                    if (beta < 0) {
                        const double delta = (gamma - beta) * 0.5;
                        s = __dsqrt_rn(delta / gamma);
                        c = p / (gamma * s * 2);
                    } else {
                        c = __dsqrt_rn((gamma + beta) / (gamma * 2));
                        s = p / (gamma * c * 2);
                    }
                    a = b = 0;
#pragma unroll
                    for (int k = 0; k < M; ++k) {
                        const double t0 = c * At[i][k] + s * At[j][k];
                        const double t1 = -s * At[i][k] + c * At[j][k];
                        At[i][k] = t0; At[j][k] = t1;
                        a += t0 * t0; b += t1 * t1;
                    }
                    Wd[i] = a; Wd[j] = b;
                    changed = true;
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        const double t0 = c * Vt[i][k] + s * Vt[j][k];
                        const double t1 = -s * Vt[i][k] + c * Vt[j][k];
                        Vt[i][k] = t0; Vt[j][k] = t1;
                    }
                }
            }
        if (!changed) break;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; ++k) sd += At[i][k] * At[i][k];
        Wd[i] = __dsqrt_rn(sd);
    }
#pragma unroll
    for (int i = 0; i < N - 1; ++i) {
        // selection of the largest remaining value (first one on ties), then a swap: written without dynamic indices
        int j = i;  // "if (Wd[j] < Wd[k]) j = k" followed by one swap(i, j), without dynamic indices
#pragma unroll
        for (int k = i + 1; k < N; ++k) {
            double wj = Wd[i];
#pragma unroll
            for (int q = i + 1; q < N; ++q) wj = (j == q) ? Wd[q] : wj;
            if (wj < Wd[k]) j = k;
        }
#pragma unroll
        for (int q = i + 1; q < N; ++q)
            if (j == q) {
                const double t = Wd[i]; Wd[i] = Wd[q]; Wd[q] = t;
#pragma unroll
                for (int k = 0; k < M; ++k) { const double u = At[i][k]; At[i][k] = At[q][k]; At[q][k] = u; }
#pragma unroll
                for (int k = 0; k < N; ++k) { const double u = Vt[i][k]; Vt[i][k] = Vt[q][k]; Vt[q][k] = u; }
            }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        W[i] = Wd[i];
        const double s = Wd[i] > 2.2250738585072014e-308 ? 1 / Wd[i] : 0.;
#pragma unroll
        for (int k = 0; k < M; ++k) At[i][k] *= s;
    }
}

// LINEextractor::compPt3dCov (:40-93): DU = diag(1/sqrt(W)) * U^T of cov0 = J0 * diag(1, 1, sigma_z^2) * J0^T
__device__ void psl_comp_du(const GlueP3& pt, double f, double* DU) {
    const double J0[3][3] = {{pt.z / f, 0, pt.x / pt.z}, {0, pt.z / f, pt.y / pt.z}, {0, 0, 1}};
    const double c1 = 0.00273, c2 = 0.00074, c3 = -0.00058;
    const double sd = c1 * pt.z * pt.z + c2 * pt.z + c3;  // depthStdDev (:27-38)
    const double G[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, sd * sd}};
    double JG[3][3], cov[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) JG[i][j] = J0[i][0] * G[0][j] + J0[i][1] * G[1][j] + J0[i][2] * G[2][j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) cov[i][j] = JG[i][0] * J0[j][0] + JG[i][1] * J0[j][1] + JG[i][2] * J0[j][2];
    double At[3][3], W[3], Vt[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) At[i][j] = cov[j][i];
    psl_jacobi_rows<3, 3>(At, W, Vt);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double d = 1 / __dsqrt_rn(W[r]);
#pragma unroll
        for (int c = 0; c < 3; ++c) DU[3 * r + c] = d * At[r][c];
    }
}

// LINEextractor::mah_dist3d_pt_line (:187-214)
__device__ double psl_mah_dist(const double* DU, const GlueP3& pos, const GlueP3& q1, const GlueP3& q2) {
    const double xa = q1.x, ya = q1.y, za = q1.z, xb = q2.x, yb = q2.y, zb = q2.z;
    const double c1 = DU[0], c2 = DU[1], c3 = DU[2], c4 = DU[3], c5 = DU[4], c6 = DU[5], c7 = DU[6], c8 = DU[7], c9 = DU[8];
    const double x1 = pos.x, x2 = pos.y, x3 = pos.z;
    const double term1 = ((c1 * (x1 - xa) + c2 * (x2 - ya) + c3 * (x3 - za)) * (c4 * (x1 - xb) + c5 * (x2 - yb) + c6 * (x3 - zb)) -
                          (c4 * (x1 - xa) + c5 * (x2 - ya) + c6 * (x3 - za)) * (c1 * (x1 - xb) + c2 * (x2 - yb) + c3 * (x3 - zb))),
                 term2 = ((c1 * (x1 - xa) + c2 * (x2 - ya) + c3 * (x3 - za)) * (c7 * (x1 - xb) + c8 * (x2 - yb) + c9 * (x3 - zb)) -
                          (c7 * (x1 - xa) + c8 * (x2 - ya) + c9 * (x3 - za)) * (c1 * (x1 - xb) + c2 * (x2 - yb) + c3 * (x3 - zb))),
                 term3 = ((c4 * (x1 - xa) + c5 * (x2 - ya) + c6 * (x3 - za)) * (c7 * (x1 - xb) + c8 * (x2 - yb) + c9 * (x3 - zb)) -
                          (c7 * (x1 - xa) + c8 * (x2 - ya) + c9 * (x3 - za)) * (c4 * (x1 - xb) + c5 * (x2 - yb) + c6 * (x3 - zb))),
                 term4 = (c1 * (x1 - xa) - c1 * (x1 - xb) + c2 * (x2 - ya) - c2 * (x2 - yb) + c3 * (x3 - za) - c3 * (x3 - zb)),
                 term5 = (c4 * (x1 - xa) - c4 * (x1 - xb) + c5 * (x2 - ya) - c5 * (x2 - yb) + c6 * (x3 - za) - c6 * (x3 - zb)),
                 term6 = (c7 * (x1 - xa) - c7 * (x1 - xb) + c8 * (x2 - ya) - c8 * (x2 - yb) + c9 * (x3 - za) - c9 * (x3 - zb));
    return __dsqrt_rn((term1 * term1 + term2 * term2 + term3 * term3) / (term4 * term4 + term5 * term5 + term6 * term6));
}

__device__ __forceinline__ GlueP3 psl_proj_pt_ln(const GlueP3& P, const GlueP3& mid, const GlueP3& drct) {  // projPt3d2Ln3d
    const GlueP3 A = mid, B = mid + drct, AB = B - A, AP = P - A;
    return A + AB * (gdot(AB, AP) / gdot(AB, AB));
}

// First index (lowest lane) holding the minimum / maximum of v over the lanes of `mask`, with the reference's start values:
// "if (v < minv) ..." from minv = 100 and "if (v > maxv) ..." from maxv = -100, index 0 of the LIST (= the lowest lane of the
// mask) when nothing beats the start value.
__device__ int psl_first_arg(double v, unsigned long long mask, bool want_min) {
    const int lane = threadIdx.x & 63;
    const bool in = (mask >> lane) & 1ull;
    double best = in ? v : (want_min ? 1e300 : -1e300);
    int idx = in ? lane : 64;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(idx, o);
        const bool take = want_min ? (ob < best || (ob == best && oi < idx)) : (ob > best || (ob == best && oi < idx));
        if (take) { best = ob; idx = oi; }
    }
    const bool beats = want_min ? best < 100 : best > -100;
    return beats ? idx : (int)__ffsll((long long)mask) - 1;
}

struct GlueLds {
    double pos[PSL_GLUE_MAXPTS][3];
    double DU[PSL_GLUE_MAXPTS][9];
    uint32_t ring[34];
    int idx[PSL_GLUE_MAXPTS + 3];   // the RANSAC shuffle (`indexes`)
    int rank[PSL_GLUE_MAXPTS + 3];  // rank -> point of an inlier set
    double term[3][24];             // psl_ordered_sum3: the terms of three ordered sums, one row each
};

__device__ __forceinline__ GlueP3 glue_pos(const GlueLds& S, int i) { return {S.pos[i][0], S.pos[i][1], S.pos[i][2]}; }

// verify3dLine (:95-161) on the points of `mask`
__device__ bool psl_verify_line(const GlueLds& S, unsigned long long mask, const GlueP3& A, const GlueP3& B, int np) {
    const int lane = threadIdx.x & 63;
    const GlueP3 me = lane < np ? glue_pos(S, lane) : GlueP3{0, 0, 0};
    const double v = gdot(me - A, B - A);
    const int i1 = psl_first_arg(v, mask, true), i2 = psl_first_arg(v, mask, false);
    const GlueP3 C = psl_proj_pt_ln(glue_pos(S, i1), (A + B) * 0.5, B - A);
    const GlueP3 D = psl_proj_pt_ln(glue_pos(S, i2), (A + B) * 0.5, B - A);
    const double cd = gnorm(D - C);
    if (cd < 0.0000000001) return false;
    uint32_t bit = 0;
    if ((mask >> lane) & 1ull) {
        double lambda = gdot(me - C, D - C) / cd / cd;
        lambda = lambda < 0 ? -lambda : lambda;
        bit = lambda >= 1 ? (1u << 9) : (1u << (unsigned int)floor(lambda * 10));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bit |= __shfl_xor(bit, o);
    const double sum = (double)__popc(bit);
    return sum / 10 > 0.7;
}

// Sum of `term` over lanes 0..m-1 in lane order (the reference's sequential accumulation), the same value in every lane.
// THREE sums over the first m lanes' terms, each added strictly in lane order starting from +0.0 (the reference's `s += x[k]` loops:
// the rounding depends on the order), at the price of one: "terms in parallel, additions in series" - the terms are staged in three
// LDS rows, lane r (r = 0, 1, 2) adds row r, so ONE chain of m dependent f64 adds serves the three sums (one readlane pair + add
// per term and sum before: the n x 3 Jacobi SVD of isLineGood's refinement spent three quarters of its instructions there).  Rows
// are padded with +0.0 to the batch of 8: x + (+0.0) == x for every value such a running sum can hold (it is never -0.0).
__device__ __forceinline__ void psl_ordered_sum3(GlueLds& S, double t0, double t1, double t2, int m, double* s0, double* s1, double* s2) {
    const int lane = threadIdx.x & 63;
    if (lane < 24) {
        const bool in = lane < m;
        S.term[0][lane] = in ? t0 : 0.0; S.term[1][lane] = in ? t1 : 0.0; S.term[2][lane] = in ? t2 : 0.0;
    }
    __builtin_amdgcn_wave_barrier();
    const double* my = S.term[lane < 3 ? lane : 0];
    double acc = 0;
    for (int t = 0; t < m; t += 8) {   // m <= 21 (uniform)
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = my[t + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
    }
    __builtin_amdgcn_wave_barrier();   // the rows are rewritten by the next call
    const int lo = __double2loint(acc), hi = __double2hiint(acc);
    *s0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    *s1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 1), __builtin_amdgcn_readlane(lo, 1));
    *s2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 2), __builtin_amdgcn_readlane(lo, 2));
}

// OpenCV's JacobiSVDImpl_<double> on a matrix whose n <= 3 rows of length m <= 21 are spread over the lanes (lane k holds column k:
// a[0..2]).  The reference's sweep order (0,1), (0,2), (1,2) and every sum's order are kept; a rotation's two new row norms and the
// NEXT pair's scalar product - it is formed from the rows as the rotation leaves them, and nothing changes them before that pair is
// looked at - share one psl_ordered_sum3.
__device__ void psl_jacobi_wave(GlueLds& S, double a[3], int m, int n, double W[3], double Vt[9]) {
    const double eps = 2.220446049250313e-16 * 10;
    const int max_iter = m > 30 ? m : 30;
    double Wd[3] = {0, 0, 0};
    auto get = [&](const double* v, int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : v[2]); };
    auto put = [&](double* v, int i, double x) { v[0] = i == 0 ? x : v[0]; v[1] = i == 1 ? x : v[1]; v[2] = i == 2 ? x : v[2]; };
    {
        double s0, s1, s2;   // rows i >= n hold zeros: their sums are the +0.0 Wd starts with
        psl_ordered_sum3(S, a[0] * a[0], a[1] * a[1], a[2] * a[2], m, &s0, &s1, &s2);
        Wd[0] = s0; Wd[1] = n > 1 ? s1 : 0.0; Wd[2] = n > 2 ? s2 : 0.0;
    }
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < n; ++k) Vt[i * 3 + k] = 0;
        Vt[i * 3 + i] = 1;
    }
    const int npairs = n == 3 ? 3 : (n == 2 ? 1 : 0);
    bool have_p = false;
    double p_next = 0;
    for (int iter = 0; iter < max_iter; ++iter) {
        bool changed = false;
        for (int pr = 0; pr < npairs; ++pr) {
            const int i = pr == 2 ? 1 : 0, j = pr == 0 ? 1 : 2;
            const double ai = get(a, i), aj = get(a, j);
            const double aa = get(Wd, i), bb = get(Wd, j);
            double p;
            if (have_p) p = p_next;
            else { double d1, d2; psl_ordered_sum3(S, ai * aj, 0.0, 0.0, m, &p, &d1, &d2); }
            have_p = false;
            if (fabs(p) <= eps * __dsqrt_rn(aa * bb)) continue;
            p *= 2;
            const double beta = aa - bb, gamma = __dsqrt_rn(p * p + beta * beta);
            double c, sn;
            if (beta < 0) {
                const double delta = (gamma - beta) * 0.5;
                sn = __dsqrt_rn(delta / gamma);
                c = p / (gamma * sn * 2);
            } else {
                c = __dsqrt_rn((gamma + beta) / (gamma * 2));
                sn = p / (gamma * c * 2);
            }
            const double t0 = c * ai + sn * aj;
            const double t1 = -sn * ai + c * aj;
            put(a, i, t0); put(a, j, t1);
            // the pair the sweep looks at next (the first pair of the next sweep after the last one), on the rows as they are now
            const int npr = pr + 1 < npairs ? pr + 1 : 0;
            const int ni = npr == 2 ? 1 : 0, nj = npr == 0 ? 1 : 2;
            double w0, w1;
            psl_ordered_sum3(S, t0 * t0, t1 * t1, get(a, ni) * get(a, nj), m, &w0, &w1, &p_next);
            put(Wd, i, w0); put(Wd, j, w1);
            have_p = true;
            changed = true;
            for (int k = 0; k < n; ++k) {
                const double v0 = c * Vt[i * 3 + k] + sn * Vt[j * 3 + k];
                const double v1 = -sn * Vt[i * 3 + k] + c * Vt[j * 3 + k];
                Vt[i * 3 + k] = v0; Vt[j * 3 + k] = v1;
            }
        }
        if (!changed) break;
    }
    {
        double s0, s1, s2;
        psl_ordered_sum3(S, a[0] * a[0], a[1] * a[1], a[2] * a[2], m, &s0, &s1, &s2);
        Wd[0] = __dsqrt_rn(s0);
        if (n > 1) Wd[1] = __dsqrt_rn(s1);
        if (n > 2) Wd[2] = __dsqrt_rn(s2);
    }
    for (int i = 0; i < n - 1; ++i) {
        int j = i;
        for (int k = i + 1; k < n; ++k)
            if (Wd[j] < Wd[k]) j = k;
        if (i != j) {
            const double t = Wd[i]; Wd[i] = Wd[j]; Wd[j] = t;
            const double u = a[i]; a[i] = a[j]; a[j] = u;
            for (int k = 0; k < n; ++k) { const double v = Vt[i * 3 + k]; Vt[i * 3 + k] = Vt[j * 3 + k]; Vt[j * 3 + k] = v; }
        }
    }
    for (int i = 0; i < n; ++i) {
        W[i] = Wd[i];
        const double sc = Wd[i] > 2.2250738585072014e-308 ? 1 / Wd[i] : 0.;
        a[i] *= sc;
    }
}

// computeLine3d_svd (:163-185) on the points of `mask` (lane = point index); returns mean and direction in every lane
__device__ void psl_line_svd(GlueLds& S, unsigned long long mask, const GlueP3& me, GlueP3* mean_out, GlueP3* drct_out) {
    const int lane = threadIdx.x & 63;
    const int n = __popcll(mask);
    // rank r of an inlier = its column / row in the matrix
    if ((mask >> lane) & 1ull) S.rank[__popcll(mask & ((1ull << lane) - 1ull))] = lane;
    __builtin_amdgcn_wave_barrier();
    GlueP3 mean = {0, 0, 0};
    {   // mean = mean + pts[idx[i]].pos, in index order (= rank order): the three coordinate sums in one pass
        const GlueP3 q = lane < n ? glue_pos(S, S.rank[lane]) : GlueP3{0, 0, 0};
        psl_ordered_sum3(S, q.x, q.y, q.z, n, &mean.x, &mean.y, &mean.z);
    }
    mean = mean * (1.0 / n);
    double W[3], Vt[9], a[3] = {0, 0, 0};
    GlueP3 drct;
    if (n >= 3) {  // cv::SVD(P.t()), P.t() n x 3: A^T has 3 rows of length n, vt = V^T
        if (lane < n) {
            const GlueP3 p = glue_pos(S, S.rank[lane]);
            a[0] = p.x - mean.x; a[1] = p.y - mean.y; a[2] = p.z - mean.z;
        }
        psl_jacobi_wave(S, a, n, 3, W, Vt);
        drct = {Vt[0], Vt[1], Vt[2]};
    } else {       // fewer rows than columns: the rows of P.t() themselves (n rows of length 3), vt = their normalised rotations
        if (lane < 3) {
            for (int r = 0; r < n; ++r) {
                const GlueP3 p = glue_pos(S, S.rank[r]);
                a[r] = lane == 0 ? p.x - mean.x : (lane == 1 ? p.y - mean.y : p.z - mean.z);
            }
        }
        psl_jacobi_wave(S, a, 3, n, W, Vt);
        const double r0 = a[0];
        drct.x = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(r0), 0), __builtin_amdgcn_readlane(__double2loint(r0), 0));
        drct.y = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(r0), 1), __builtin_amdgcn_readlane(__double2loint(r0), 1));
        drct.z = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(r0), 2), __builtin_amdgcn_readlane(__double2loint(r0), 2));
    }
    __builtin_amdgcn_wave_barrier();  // S.rank is rewritten by the next call
    (void)me;
    *mean_out = mean;
    *drct_out = drct;
}

// glibc rand(): TYPE_3 ring in LDS, advanced by lane 0; *k is the (uniform) stream position
__device__ void psl_glibc_srand(GlueLds& S, uint32_t seed, int* k) {
    if ((threadIdx.x & 63) == 0) {
        int32_t s[34];
        s[0] = seed == 0 ? 1 : (int32_t)seed;
        for (int i = 1; i < 31; ++i) {
            const long long hi = s[i - 1] / 127773, lo = s[i - 1] % 127773;
            long long word = 16807 * lo - 2836 * hi;
            if (word < 0) word += 2147483647;
            s[i] = (int32_t)word;
        }
        for (int i = 31; i < 34; ++i) s[i] = s[i - 31];
        for (int i = 0; i < 34; ++i) S.ring[i] = (uint32_t)s[i];
        for (int kk = 34; kk < 34 + 310; ++kk) S.ring[kk % 34] = S.ring[(kk - 31) % 34] + S.ring[(kk - 3) % 34];
    }
    *k = 34 + 310;
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int psl_glibc_rand_lane0(GlueLds& S, int k) {  // lane 0 only
    const uint32_t v = S.ring[(k - 31) % 34] + S.ring[(k - 3) % 34];
    S.ring[k % 34] = v;
    return (int)(v >> 1);
}

#ifndef PSL_GOOD_WAVES
#define PSL_GOOD_WAVES 4
#endif
__global__ __launch_bounds__(64, PSL_GOOD_WAVES) void k_line_good(const PslKeyLine* __restrict__ kls, int kl_stride, const int32_t* __restrict__ nkl,
                                                   int nkl_single, const float* __restrict__ depth, int cols, int rows, int dstride,
                                                   size_t dframe, PslCamera cam, uint32_t seed0, double* __restrict__ lines3d,
                                                   float* __restrict__ lineEq) {
    __shared__ GlueLds S;
    const int frame = blockIdx.x, lane = threadIdx.x;
    const int n = min(nkl ? nkl[frame] : nkl_single, kl_stride);
    const PslKeyLine* K = kls + (size_t)frame * kl_stride;
    const float* D = depth + (size_t)frame * dframe;
    double* L3 = lines3d + (size_t)frame * kl_stride * 6;
    float* LE = lineEq + (size_t)frame * kl_stride * 3;
    const float cx = cam.cx, cy = cam.cy;
    const float invfx = 1.0f / cam.fx, invfy = 1.0f / cam.fy;
    for (int i = lane; i < n * 6; i += 64) L3[i] = 0.0;
    for (int i = lane; i < n * 3; i += 64) LE[i] = -1.0f;
    int rk;
    psl_glibc_srand(S, seed0 + (uint32_t)frame, &rk);
    const unsigned long long lt = (1ull << lane) - 1ull;

    for (int i = 0; i < n; ++i) {
        const float spx = K[i].startPointX, spy = K[i].startPointY, epx = K[i].endPointX, epy = K[i].endPointY;
        const float dxf = spx - epx, dyf = spy - epy;
        const double len = __dsqrt_rn((double)dxf * dxf + (double)dyf * dyf);
        const int ilen = (int)len;
        const double numSmp = (double)(ilen < 20 ? ilen : 20);
        if (numSmp == 0) continue;  // convention: 0/0 upstream
        // ---- depth samples (:674-716), lane = j
        bool valid = false;
        GlueP3 p = {0, 0, 0};
        if (lane <= (int)numSmp) {
            const int j = lane;
            const double w1 = 1 - j / numSmp, w2 = j / numSmp;
            const float ax = (float)(spx * w1), ay = (float)(spy * w1), bx = (float)(epx * w2), by = (float)(epy * w2);
            const double ptx = (double)(ax + bx), pty = (double)(ay + by);
            if (!(ptx < 0 || pty < 0 || ptx >= cols || pty >= rows)) {
                int row, col;
                if ((floor(ptx) == ptx) && (floor(pty) == pty)) {
                    col = max((int)(ptx - 1), 0);
                    row = max((int)(pty - 1), 0);
                } else {
                    col = (int)ptx;
                    row = (int)pty;
                }
                const float dv = D[(size_t)row * dstride + col];
                if (!((double)dv <= 0.01)) {
                    valid = true;
                    p.z = dv;
                    p.x = (col - cx) * p.z * invfx;
                    p.y = (row - cy) * p.z * invfy;
                }
            }
        }
        const unsigned long long vmask = __ballot(valid);
        const int np = __popcll(vmask);
        if (np < 5) continue;
        if (valid) {
            const int r = __popcll(vmask & lt);
            S.pos[r][0] = p.x; S.pos[r][1] = p.y; S.pos[r][2] = p.z;
            double DU[9];
            psl_comp_du(p, (double)cam.fx, DU);
#pragma unroll
            for (int c = 0; c < 9; ++c) S.DU[r][c] = DU[c];
        }
        if (lane < np) S.idx[lane] = lane;
        __builtin_amdgcn_wave_barrier();
        // ---- extract3dline_mahdist (:216-322)
        const GlueP3 me = lane < np ? glue_pos(S, lane) : GlueP3{0, 0, 0};
        double myDU[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) myDU[c] = lane < np ? S.DU[lane][c] : 0.0;
        const int maxIterNo = min(10, (int)(np * (np - 1) * 0.5));
        const double distThresh = 3.0;
        unsigned long long maxMask = 0;
        int maxCnt = 0, bestA = 0, bestB = 0;
        for (int iter = 0; iter < maxIterNo; ++iter) {
            if (lane == 0) {  // random_unique(indexes.begin(), indexes.end(), 2)
                int left = np, begin = 0;
                for (int num = 0; num < 2; ++num) {
                    const int r = begin + psl_glibc_rand_lane0(S, rk + num) % left;
                    const int t = S.idx[begin]; S.idx[begin] = S.idx[r]; S.idx[r] = t;
                    ++begin; --left;
                }
            }
            rk += 2;
            __builtin_amdgcn_wave_barrier();
            const int ia = S.idx[0], ib = S.idx[1];
            const GlueP3 A = glue_pos(S, ia), B = glue_pos(S, ib);
            if (gnorm(B - A) < 0.0000000001) continue;
            const bool in = lane < np && psl_mah_dist(myDU, me, A, B) < distThresh;
            const unsigned long long inMask = __ballot(in);
            const int cnt = __popcll(inMask);
            if (cnt > maxCnt) {
                if (psl_verify_line(S, inMask, A, B, np)) { maxMask = inMask; maxCnt = cnt; bestA = ia; bestB = ib; }
            }
            if (maxCnt > np * 0.6) break;
        }
        GlueP3 outA = {0, 0, 0}, outB = {0, 0, 0};
        if (maxCnt >= 2) {
            GlueP3 m = (glue_pos(S, bestA) + glue_pos(S, bestB)) * 0.5, d = glue_pos(S, bestB) - glue_pos(S, bestA);
            while (true) {
                GlueP3 tm, td;
                psl_line_svd(S, maxMask, me, &tm, &td);
                const bool in = lane < np && psl_mah_dist(myDU, me, tm, tm + td) < distThresh;
                const unsigned long long tmask = __ballot(in);
                if (__popcll(tmask) > maxCnt) { maxMask = tmask; maxCnt = __popcll(tmask); m = tm; d = td; }
                else break;
            }
            const double dp = gdot(me - m, d);
            const int e1 = psl_first_arg(dp, maxMask, true), e2 = psl_first_arg(dp, maxMask, false);
            outA = glue_pos(S, e1);
            outB = glue_pos(S, e2);
        }
        if (gnorm(outA - outB) > 0.02 && lane == 0) {  // (:731-748)
            const float e0 = (float)(outB.x - outA.x), e1f = (float)(outB.y - outA.y), e2f = (float)(outB.z - outA.z);
            const float magn = sqrtf(e0 * e0 + e1f * e1f + e2f * e2f);
            L3[6 * i] = outA.x; L3[6 * i + 1] = outA.y; L3[6 * i + 2] = outA.z;
            L3[6 * i + 3] = outB.x; L3[6 * i + 4] = outB.y; L3[6 * i + 5] = outB.z;
            LE[3 * i] = e0 / magn; LE[3 * i + 1] = e1f / magn; LE[3 * i + 2] = e2f / magn;
        }
        __builtin_amdgcn_wave_barrier();  // S.pos / S.idx are rewritten by the next line
    }
}

// convertFansToKeyLines + the plane loop of ExtractLSD.  One wave per frame: phase 1 evaluates the fans in parallel and
// compacts the crossings in fan order; phase 2 (planes) is a short serial chain through Frame::OldPlane.
__global__ __launch_bounds__(64) void k_fans_planes(const PslKeyLine* __restrict__ kls, int kl_stride, const float* __restrict__ fans,
                                                     int fan_stride, const int32_t* __restrict__ nfans_arr, int nfans_single,
                                                     const double* __restrict__ lines3d, const float* __restrict__ lineEq,
                                                     int32_t* __restrict__ pair, float* __restrict__ xy, double* __restrict__ cross,
                                                     int32_t* __restrict__ nint_out, int int_cap, float* __restrict__ planes,
                                                     double* __restrict__ normals, int32_t* __restrict__ lineNo, double* __restrict__ cross3d,
                                                     double* __restrict__ cross2d, double* __restrict__ le_l, int32_t* __restrict__ nplanes_out,
                                                     int plane_cap) {
    const int frame = blockIdx.x, lane = threadIdx.x;
    const int nf = min(nfans_arr ? nfans_arr[frame] : nfans_single, fan_stride);
    const PslKeyLine* K = kls + (size_t)frame * kl_stride;
    const float* F = fans + (size_t)frame * fan_stride * 4;
    const double* L3 = lines3d + (size_t)frame * kl_stride * 6;
    const float* LE = lineEq + (size_t)frame * kl_stride * 3;
    int32_t* PR = pair + (size_t)frame * int_cap * 2;
    float* XY = xy + (size_t)frame * int_cap * 2;
    double* CR = cross + (size_t)frame * int_cap * 3;
    double* LL = le_l + (size_t)frame * int_cap * 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int k = 0;
    for (int base = 0; base < nf; base += 64) {
        const int i = base + lane;
        bool ok = false;
        float x = 0, y = 0;
        int i1 = 0, i2 = 0;
        GlueP3 cp = {0, 0, 0};
        if (i < nf) {
            x = F[4 * i]; y = F[4 * i + 1];
            i1 = (int)F[4 * i + 2]; i2 = (int)F[4 * i + 3];
            if (i1 >= 0 && i1 < kl_stride && i2 >= 0 && i2 < kl_stride) {
                const double* A1 = L3 + 6 * i1;
                const double* A2 = L3 + 6 * i2;
                const GlueP3 p1 = {A1[0], A1[1], A1[2]}, p2 = {A2[0], A2[1], A2[2]}, e1 = {A1[3], A1[4], A1[5]}, e2 = {A2[3], A2[4], A2[5]};
                const GlueP3 d1 = e1 - p1, d2 = e2 - p2, p2p1 = p1 - p2;
                const double d11 = gdot(d1, d1), d12 = gdot(d1, d2), d22 = gdot(d2, d2), pd1 = gdot(p2p1, d1), pd2 = gdot(p2p1, d2);
                const double det = d11 * (-d22) - (-d12) * d12;
                if (det != 0) {
                    const double b0 = -pd1, b1 = -pd2;
                    const double x0 = (b0 * (-d22) - (-d12) * b1) / det, x1 = (d11 * b1 - d12 * b0) / det;
                    const GlueP3 root1 = p1 + d1 * x0, root2 = p2 + d2 * x1;
                    cp = (root1 + root2) * 0.5;
                    const GlueP3 mid_x = (p1 + p2) * 0.5, mid_y = (e1 + e2) * 0.5;
                    const double distmid = gnorm(mid_x - mid_y) * 2;
                    const double n1 = __dsqrt_rn(gdot(p1, p1) + gdot(e1, e1)), n2 = __dsqrt_rn(gdot(p2, p2) + gdot(e2, e2));
                    ok = distmid < n1 + n2 && gnorm(cp) > 2.220446049250313e-16;
                }
            }
        }
        const unsigned long long m = __ballot(ok);
        if (ok) {
            const int pos = k + __popcll(m & lt);
            if (pos < int_cap) {
                PR[2 * pos] = i1; PR[2 * pos + 1] = i2;
                XY[2 * pos] = x; XY[2 * pos + 1] = y;
                CR[3 * pos] = cp.x; CR[3 * pos + 1] = cp.y; CR[3 * pos + 2] = cp.z;
            }
        }
        k += __popcll(m);
    }
    const int nint = min(k, int_cap);
    if (lane == 0) nint_out[frame] = k;
    __threadfence();
    __builtin_amdgcn_wave_barrier();
    // mvle_l for every crossing (:517-527), parallel
    for (int i = lane; i < nint; i += 64) {
        for (int s = 0; s < 2; ++s) {
            const PslKeyLine& L = K[PR[2 * i + s]];
            const double sx = L.startPointX, sy = L.startPointY, ex = L.endPointX, ey = L.endPointY;
            const double c0 = sy * 1.0 - 1.0 * ey, c1 = 1.0 * ex - sx * 1.0, c2 = sx * ey - sy * ex;
            const double nrm = __dsqrt_rn(c0 * c0 + c1 * c1);
            LL[6 * i + 3 * s] = c0 / nrm; LL[6 * i + 3 * s + 1] = c1 / nrm; LL[6 * i + 3 * s + 2] = c2 / nrm;
        }
    }
    // planes (:528-659): serial through OldPlane; lane 0
    if (lane == 0) {
        float* PL = planes + (size_t)frame * plane_cap * 4;
        double* NN = normals + (size_t)frame * plane_cap * 3;
        int32_t* LN = lineNo + (size_t)frame * plane_cap * 2;
        double* C3 = cross3d + (size_t)frame * plane_cap * 3;
        double* C2 = cross2d + (size_t)frame * plane_cap * 2;
        int np = 0;
        for (int i = 0; i < nint; ++i) {
            const int l1 = PR[2 * i], l2 = PR[2 * i + 1];
            const float* q1 = LE + 3 * l1;
            const float* q2 = LE + 3 * l2;
            if (q1[0] == 0 && q1[1] == 0 && q1[2] == 0) continue;
            if (q2[0] == 0 && q2[1] == 0 && q2[2] == 0) continue;
            const double* A1 = L3 + 6 * l1;
            const double* A2 = L3 + 6 * l2;
            if (A1[0] == 0 && A1[1] == 0 && A1[2] == 0 && A1[3] == 0 && A1[4] == 0 && A1[5] == 0) continue;
            if (A2[0] == 0 && A2[1] == 0 && A2[2] == 0 && A2[3] == 0 && A2[4] == 0 && A2[5] == 0) continue;
            float pn[3] = {q1[1] * q2[2] - q1[2] * q2[1], q1[2] * q2[0] - q1[0] * q2[2], q1[0] * q2[1] - q1[1] * q2[0]};
            const float nr = sqrtf(pn[0] * pn[0] + pn[1] * pn[1] + pn[2] * pn[2]);
            pn[0] = pn[0] / nr; pn[1] = pn[1] / nr; pn[2] = pn[2] / nr;
            const double nx = pn[0], ny = pn[1], nz = pn[2];
            const double* c3 = CR + 3 * i;
            const float d1 = (float)(nx * A1[0] + ny * A1[1] + nz * A1[2]);
            const float d2 = (float)(nx * A1[3] + ny * A1[4] + nz * A1[5]);
            const float d3 = (float)(nx * A2[0] + ny * A2[1] + nz * A2[2]);
            const float d4 = (float)(nx * A2[3] + ny * A2[4] + nz * A2[5]);
            const float d5 = (float)(nx * c3[0] + ny * c3[1] + nz * c3[2]);
            float dmin = 10000, dmax = -10000;
            dmin = dmin < d1 ? dmin : d1; dmin = dmin < d2 ? dmin : d2; dmin = dmin < d3 ? dmin : d3; dmin = dmin < d4 ? dmin : d4;
            dmax = dmax > d1 ? dmax : d1; dmax = dmax > d2 ? dmax : d2; dmax = dmax > d3 ? dmax : d3; dmax = dmax > d4 ? dmax : d4;
            dmin = dmin < d5 ? dmin : d5;
            dmax = dmax > d5 ? dmax : d5;
            if ((double)(dmax - dmin) > 0.05) continue;
            const float planeDis = -(d1 + d2 + d3 + d4 + d5) / 5;
            float pl[4] = {(float)nx, (float)ny, (float)nz, planeDis};
            double nn[3] = {nx, ny, nz};
            if (pl[3] < 0) {
                pl[0] = -pl[0]; pl[1] = -pl[1]; pl[2] = -pl[2]; pl[3] = -pl[3];
                nn[0] = -nn[0]; nn[1] = -nn[1]; nn[2] = -nn[2];
            }
            bool old = false;  // Frame::OldPlane (:474-488)
            for (int kk = 0; kk < np && kk < plane_cap; ++kk) {
                const float* pli = PL + 4 * kk;
                const float dd = pl[3] - pli[3];
                const float angle = pl[0] * pli[0] + pl[1] * pli[1] + pl[2] * pli[2];
                if ((double)dd > 0.2 || (double)dd < -0.2) continue;
                if ((double)angle < 0.9397 && (double)angle > -0.9397) continue;
                old = true;
                break;
            }
            if (old) continue;
            if (np < plane_cap) {
                for (int c = 0; c < 4; ++c) PL[4 * np + c] = pl[c];
                for (int c = 0; c < 3; ++c) NN[3 * np + c] = nn[c];
                LN[2 * np] = l1; LN[2 * np + 1] = l2;
                for (int c = 0; c < 3; ++c) C3[3 * np + c] = c3[c];
                C2[2 * np] = XY[2 * i]; C2[2 * np + 1] = XY[2 * i + 1];
            }
            ++np;
        }
        nplanes_out[frame] = np;
    }
}

// ---------------------------------------------------------------------------------------------
struct pslfe_glue {
    pslfe_ctx* ctx = nullptr;
    int max_lines = 0, max_fans = 0, max_batch = 0, int_cap = 0, plane_cap = 0;
    int last_nframes = 0;
    // outputs, per frame
    double* d_lines3d = nullptr;  // [F][max_lines][6]
    float* d_lineEq = nullptr;    // [F][max_lines][3]
    int32_t* d_pair = nullptr;    // [F][int_cap][2]
    float* d_xy = nullptr;        // [F][int_cap][2]
    double* d_cross = nullptr;    // [F][int_cap][3]
    double* d_le_l = nullptr;     // [F][int_cap][6]
    int32_t* d_nint = nullptr;    // [F]
    float* d_planes = nullptr;    // [F][plane_cap][4]
    double* d_normals = nullptr;  // [F][plane_cap][3]
    int32_t* d_lineNo = nullptr;  // [F][plane_cap][2]
    double* d_cross3d = nullptr;  // [F][plane_cap][3]
    double* d_cross2d = nullptr;  // [F][plane_cap][2]
    int32_t* d_nplanes = nullptr; // [F]
    // staging for the host-pointer entry point (one frame)
    PslKeyLine* d_kls = nullptr;
    float* d_fans = nullptr;
    float* d_depth = nullptr;
    size_t depth_cap = 0;
};

static int glue_run(pslfe_glue* g, int nframes, const PslKeyLine* d_kls, int kl_stride, const int32_t* d_nkl, int nkl_single,
                    const float* d_fans, int fan_stride, const int32_t* d_nfans, int nfans_single, const float* d_depth, int w, int h,
                    int dstride, size_t dframe, const PslCamera* cam, uint32_t seed0) {
    hipStream_t st = g->ctx->stream;
    {
        PSL_STAGE_BEGIN(g->ctx, "line.good");
        k_line_good<<<nframes, 64, 0, st>>>(d_kls, kl_stride, d_nkl, nkl_single, d_depth, w, h, dstride, dframe, *cam, seed0, g->d_lines3d,
                                           g->d_lineEq);
        PSL_STAGE_END(g->ctx, "line.good");
    }
    {
        PSL_STAGE_BEGIN(g->ctx, "line.planes");
        k_fans_planes<<<nframes, 64, 0, st>>>(d_kls, kl_stride, d_fans, fan_stride, d_nfans, nfans_single, g->d_lines3d, g->d_lineEq, g->d_pair,
                                             g->d_xy, g->d_cross, g->d_nint, g->int_cap, g->d_planes, g->d_normals, g->d_lineNo, g->d_cross3d,
                                             g->d_cross2d, g->d_le_l, g->d_nplanes, g->plane_cap);
        PSL_STAGE_END(g->ctx, "line.planes");
    }
    PSL_HIP(hipGetLastError());
    g->last_nframes = nframes;
    return PSLFE_OK;
}

extern "C" {

void pslfe_glue_destroy(pslfe_glue* g) {
    if (!g) return;
    hipSetDevice(g->ctx->device);
    hipStreamSynchronize(g->ctx->stream);
    hipFree(g->d_lines3d); hipFree(g->d_lineEq); hipFree(g->d_pair); hipFree(g->d_xy); hipFree(g->d_cross); hipFree(g->d_le_l);
    hipFree(g->d_nint); hipFree(g->d_planes); hipFree(g->d_normals); hipFree(g->d_lineNo); hipFree(g->d_cross3d); hipFree(g->d_cross2d);
    hipFree(g->d_nplanes); hipFree(g->d_kls); hipFree(g->d_fans); hipFree(g->d_depth);
    delete g;
}

int pslfe_glue_create(pslfe_ctx* ctx, int max_lines, int max_fans, int max_batch, pslfe_glue** out) {
    PSL_REQUIRE(ctx && out, PSLFE_E_INVALID, "pslfe_glue_create: NULL argument");
    *out = nullptr;
    PSL_REQUIRE(max_lines >= 1 && max_fans >= 1 && max_batch >= 1, PSLFE_E_INVALID, "pslfe_glue_create: max_lines %d max_fans %d max_batch %d",
                max_lines, max_fans, max_batch);
    PSL_HIP(hipSetDevice(ctx->device));
    pslfe_glue* g = new pslfe_glue();
    g->ctx = ctx; g->max_lines = max_lines; g->max_fans = max_fans; g->max_batch = max_batch;
    g->int_cap = max_fans; g->plane_cap = max_fans;
    const size_t F = (size_t)max_batch, L = (size_t)max_lines, I = (size_t)g->int_cap, Pn = (size_t)g->plane_cap;
    hipError_t e = hipSuccess;
    auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 1); };
    A((void**)&g->d_lines3d, F * L * 6 * sizeof(double));
    A((void**)&g->d_lineEq, F * L * 3 * sizeof(float));
    A((void**)&g->d_pair, F * I * 2 * sizeof(int32_t));
    A((void**)&g->d_xy, F * I * 2 * sizeof(float));
    A((void**)&g->d_cross, F * I * 3 * sizeof(double));
    A((void**)&g->d_le_l, F * I * 6 * sizeof(double));
    A((void**)&g->d_nint, F * sizeof(int32_t));
    A((void**)&g->d_planes, F * Pn * 4 * sizeof(float));
    A((void**)&g->d_normals, F * Pn * 3 * sizeof(double));
    A((void**)&g->d_lineNo, F * Pn * 2 * sizeof(int32_t));
    A((void**)&g->d_cross3d, F * Pn * 3 * sizeof(double));
    A((void**)&g->d_cross2d, F * Pn * 2 * sizeof(double));
    A((void**)&g->d_nplanes, F * sizeof(int32_t));
    A((void**)&g->d_kls, L * sizeof(PslKeyLine));
    A((void**)&g->d_fans, (size_t)max_fans * 4 * sizeof(float));
    if (e != hipSuccess) {
        pslfe_set_error("pslfe_glue_create: hipMalloc failed: %s", hipGetErrorString(e));
        pslfe_glue_destroy(g);
        return PSLFE_E_HIP;
    }
    *out = g;
    return PSLFE_OK;
}

int pslfe_glue_run(pslfe_glue* g, const PslKeyLine* kls, int nlines, const float* fans, int nfans, const float* depth, int width, int height,
                   int depth_stride, const PslCamera* cam, uint32_t seed) {
    PSL_REQUIRE(g && cam && depth && (nlines == 0 || kls) && (nfans == 0 || fans), PSLFE_E_INVALID, "pslfe_glue_run: NULL argument");
    PSL_REQUIRE(nlines >= 0 && nlines <= g->max_lines, PSLFE_E_CAPACITY, "pslfe_glue_run: %d lines, capacity %d", nlines, g->max_lines);
    PSL_REQUIRE(nfans >= 0 && nfans <= g->max_fans, PSLFE_E_CAPACITY, "pslfe_glue_run: %d fans, capacity %d", nfans, g->max_fans);
    PSL_REQUIRE(width > 0 && height > 0 && depth_stride >= width, PSLFE_E_INVALID, "pslfe_glue_run: depth %dx%d stride %d", width, height, depth_stride);
    for (int i = 0; i < nfans; ++i)
        PSL_REQUIRE(fans[4 * i + 2] >= 0 && fans[4 * i + 2] < nlines && fans[4 * i + 3] >= 0 && fans[4 * i + 3] < nlines, PSLFE_E_INVALID,
                    "pslfe_glue_run: fan %d refers to a line outside 0..%d", i, nlines - 1);
    PSL_HIP(hipSetDevice(g->ctx->device));
    hipStream_t st = g->ctx->stream;
    const size_t need = (size_t)height * depth_stride;
    if (need > g->depth_cap) {
        PSL_HIP(hipStreamSynchronize(st));
        hipFree(g->d_depth); g->d_depth = nullptr; g->depth_cap = 0;
        PSL_HIP(hipMalloc((void**)&g->d_depth, need * sizeof(float)));
        g->depth_cap = need;
    }
    PSL_HIP(hipMemcpyAsync(g->d_depth, depth, need * sizeof(float), hipMemcpyHostToDevice, st));
    if (nlines) PSL_HIP(hipMemcpyAsync(g->d_kls, kls, (size_t)nlines * sizeof(PslKeyLine), hipMemcpyHostToDevice, st));
    if (nfans) PSL_HIP(hipMemcpyAsync(g->d_fans, fans, (size_t)nfans * 4 * sizeof(float), hipMemcpyHostToDevice, st));
    PSL_HIP(hipStreamSynchronize(st));
    // strides of one frame: lines3d / lineEq are indexed with kl_stride = max_lines
    return glue_run(g, 1, g->d_kls, g->max_lines, nullptr, nlines, g->d_fans, g->max_fans, nullptr, nfans, g->d_depth, width, height,
                    depth_stride, 0, cam, seed);
}

int pslfe_glue_run_batch_device(pslfe_glue* g, int nframes, const PslKeyLine* d_kls, int kl_stride, const int32_t* d_nkl, const float* d_fans,
                                int fan_stride, const int32_t* d_nfans, const float* d_depth, int width, int height, const PslCamera* cam,
                                uint32_t seed0) {
    PSL_REQUIRE(g && d_kls && d_nkl && d_fans && d_nfans && d_depth && cam, PSLFE_E_INVALID, "pslfe_glue_run_batch_device: NULL argument");
    PSL_REQUIRE(nframes >= 1 && nframes <= g->max_batch, PSLFE_E_CAPACITY, "pslfe_glue_run_batch_device: %d frames, capacity %d", nframes, g->max_batch);
    PSL_REQUIRE(kl_stride >= 1 && kl_stride <= g->max_lines && fan_stride >= 1 && fan_stride <= g->max_fans, PSLFE_E_CAPACITY,
                "pslfe_glue_run_batch_device: strides %d / %d exceed capacities %d / %d", kl_stride, fan_stride, g->max_lines, g->max_fans);
    PSL_REQUIRE(kl_stride == g->max_lines, PSLFE_E_INVALID, "pslfe_glue_run_batch_device: kl_stride %d must equal max_lines %d", kl_stride, g->max_lines);
    PSL_HIP(hipSetDevice(g->ctx->device));
    return glue_run(g, nframes, d_kls, kl_stride, d_nkl, 0, d_fans, fan_stride, d_nfans, 0, d_depth, width, height, width, (size_t)width * height, cam, seed0);
}

int pslfe_glue_fetch(pslfe_glue* g, int frame, int nlines, double* lines3d, float* lineEq, int32_t* pair, float* xy, double* cross, double* le_l,
                     int int_cap, int* nint, float* planes, double* normals, int32_t* lineNo, double* cross3d, double* cross2d, int plane_cap,
                     int* nplanes) {
    PSL_REQUIRE(g && nint && nplanes, PSLFE_E_INVALID, "pslfe_glue_fetch: NULL argument");
    PSL_REQUIRE(g->last_nframes > 0 && frame >= 0 && frame < g->last_nframes, PSLFE_E_STATE, "pslfe_glue_fetch: frame %d of %d", frame, g->last_nframes);
    PSL_REQUIRE(nlines >= 0 && nlines <= g->max_lines, PSLFE_E_CAPACITY, "pslfe_glue_fetch: %d lines, capacity %d", nlines, g->max_lines);
    PSL_HIP(hipSetDevice(g->ctx->device));
    hipStream_t st = g->ctx->stream;
    const size_t f = (size_t)frame;
    // counts first (8 bytes through the pinned staging buffer), then every slice into the staging buffer in one go, one wait, host copies
    const size_t L = (size_t)g->max_lines, I = (size_t)g->int_cap, Pn = (size_t)g->plane_cap;
    char* hs = psl_host_stage(g->ctx, 64);
    PSL_REQUIRE(hs, PSLFE_E_HIP, "pslfe_glue_fetch: no pinned staging memory (hipHostMalloc)");
    PSL_HIP(hipMemcpyAsync(hs, g->d_nint + f, sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipMemcpyAsync(hs + 4, g->d_nplanes + f, sizeof(int), hipMemcpyDeviceToHost, st));
    PSL_HIP(hipStreamSynchronize(st));
    int ni = 0, np = 0;
    memcpy(&ni, hs, 4); memcpy(&np, hs + 4, 4);
    *nint = ni; *nplanes = np;
    PSL_REQUIRE(ni <= g->int_cap && np <= g->plane_cap, PSLFE_E_CAPACITY, "pslfe_glue_fetch: %d crossings / %d planes exceed the handle's capacity %d", ni, np, g->int_cap);
    PSL_REQUIRE(ni <= int_cap && np <= plane_cap, PSLFE_E_CAPACITY, "pslfe_glue_fetch: %d crossings / %d planes, capacities %d / %d", ni, np, int_cap, plane_cap);
    struct Piece { void* dst; const void* src; size_t bytes; };
    const Piece pc[11] = {
        {lines3d, g->d_lines3d + f * L * 6, (size_t)nlines * 6 * sizeof(double)}, {lineEq, g->d_lineEq + f * L * 3, (size_t)nlines * 3 * sizeof(float)},
        {pair, g->d_pair + f * I * 2, (size_t)ni * 2 * sizeof(int32_t)},           {xy, g->d_xy + f * I * 2, (size_t)ni * 2 * sizeof(float)},
        {cross, g->d_cross + f * I * 3, (size_t)ni * 3 * sizeof(double)},          {le_l, g->d_le_l + f * I * 6, (size_t)ni * 6 * sizeof(double)},
        {planes, g->d_planes + f * Pn * 4, (size_t)np * 4 * sizeof(float)},        {normals, g->d_normals + f * Pn * 3, (size_t)np * 3 * sizeof(double)},
        {lineNo, g->d_lineNo + f * Pn * 2, (size_t)np * 2 * sizeof(int32_t)},      {cross3d, g->d_cross3d + f * Pn * 3, (size_t)np * 3 * sizeof(double)},
        {cross2d, g->d_cross2d + f * Pn * 2, (size_t)np * 2 * sizeof(double)}};
    size_t total = 0;
    for (const Piece& q : pc) if (q.dst && q.bytes) total += psl_align_up(q.bytes, 16);
    if (total) {
        hs = psl_host_stage(g->ctx, total);
        PSL_REQUIRE(hs, PSLFE_E_HIP, "pslfe_glue_fetch: no pinned staging memory (hipHostMalloc)");
        size_t o = 0;
        for (const Piece& q : pc)
            if (q.dst && q.bytes) { PSL_HIP(hipMemcpyAsync(hs + o, q.src, q.bytes, hipMemcpyDeviceToHost, st)); o += psl_align_up(q.bytes, 16); }
        PSL_HIP(hipStreamSynchronize(st));
        o = 0;
        for (const Piece& q : pc)
            if (q.dst && q.bytes) { memcpy(q.dst, hs + o, q.bytes); o += psl_align_up(q.bytes, 16); }
    }
    return PSLFE_OK;
}

int pslfe_glue_planes_device(pslfe_glue* g, const float** d_planes, const int32_t** d_plane_lines, const int32_t** d_plane_counts, int* plane_stride) {
    PSL_REQUIRE(g, PSLFE_E_INVALID, "pslfe_glue_planes_device: glue is NULL");
    PSL_REQUIRE(g->last_nframes > 0, PSLFE_E_STATE, "pslfe_glue_planes_device: no batch processed yet");
    if (d_planes) *d_planes = g->d_planes;
    if (d_plane_lines) *d_plane_lines = g->d_lineNo;
    if (d_plane_counts) *d_plane_counts = g->d_nplanes;
    if (plane_stride) *plane_stride = g->plane_cap;
    return PSLFE_OK;
}

}  // extern "C"
