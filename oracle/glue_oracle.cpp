// ORACLE — test infrastructure only (see orb_oracle.cpp header). CPU restatement of the RGB-D line glue of the Frame
// constructor, sequential exactly as the reference runs it:
//   Frame::isLineGood                         src/Frame.cc:662-750
//   LINEextractor::compPt3dCov                add_src/LineExtractor.cpp:40-93  (depthStdDev :27-38)
//   LINEextractor::verify3dLine               :95-161
//   LINEextractor::computeLine3d_svd          :163-185
//   LINEextractor::mah_dist3d_pt_line         :187-214
//   LINEextractor::extract3dline_mahdist      :216-322 ; random_unique add_inc/LineExtractor.h:23-37 ; projPt3d2Ln3d :199-207
//   Frame::convertFansToKeyLines              src/Frame.cc:426-472 ; Frame_shortestDistance :381-424
//   plane from a pair of 3-D lines            src/Frame.cc:505-660 ; Frame::OldPlane :474-488
// Third-party arithmetic that is not under /root/reference, restated from the published algorithms:
//   * rand(): glibc's TYPE_3 additive feedback generator (r[i] = r[i-3] + r[i-31], 310 outputs discarded, result >> 1),
//     seeded as srand(seed) immediately before isLineGood (convention H7: the reference never seeds, so its stream
//     position depends on the process history).
//   * cv::SVD (OpenCV 3.x lapack.cpp JacobiSVDImpl_, f64: one-sided Jacobi on the rows of A^T, eps = 10 * DBL_EPSILON,
//     max(m, 30) sweeps, singular values sorted descending by selection).  std::hypot is taken as sqrt(p*p + b*b); the
//     zero-singular-value randomisation is omitted (it never reaches the vectors used here).  Only the SIGN of the first
//     right singular vector and the whitening D*U^T matter downstream; the whitening enters through Mahalanobis
//     distances, which do not depend on the choice of U.
//   * cv::gemm 3x3: each element as a[i][0]*b[0][j] + a[i][1]*b[1][j] + a[i][2]*b[2][j].
//   * Eigen colPivHouseholderQr().solve on the 2x2 system of Frame_shortestDistance: solved by Cramer's rule; the
//     crossing point is compared with a relative tolerance in the tests.
// Conventions: a line shorter than 1 px (numSmp == 0, 0/0 in the reference) is skipped; Frame_shortestDistance falling
// off its end without a return value (undefined behaviour upstream) counts as "no crossing".
// PARITY UNPINNED: the reference holds no fixtures for these functions.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "psl_oracle.h"

namespace {

struct P3 { double x, y, z; };
inline P3 operator+(const P3& a, const P3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline P3 operator-(const P3& a, const P3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline P3 operator*(const P3& a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(const P3& a, const P3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double norm(const P3& a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }

// glibc rand() / srand(), TYPE_3
struct GlibcRand {
    uint32_t ring[34];
    int k;
    explicit GlibcRand(uint32_t seed) {
        int32_t s[34];
        s[0] = seed == 0 ? 1 : (int32_t)seed;
        for (int i = 1; i < 31; ++i) {
            const long hi = s[i - 1] / 127773, lo = s[i - 1] % 127773;
            long word = 16807 * lo - 2836 * hi;
            if (word < 0) word += 2147483647;
            s[i] = (int32_t)word;
        }
        for (int i = 31; i < 34; ++i) s[i] = s[i - 31];
        for (int i = 0; i < 34; ++i) ring[i] = (uint32_t)s[i];
        k = 34;
        for (int i = 0; i < 310; ++i) step();
    }
    uint32_t step() {  // r[k] = r[k-31] + r[k-3] on a ring of 34
        const uint32_t v = ring[(k - 31) % 34] + ring[(k - 3) % 34];
        ring[k % 34] = v;
        ++k;
        return v;
    }
    int next() { return (int)(step() >> 1); }
};

// OpenCV JacobiSVDImpl_<double>: At has n rows of length m (n <= 3, m <= 32), Vt n x n.  On return the rows of At are
// the left singular vectors times nothing (normalised), W descending.
void jacobi_svd(double At[3][32], double W[3], double Vt[3][3], int m, int n) {
    const double eps = 2.220446049250313e-16 * 10;
    const int max_iter = std::max(m, 30);
    double Wd[3];
    for (int i = 0; i < n; ++i) {
        double sd = 0;
        for (int k = 0; k < m; ++k) sd += At[i][k] * At[i][k];
        Wd[i] = sd;
        for (int k = 0; k < n; ++k) Vt[i][k] = 0;
        Vt[i][i] = 1;
    }
    for (int iter = 0; iter < max_iter; ++iter) {
        bool changed = false;
        for (int i = 0; i < n - 1; ++i)
            for (int j = i + 1; j < n; ++j) {
                double a = Wd[i], p = 0, b = Wd[j];
                for (int k = 0; k < m; ++k) p += At[i][k] * At[j][k];
                if (std::fabs(p) <= eps * std::sqrt(a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = std::sqrt(p * p + beta * beta);
                double c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = std::sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = std::sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (int k = 0; k < m; ++k) {
                    const double t0 = c * At[i][k] + s * At[j][k];
                    const double t1 = -s * At[i][k] + c * At[j][k];
                    At[i][k] = t0; At[j][k] = t1;
                    a += t0 * t0; b += t1 * t1;
                }
                Wd[i] = a; Wd[j] = b;
                changed = true;
                for (int k = 0; k < n; ++k) {
                    const double t0 = c * Vt[i][k] + s * Vt[j][k];
                    const double t1 = -s * Vt[i][k] + c * Vt[j][k];
                    Vt[i][k] = t0; Vt[j][k] = t1;
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < n; ++i) {
        double sd = 0;
        for (int k = 0; k < m; ++k) sd += At[i][k] * At[i][k];
        Wd[i] = std::sqrt(sd);
    }
    for (int i = 0; i < n - 1; ++i) {
        int j = i;
        for (int k = i + 1; k < n; ++k)
            if (Wd[j] < Wd[k]) j = k;
        if (i != j) {
            std::swap(Wd[i], Wd[j]);
            for (int k = 0; k < m; ++k) std::swap(At[i][k], At[j][k]);
            for (int k = 0; k < n; ++k) std::swap(Vt[i][k], Vt[j][k]);
        }
    }
    for (int i = 0; i < n; ++i) {
        W[i] = Wd[i];
        const double s = Wd[i] > 2.2250738585072014e-308 ? 1 / Wd[i] : 0.;
        for (int k = 0; k < m; ++k) At[i][k] *= s;
    }
}

struct RPt {
    P3 pos;
    double DU[9];
};

double depth_std_dev(double d) {
    const double c1 = 0.00273, c2 = 0.00074, c3 = -0.00058;
    return c1 * d * d + c2 * d + c3;
}

void mul33(const double a[3][3], const double b[3][3], double o[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) o[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
}

RPt comp_pt3d_cov(const P3& pt, double f) {
    RPt rp;
    rp.pos = pt;
    const double J0[3][3] = {{pt.z / f, 0, pt.x / pt.z}, {0, pt.z / f, pt.y / pt.z}, {0, 0, 1}};
    const double sd = depth_std_dev(pt.z);
    const double G[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, sd * sd}};
    double JG[3][3], Jt[3][3], cov[3][3];
    mul33(J0, G, JG);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Jt[i][j] = J0[j][i];
    mul33(JG, Jt, cov);
    // cv::SVD(cov0): m == n == 3, At = cov0^T; U = (rotated, normalised At)^T, so U^T = At
    double At[3][32], W[3], Vt[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) At[i][j] = cov[j][i];
    jacobi_svd(At, W, Vt, 3, 3);
    for (int r = 0; r < 3; ++r) {
        const double d = 1 / std::sqrt(W[r]);
        // du = D * U^T with D = diag(1/W_sqrt), as the gemm computes it: d * U^T[r][c] (+ 0 * ... terms)
        for (int c = 0; c < 3; ++c) rp.DU[3 * r + c] = d * At[r][c];
    }
    return rp;
}

double mah_dist3d_pt_line(const RPt& pt, const P3& q1, const P3& q2) {
    const double xa = q1.x, ya = q1.y, za = q1.z, xb = q2.x, yb = q2.y, zb = q2.z;
    const double c1 = pt.DU[0], c2 = pt.DU[1], c3 = pt.DU[2], c4 = pt.DU[3], c5 = pt.DU[4], c6 = pt.DU[5], c7 = pt.DU[6], c8 = pt.DU[7], c9 = pt.DU[8];
    const double x1 = pt.pos.x, x2 = pt.pos.y, x3 = pt.pos.z;
    const double term1 = ((c1 * (x1 - xa) + c2 * (x2 - ya) + c3 * (x3 - za)) * (c4 * (x1 - xb) + c5 * (x2 - yb) + c6 * (x3 - zb)) -
                          (c4 * (x1 - xa) + c5 * (x2 - ya) + c6 * (x3 - za)) * (c1 * (x1 - xb) + c2 * (x2 - yb) + c3 * (x3 - zb))),
                 term2 = ((c1 * (x1 - xa) + c2 * (x2 - ya) + c3 * (x3 - za)) * (c7 * (x1 - xb) + c8 * (x2 - yb) + c9 * (x3 - zb)) -
                          (c7 * (x1 - xa) + c8 * (x2 - ya) + c9 * (x3 - za)) * (c1 * (x1 - xb) + c2 * (x2 - yb) + c3 * (x3 - zb))),
                 term3 = ((c4 * (x1 - xa) + c5 * (x2 - ya) + c6 * (x3 - za)) * (c7 * (x1 - xb) + c8 * (x2 - yb) + c9 * (x3 - zb)) -
                          (c7 * (x1 - xa) + c8 * (x2 - ya) + c9 * (x3 - za)) * (c4 * (x1 - xb) + c5 * (x2 - yb) + c6 * (x3 - zb))),
                 term4 = (c1 * (x1 - xa) - c1 * (x1 - xb) + c2 * (x2 - ya) - c2 * (x2 - yb) + c3 * (x3 - za) - c3 * (x3 - zb)),
                 term5 = (c4 * (x1 - xa) - c4 * (x1 - xb) + c5 * (x2 - ya) - c5 * (x2 - yb) + c6 * (x3 - za) - c6 * (x3 - zb)),
                 term6 = (c7 * (x1 - xa) - c7 * (x1 - xb) + c8 * (x2 - ya) - c8 * (x2 - yb) + c9 * (x3 - za) - c9 * (x3 - zb));
    return std::sqrt((term1 * term1 + term2 * term2 + term3 * term3) / (term4 * term4 + term5 * term5 + term6 * term6));
}

P3 proj_pt3d_ln3d(const P3& P, const P3& mid, const P3& drct) {
    const P3 A = mid, B = mid + drct, AB = B - A, AP = P - A;
    return A + AB * (dot(AB, AP) / dot(AB, AB));
}

bool verify3d_line(const std::vector<P3>& pts, const P3& A, const P3& B) {
    const int nCells = 10;
    int cells[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const double ratio = 0.7;
    const int nPts = (int)pts.size();
    double minv = 100, maxv = -100;
    int idx1 = 0, idx2 = 0;
    for (int i = 0; i < nPts; ++i) {
        if (dot(pts[i] - A, B - A) < minv) { minv = dot(pts[i] - A, B - A); idx1 = i; }
        if (dot(pts[i] - A, B - A) > maxv) { maxv = dot(pts[i] - A, B - A); idx2 = i; }
    }
    const P3 C = proj_pt3d_ln3d(pts[idx1], (A + B) * 0.5, B - A);
    const P3 D = proj_pt3d_ln3d(pts[idx2], (A + B) * 0.5, B - A);
    const double cd = norm(D - C);
    if (cd < 0.0000000001) return false;
    for (int i = 0; i < nPts; ++i) {
        const double lambda = std::fabs(dot(pts[i] - C, D - C) / cd / cd);
        if (lambda >= 1) cells[nCells - 1] += 1;
        else cells[(unsigned int)std::floor(lambda * 10)] += 1;
    }
    double sum = 0;
    for (int i = 0; i < nCells; ++i)
        if (cells[i] > 0) sum = sum + 1;
    return sum / nCells > ratio;
}

void compute_line3d_svd(const std::vector<RPt>& pts, const std::vector<int>& idx, P3& mean, P3& drct) {
    const int n = (int)idx.size();
    mean = {0, 0, 0};
    for (int i = 0; i < n; ++i) mean = mean + pts[idx[i]].pos;
    mean = mean * (1.0 / n);
    // cv::SVD(P.t()) with P.t() n x 3: n >= 3 -> At = (P.t())^T (3 rows of length n), vt = V^T; n < 3 -> At = P.t() itself
    // (n rows of length 3), vt = the normalised rows of At
    double At[3][32], W[3], Vt[3][3];
    if (n >= 3) {
        for (int i = 0; i < n; ++i) {
            At[0][i] = pts[idx[i]].pos.x - mean.x; At[1][i] = pts[idx[i]].pos.y - mean.y; At[2][i] = pts[idx[i]].pos.z - mean.z;
        }
        jacobi_svd(At, W, Vt, n, 3);
        drct = {Vt[0][0], Vt[0][1], Vt[0][2]};
    } else {
        for (int i = 0; i < n; ++i) {
            At[i][0] = pts[idx[i]].pos.x - mean.x; At[i][1] = pts[idx[i]].pos.y - mean.y; At[i][2] = pts[idx[i]].pos.z - mean.z;
        }
        jacobi_svd(At, W, Vt, 3, n);
        drct = {At[0][0], At[0][1], At[0][2]};
    }
}

// returns A, B (zero when no line)
void extract3dline_mahdist(const std::vector<RPt>& pts, GlibcRand& rng, P3& outA, P3& outB) {
    const int np = (int)pts.size();
    const int maxIterNo = std::min(10, int(np * (np - 1) * 0.5));
    const double distThresh = 3.0;
    std::vector<int> indexes(np);
    for (int i = 0; i < np; ++i) indexes[i] = i;
    std::vector<int> maxInlierSet;
    P3 bestA = {0, 0, 0}, bestB = {0, 0, 0};
    for (int iter = 0; iter < maxIterNo; iter++) {
        std::vector<int> inlierSet;
        {  // random_unique(begin, end, 2)
            size_t left = indexes.size();
            size_t begin = 0;
            for (int num = 0; num < 2; ++num) {
                const size_t r = begin + (size_t)rng.next() % left;
                std::swap(indexes[begin], indexes[r]);
                ++begin;
                --left;
            }
        }
        const RPt& A = pts[indexes[0]];
        const RPt& B = pts[indexes[1]];
        if (norm(B.pos - A.pos) < 0.0000000001) continue;
        for (int i = 0; i < np; ++i)
            if (mah_dist3d_pt_line(pts[i], A.pos, B.pos) < distThresh) inlierSet.push_back(i);
        if (inlierSet.size() > maxInlierSet.size()) {
            std::vector<P3> inlierPts(inlierSet.size());
            for (size_t ii = 0; ii < inlierSet.size(); ++ii) inlierPts[ii] = pts[inlierSet[ii]].pos;
            if (verify3d_line(inlierPts, A.pos, B.pos)) {
                maxInlierSet = inlierSet;
                bestA = pts[indexes[0]].pos;
                bestB = pts[indexes[1]].pos;
            }
        }
        if (maxInlierSet.size() > np * 0.6) break;
    }
    outA = {0, 0, 0};
    outB = {0, 0, 0};
    if (maxInlierSet.size() >= 2) {
        P3 m = (bestA + bestB) * 0.5, d = bestB - bestA;
        while (true) {
            std::vector<int> tmpInlierSet;
            P3 tmp_m, tmp_d;
            compute_line3d_svd(pts, maxInlierSet, tmp_m, tmp_d);
            for (int i = 0; i < np; ++i)
                if (mah_dist3d_pt_line(pts[i], tmp_m, tmp_m + tmp_d) < distThresh) tmpInlierSet.push_back(i);
            if (tmpInlierSet.size() > maxInlierSet.size()) {
                maxInlierSet = tmpInlierSet;
                m = tmp_m;
                d = tmp_d;
            } else
                break;
        }
        double minv = 100, maxv = -100;
        int idx_end1 = 0, idx_end2 = 0;
        for (size_t i = 0; i < maxInlierSet.size(); ++i) {
            const double dproduct = dot(pts[maxInlierSet[i]].pos - m, d);
            if (dproduct < minv) { minv = dproduct; idx_end1 = (int)i; }
            if (dproduct > maxv) { maxv = dproduct; idx_end2 = (int)i; }
        }
        outA = pts[maxInlierSet[idx_end1]].pos;
        outB = pts[maxInlierSet[idx_end2]].pos;
    }
}

}  // namespace

extern "C" {

int pso_glibc_rand(uint32_t seed, int n, int32_t* out) {  // the first n values of rand() after srand(seed)
    GlibcRand g(seed);
    for (int i = 0; i < n; ++i) out[i] = g.next();
    return n;
}

// lines3d: [n][6] (start, end; zeros when the line is not good); lineEq: [n][3] (-1 when not good).  cam = fx, fy, cx, cy.
void pso_line_good(const PsoKeyLine* kls, int n, const float* depth, int cols, int rows, int dstride, const float* cam, uint32_t seed,
                   double* lines3d, float* lineEq) {
    const float fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3];
    const float invfx = 1.0f / fx, invfy = 1.0f / fy;
    GlibcRand rng(seed);
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 6; ++k) lines3d[6 * i + k] = 0.0;
        lineEq[3 * i] = lineEq[3 * i + 1] = lineEq[3 * i + 2] = -1.0f;
    }
    for (int i = 0; i < n; ++i) {
        const float spx = kls[i].startPointX, spy = kls[i].startPointY, epx = kls[i].endPointX, epy = kls[i].endPointY;
        const float dxf = spx - epx, dyf = spy - epy;
        const double len = std::sqrt((double)dxf * dxf + (double)dyf * dyf);
        const double numSmp = (double)std::min((int)len, 20);
        if (numSmp == 0) continue;  // convention: 0/0 upstream
        std::vector<P3> pts3d;
        for (int j = 0; j <= numSmp; ++j) {
            const double w1 = 1 - j / numSmp, w2 = j / numSmp;
            const float ax = (float)(spx * w1), ay = (float)(spy * w1), bx = (float)(epx * w2), by = (float)(epy * w2);
            const double ptx = (double)(ax + bx), pty = (double)(ay + by);
            if (ptx < 0 || pty < 0 || ptx >= cols || pty >= rows) continue;
            int row, col;
            if ((std::floor(ptx) == ptx) && (std::floor(pty) == pty)) {
                col = std::max(int(ptx - 1), 0);
                row = std::max(int(pty - 1), 0);
            } else {
                col = int(ptx);
                row = int(pty);
            }
            const float dv = depth[(size_t)row * dstride + col];
            if (dv <= 0.01) continue;
            P3 p;
            p.z = dv;
            p.x = (col - cx) * p.z * invfx;
            p.y = (row - cy) * p.z * invfy;
            pts3d.push_back(p);
        }
        if (pts3d.size() < 5) continue;
        std::vector<RPt> rnd(pts3d.size());
        for (size_t j = 0; j < pts3d.size(); ++j) rnd[j] = comp_pt3d_cov(pts3d[j], (double)fx);
        P3 A, B;
        extract3dline_mahdist(rnd, rng, A, B);
        if (norm(A - B) > 0.02) {
            const float e0 = (float)(B.x - A.x), e1 = (float)(B.y - A.y), e2 = (float)(B.z - A.z);
            const float magn = std::sqrt(e0 * e0 + e1 * e1 + e2 * e2);
            lines3d[6 * i] = A.x; lines3d[6 * i + 1] = A.y; lines3d[6 * i + 2] = A.z;
            lines3d[6 * i + 3] = B.x; lines3d[6 * i + 4] = B.y; lines3d[6 * i + 5] = B.z;
            lineEq[3 * i] = e0 / magn; lineEq[3 * i + 1] = e1 / magn; lineEq[3 * i + 2] = e2 / magn;
        }
    }
}

// intersection_lines_plane: for every fan row (x, y, index1, index2) whose 3-D lines pass the test of
// Frame_shortestDistance: pair[k] = (index1, index2), xy[k] = (x, y), cross[k] = the 3-D crossing point.
int pso_fans_to_intersections(const float* fans, int nfans, const double* lines3d, int32_t* pair, float* xy, double* cross, int cap) {
    int k = 0;
    for (int i = 0; i < nfans; ++i) {
        const float x = fans[4 * i], y = fans[4 * i + 1];
        const int i1 = (int)fans[4 * i + 2], i2 = (int)fans[4 * i + 3];
        const double* L1 = lines3d + 6 * i1;
        const double* L2 = lines3d + 6 * i2;
        const P3 p1 = {L1[0], L1[1], L1[2]}, p2 = {L2[0], L2[1], L2[2]};
        const P3 e1 = {L1[3], L1[4], L1[5]}, e2 = {L2[3], L2[4], L2[5]};
        const P3 d1 = e1 - p1, d2 = e2 - p2, p2p1 = p1 - p2;
        const double d11 = dot(d1, d1), d12 = dot(d1, d2), d22 = dot(d2, d2), pd1 = dot(p2p1, d1), pd2 = dot(p2p1, d2);
        // A = [d11, -d12; d12, -d22], b = (-pd1, -pd2)
        const double det = d11 * (-d22) - (-d12) * d12;
        if (det == 0) continue;
        const double b0 = -pd1, b1 = -pd2;
        const double x0 = (b0 * (-d22) - (-d12) * b1) / det, x1 = (d11 * b1 - d12 * b0) / det;
        const P3 root1 = p1 + d1 * x0, root2 = p2 + d2 * x1;
        const P3 crosspoint = (root1 + root2) * 0.5;
        const P3 mid_x = (p1 + p2) * 0.5, mid_y = (e1 + e2) * 0.5;
        const double distmid = norm(mid_x - mid_y) * 2;
        const double n1 = std::sqrt(dot(p1, p1) + dot(e1, e1)), n2 = std::sqrt(dot(p2, p2) + dot(e2, e2));
        if (!(distmid < n1 + n2)) continue;  // falls off the end upstream
        if (!(norm(crosspoint) > 2.220446049250313e-16)) continue;
        if (k < cap) {
            pair[2 * k] = i1; pair[2 * k + 1] = i2;
            xy[2 * k] = x; xy[2 * k + 1] = y;
            cross[3 * k] = crosspoint.x; cross[3 * k + 1] = crosspoint.y; cross[3 * k + 2] = crosspoint.z;
        }
        ++k;
    }
    return k;
}

// The plane loop of Frame::ExtractLSD (src/Frame.cc:505-660).  Outputs per accepted plane: plane (4 floats), normal
// (3 doubles), line pair, 3-D and 2-D crossing points; le_l: the normalised 2-D line equations of every intersection
// (mvle_l, pushed before the filters), [nint][6].  Returns the number of planes.
int pso_planes_from_pairs(const PsoKeyLine* kls, const float* lineEq, const double* lines3d, const int32_t* pair, const float* xy,
                          const double* cross, int nint, float* planes, double* normals, int32_t* lineNo, double* cross3d, double* cross2d,
                          double* le_l, int cap) {
    int np = 0;
    for (int i = 0; i < nint; ++i) {
        const int l1 = pair[2 * i], l2 = pair[2 * i + 1];
        for (int s = 0; s < 2; ++s) {
            const PsoKeyLine& L = kls[s == 0 ? l1 : l2];
            // sp x ep with sp = (sx, sy, 1), ep = (ex, ey, 1), in double
            const double sx = L.startPointX, sy = L.startPointY, ex = L.endPointX, ey = L.endPointY;
            const double c0 = sy * 1.0 - 1.0 * ey, c1 = 1.0 * ex - sx * 1.0, c2 = sx * ey - sy * ex;
            const double nrm = std::sqrt(c0 * c0 + c1 * c1);
            le_l[6 * i + 3 * s] = c0 / nrm; le_l[6 * i + 3 * s + 1] = c1 / nrm; le_l[6 * i + 3 * s + 2] = c2 / nrm;
        }
        const float* q1 = lineEq + 3 * l1;
        const float* q2 = lineEq + 3 * l2;
        if (q1[0] == 0 && q1[1] == 0 && q1[2] == 0) continue;
        if (q2[0] == 0 && q2[1] == 0 && q2[2] == 0) continue;
        const double* A1 = lines3d + 6 * l1;
        const double* A2 = lines3d + 6 * l2;
        auto zero6 = [](const double* a) { return a[0] == 0 && a[1] == 0 && a[2] == 0 && a[3] == 0 && a[4] == 0 && a[5] == 0; };
        if (zero6(A1)) continue;
        if (zero6(A2)) continue;
        float pn[3] = {q1[1] * q2[2] - q1[2] * q2[1], q1[2] * q2[0] - q1[0] * q2[2], q1[0] * q2[1] - q1[1] * q2[0]};  // cv::Vec3f::cross
        const float nr = std::sqrt(pn[0] * pn[0] + pn[1] * pn[1] + pn[2] * pn[2]);
        pn[0] = pn[0] / nr; pn[1] = pn[1] / nr; pn[2] = pn[2] / nr;
        const double nx = pn[0], ny = pn[1], nz = pn[2];
        const double* c3 = cross + 3 * i;
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
        if (dmax - dmin > 0.05) continue;
        const float planeDis = -(d1 + d2 + d3 + d4 + d5) / 5;
        float pl[4] = {(float)nx, (float)ny, (float)nz, planeDis};
        double nn[3] = {nx, ny, nz};
        if (pl[3] < 0) {
            for (int k = 0; k < 4; ++k) pl[k] = -pl[k];
            for (int k = 0; k < 3; ++k) nn[k] = -nn[k];
        }
        bool old = false;  // Frame::OldPlane
        for (int k = 0; k < np && k < cap; ++k) {
            const float* pli = planes + 4 * k;
            const float d = pl[3] - pli[3];
            const float angle = pl[0] * pli[0] + pl[1] * pli[1] + pl[2] * pli[2];
            if (d > 0.2 || d < -0.2) continue;
            if (angle < 0.9397 && angle > -0.9397) continue;
            old = true;
            break;
        }
        if (old) continue;
        if (np < cap) {
            for (int k = 0; k < 4; ++k) planes[4 * np + k] = pl[k];
            for (int k = 0; k < 3; ++k) normals[3 * np + k] = nn[k];
            lineNo[2 * np] = l1; lineNo[2 * np + 1] = l2;
            for (int k = 0; k < 3; ++k) cross3d[3 * np + k] = c3[k];
            cross2d[2 * np] = xy[2 * i]; cross2d[2 * np + 1] = xy[2 * i + 1];
        }
        ++np;
    }
    return np;
}

}  // extern "C"
