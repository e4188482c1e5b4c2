// ORACLE — test infrastructure only (see orb_oracle.cpp header). CPU restatement of PSL-SLAM's
// line front-end:
//   LINEextractor::operator()             add_src/LineExtractor.cpp:325-366
//   LSDDetector::detect (contrib wrapper) Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:166-251
//   cv::createLineSegmentDetector()       OpenCV 3.x imgproc lsd.cpp — NOT in the reference tree;
//                                         restated from the published algorithm
//   optimizeAndMergeLines_lsd & friends   add_src/uselongline.cpp:5-485
//   BinaryDescriptor::compute (LBD)       Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp:
//                                         60,76-118,219-261,351-413,540-688,1027-1373
//   CPartiallyRecoverConnectivity         add_src/PartiallyRecoverConnectivity.cpp:14-247
// PARITY UNPINNED (no fixtures upstream, OpenCV absent).  Conventions chosen where the reference
// leaves behaviour open: OpenCV-3.x LSD.  The reference calls the STOCK contrib class
// cv::line_descriptor::LSDDetector (add_src/LineExtractor.cpp:336, linked with -lopencv_line_descriptor,
// CMakeLists.txt:96), whose source is not in the tree; upstream opencv_contrib's detectImpl constructs
// createLineSegmentDetector(LSD_REFINE_ADV), so the default here is LSD_REFINE_ADV: rect_improve + the
// NFA test at log_eps = 0 (restated from the published lsd.cpp; the nfa() / log_gamma() arithmetic has a
// twin in the tree, Thirdparty/line_descriptor/src/ED_Lib/NFA.cpp:106-240).  The vendored, never-called
// LSDDetectorC (Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:185) uses the default
// LSD_REFINE_STD; pso_set_lsd_refine(1) selects that.  Which of the two the linked library used cannot
// be verified offline.  The
// seed loop walks the coordinate list by index, i.e. in raster order (the bin-sorted links it
// builds are never followed); std::sort calls whose ties reach the output are stable with index
// tie-break (H16); no FMA contraction (H6); empty input -> empty output (H12); unqualified libm
// calls on float arguments (cos, sin, tan, atan, atan2) resolve to the float overloads, as they do
// with libstdc++'s <math.h> wrapper (same rule as src/ORBextractor.cc:113 for the ORB path).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <set>
#include <vector>

#include "psl_math_oracle.h"
#include "psl_oracle.h"

#define PSL_F64_QUAL static inline
#include "../psl-slam_amd/csrc/psl_f64math.h"  // optional: the product's restated log / exp / log10 (pso_set_nfa_math(1))

namespace {

const double kPI = 3.1415926535897932384626433832795;  // CV_PI
const double DEG_TO_RADS = kPI / 180;
const double NOTDEF = -1024.0;
const double M_3_2_PI = (3 * kPI) / 2;
const double M_2__PI = 2 * kPI;

inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

int g_lsd_refine = 2;   // 1 = LSD_REFINE_STD, 2 = LSD_REFINE_ADV (default, see the header)
int g_nfa_math = 0;     // 0 = this host's libm (what the reference calls), 1 = psl_f64math.h (what the device evaluates)

// ================================ LSD (OpenCV 3.x) ================================
struct RegionPoint { int x, y; double angle, modgrad; };
struct Rect { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; };

struct Lsd {
    int W = 0, H = 0;  // scaled image
    std::vector<double> scaled, angles, modgrad;
    std::vector<uint8_t> used;

    // GaussianBlur(64F, ksize 7, sigma 0.75, REFLECT_101) then resize(INTER_LINEAR, 0.8) on doubles
    void scale_image(const uint8_t* gray, int w, int h, int stride) {
        const double SCALE = 0.8, SIGMA_SCALE = 0.6;
        const double sigma = SIGMA_SCALE / SCALE, sprec = 3;
        const unsigned hk = (unsigned)std::ceil(sigma * std::sqrt(2 * sprec * std::log(10.0)));
        const int ksize = 1 + 2 * hk, r = ksize / 2;
        std::vector<double> k(ksize);
        {
            double scale2X = -0.5 / (sigma * sigma), sum = 0;
            for (int i = 0; i < ksize; ++i) { double x = i - (ksize - 1) * 0.5; k[i] = std::exp(scale2X * x * x); sum += k[i]; }
            sum = 1. / sum;
            for (int i = 0; i < ksize; ++i) k[i] *= sum;
        }
        std::vector<double> rows((size_t)w * h), blur((size_t)w * h);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {  // RowFilter: s = k0*S0; s += k1*S1; ...
                double s = k[0] * (double)gray[(size_t)y * stride + reflect101(x - r, w)];
                for (int j = 1; j < ksize; ++j) s += k[j] * (double)gray[(size_t)y * stride + reflect101(x - r + j, w)];
                rows[(size_t)y * w + x] = s;
            }
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {  // SymmColumnFilter: centre, then (S[k] + S[-k]) * ky[k]
                double s = k[r] * rows[(size_t)y * w + x];
                for (int j = 1; j <= r; ++j)
                    s += k[r + j] * (rows[(size_t)reflect101(y + j, h) * w + x] + rows[(size_t)reflect101(y - j, h) * w + x]);
                blur[(size_t)y * w + x] = s;
            }
        W = pso_cvround(w * SCALE);
        H = pso_cvround(h * SCALE);
        auto table = [](int ssize, int dsize, double inv_scale, bool clampf, std::vector<int>& ofs, std::vector<float>& co) {
            const double scale = 1. / inv_scale;
            ofs.resize(dsize); co.resize(2 * (size_t)dsize);
            for (int d = 0; d < dsize; ++d) {
                float f = (float)((d + 0.5) * scale - 0.5);
                int s = pso_cvfloor(f);
                f -= s;
                if (clampf) { if (s < 0) { f = 0; s = 0; } if (s >= ssize - 1) { f = 0; s = ssize - 1; } }
                ofs[d] = s; co[2 * d] = 1.f - f; co[2 * d + 1] = f;
            }
        };
        std::vector<int> xo, yo; std::vector<float> al, be;
        table(w, W, SCALE, true, xo, al);
        table(h, H, SCALE, false, yo, be);
        scaled.assign((size_t)W * H, 0);
        for (int dy = 0; dy < H; ++dy) {
            int sy0 = yo[dy], sy1 = sy0 + 1;
            sy0 = sy0 < 0 ? 0 : (sy0 >= h ? h - 1 : sy0);
            sy1 = sy1 < 0 ? 0 : (sy1 >= h ? h - 1 : sy1);
            for (int dx = 0; dx < W; ++dx) {
                const int sx = xo[dx];
                const double a0 = al[2 * dx], a1 = al[2 * dx + 1];
                double h0, h1;
                if (sx + 1 < w) {
                    h0 = blur[(size_t)sy0 * w + sx] * a0 + blur[(size_t)sy0 * w + sx + 1] * a1;
                    h1 = blur[(size_t)sy1 * w + sx] * a0 + blur[(size_t)sy1 * w + sx + 1] * a1;
                } else { h0 = blur[(size_t)sy0 * w + sx]; h1 = blur[(size_t)sy1 * w + sx]; }
                scaled[(size_t)dy * W + dx] = h0 * (double)be[2 * dy] + h1 * (double)be[2 * dy + 1];
            }
        }
    }

    void ll_angle(double threshold) {
        angles.assign((size_t)W * H, NOTDEF);
        modgrad.assign((size_t)W * H, 0.0);
        for (int y = 0; y < H - 1; ++y)
            for (int x = 0; x < W - 1; ++x) {
                const double* r0 = &scaled[(size_t)y * W];
                const double* r1 = &scaled[(size_t)(y + 1) * W];
                double DA = r1[x + 1] - r0[x];
                double BC = r0[x + 1] - r1[x];
                double gx = DA + BC, gy = DA - BC;
                double norm = std::sqrt((gx * gx + gy * gy) / 4);
                modgrad[(size_t)y * W + x] = norm;
                if (norm <= threshold) angles[(size_t)y * W + x] = NOTDEF;
                else angles[(size_t)y * W + x] = pso_fast_atan2(float(gx), float(-gy)) * DEG_TO_RADS;
            }
    }

    bool is_aligned(int addr, double theta, double prec) const {
        const double a = angles[addr];
        if (a == NOTDEF) return false;
        double n_theta = theta - a;
        if (n_theta < 0) n_theta = -n_theta;
        if (n_theta > M_3_2_PI) { n_theta -= M_2__PI; if (n_theta < 0) n_theta = -n_theta; }
        return n_theta <= prec;
    }

    long debug_cands = 0;  // with debug_grow: neighbours with a defined angle, not used, that were tested
    std::vector<int32_t>* debug_grow = nullptr;  // tap: per region_grow call [-1, n] then n x (x, y, index of the entry that added it)
    void region_grow(int sx, int sy, std::vector<RegionPoint>& reg, int& reg_size, double& reg_angle, double prec) {
        const size_t log0 = debug_grow ? debug_grow->size() : 0;
        if (debug_grow) { debug_grow->push_back(-1); debug_grow->push_back(0); debug_grow->push_back(sx); debug_grow->push_back(sy); debug_grow->push_back(-1); }
        reg_size = 1;
        int addr = sx + sy * W;
        reg[0] = {sx, sy, angles[addr], modgrad[addr]};
        reg_angle = angles[addr];
        float sumdx = float(std::cos(reg_angle)), sumdy = float(std::sin(reg_angle));
        used[addr] = 1;
        for (int i = 0; i < reg_size; ++i) {
            const RegionPoint rp = reg[i];
            int xx_min = std::max(rp.x - 1, 0), xx_max = std::min(rp.x + 1, W - 1);
            int yy_min = std::max(rp.y - 1, 0), yy_max = std::min(rp.y + 1, H - 1);
            for (int yy = yy_min; yy <= yy_max; ++yy) {
                int c = xx_min + yy * W;
                for (int xx = xx_min; xx <= xx_max; ++xx, ++c) {
                    if (debug_grow && used[c] != 1 && angles[c] != NOTDEF) ++debug_cands;
                    if (used[c] != 1 && is_aligned(c, reg_angle, prec)) {
                        used[c] = 1;
                        const double angle = angles[c];
                        reg[reg_size++] = {xx, yy, angle, modgrad[c]};
                        if (debug_grow) { debug_grow->push_back(xx); debug_grow->push_back(yy); debug_grow->push_back(i); }
                        sumdx += pso_cosf(float(angle));
                        sumdy += pso_sinf(float(angle));
                        reg_angle = pso_fast_atan2(sumdy, sumdx) * DEG_TO_RADS;
                    }
                }
            }
        }
        if (debug_grow) (*debug_grow)[log0 + 1] = reg_size;
    }

    static double angle_diff_signed(double a, double b) {
        double diff = a - b;
        while (diff <= -kPI) diff += M_2__PI;
        while (diff > kPI) diff -= M_2__PI;
        return diff;
    }
    static double dist_sq(double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); }
    static double dist(double x1, double y1, double x2, double y2) { return std::sqrt(dist_sq(x1, y1, x2, y2)); }

    double get_theta(const std::vector<RegionPoint>& reg, int reg_size, double x, double y, double reg_angle, double prec) const {
        double Ixx = 0, Iyy = 0, Ixy = 0;
        for (int i = 0; i < reg_size; ++i) {
            const double dx = double(reg[i].x) - x, dy = double(reg[i].y) - y, w = reg[i].modgrad;
            Ixx += dy * dy * w;
            Iyy += dx * dx * w;
            Ixy -= dx * dy * w;
        }
        const double lambda = 0.5 * (Ixx + Iyy - std::sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
        double theta = (std::fabs(Ixx) > std::fabs(Iyy)) ? double(pso_fast_atan2(float(lambda - Ixx), float(Ixy)))
                                                        : double(pso_fast_atan2(float(Ixy), float(lambda - Iyy)));
        theta *= DEG_TO_RADS;
        if (std::fabs(angle_diff_signed(theta, reg_angle)) > prec) theta += kPI;
        return theta;
    }

    void region2rect(const std::vector<RegionPoint>& reg, int reg_size, double reg_angle, double prec, double p, Rect& rec) const {
        double x = 0, y = 0, sum = 0;
        for (int i = 0; i < reg_size; ++i) {
            const double w = reg[i].modgrad;
            x += double(reg[i].x) * w;
            y += double(reg[i].y) * w;
            sum += w;
        }
        x /= sum; y /= sum;
        const double theta = get_theta(reg, reg_size, x, y, reg_angle, prec);
        const double dx = std::cos(theta), dy = std::sin(theta);
        double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
        for (int i = 0; i < reg_size; ++i) {
            const double rdx = double(reg[i].x) - x, rdy = double(reg[i].y) - y;
            const double l = rdx * dx + rdy * dy, w = -rdx * dy + rdy * dx;
            if (l > l_max) l_max = l; else if (l < l_min) l_min = l;
            if (w > w_max) w_max = w; else if (w < w_min) w_min = w;
        }
        rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy;
        rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
        rec.width = w_max - w_min;
        rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy; rec.prec = prec; rec.p = p;
        if (rec.width < 1.0) rec.width = 1.0;
    }

    long long debug_refine[10] = {0};  // tap (tools/grow_stats.py): regions of min_reg_size or more | refinements entered | their pixels | regrown pixels |
                                      // reduce_region_radius calls | its radius steps | pixels it visits | pixels of its region2rect calls | max over the steps of (2 n - m): list length + pixels removed | W * H
    bool reduce_region_radius(std::vector<RegionPoint>& reg, int& reg_size, double reg_angle, double prec, double p, Rect& rec,
                              double density, double density_th) {
        ++debug_refine[4];
        const double xc = double(reg[0].x), yc = double(reg[0].y);
        const double r1 = dist_sq(xc, yc, rec.x1, rec.y1), r2 = dist_sq(xc, yc, rec.x2, rec.y2);
        double radSq = r1 > r2 ? r1 : r2;
        while (density < density_th) {
            radSq *= 0.75 * 0.75;
            ++debug_refine[5];
            debug_refine[6] += reg_size;
            const int n0 = reg_size;
            for (int i = 0; i < reg_size; ++i)
                if (dist_sq(xc, yc, double(reg[i].x), double(reg[i].y)) > radSq) {
                    used[reg[i].x + reg[i].y * W] = 0;
                    std::swap(reg[i], reg[reg_size - 1]);
                    --reg_size;
                    --i;
                }
            if (n0 + (n0 - reg_size) > debug_refine[8]) debug_refine[8] = n0 + (n0 - reg_size);
            debug_refine[9] = (long long)W * H;
            if (reg_size < 2) return false;
            debug_refine[7] += reg_size;
            region2rect(reg, reg_size, reg_angle, prec, p, rec);
            density = double(reg_size) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        }
        return true;
    }

    bool refine(std::vector<RegionPoint>& reg, int& reg_size, double reg_angle, double prec, double p, Rect& rec, double density_th) {
        double density = double(reg_size) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        ++debug_refine[0];
        if (density >= density_th) return true;
        ++debug_refine[1];
        debug_refine[2] += reg_size;
        const double xc = double(reg[0].x), yc = double(reg[0].y), ang_c = reg[0].angle;
        double sum = 0, s_sum = 0;
        int n = 0;
        for (int i = 0; i < reg_size; ++i) {
            used[reg[i].x + reg[i].y * W] = 0;
            if (dist(xc, yc, reg[i].x, reg[i].y) < rec.width) {
                const double ang_d = angle_diff_signed(reg[i].angle, ang_c);
                sum += ang_d;
                s_sum += ang_d * ang_d;
                ++n;
            }
        }
        const double mean_angle = sum / double(n);
        const double tau = 2.0 * std::sqrt((s_sum - 2.0 * mean_angle * sum) / double(n) + mean_angle * mean_angle);
        region_grow(reg[0].x, reg[0].y, reg, reg_size, reg_angle, tau);
        debug_refine[3] += reg_size;
        if (reg_size < 2) return false;
        region2rect(reg, reg_size, reg_angle, prec, p, rec);
        density = double(reg_size) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        if (density < density_th) return reduce_region_radius(reg, reg_size, reg_angle, prec, p, rec, density, density_th);
        return true;
    }

    // ---- LSD_REFINE_ADV: rect_nfa / nfa / rect_improve (OpenCV 3.x lsd.cpp; nfa() and log_gamma() as in
    //      Thirdparty/line_descriptor/src/ED_Lib/NFA.cpp:106-240) ----
    double LOG_NT = 0;
    std::vector<double>* debug_rects = nullptr;  // tap: the rectangles handed to rect_improve (12 doubles each, Rect layout)
    static double m_log(double x) { return g_nfa_math ? psl_log(x) : std::log(x); }
    static double m_exp(double x) { return g_nfa_math ? psl_exp(x) : std::exp(x); }
    static double m_log10(double x) { return g_nfa_math ? psl_log10(x) : std::log10(x); }
    // pow(x, n) for the integer-valued arguments log_gamma sees: exact products where libm's pow is exact too
    // (x <= 15, n <= 6), x^6 = (x^3)^2 with x^3 exact (one rounding, as a < 1 ulp pow returns) for the Windschitl term
    static double m_pow(double x, double y) {
        if (!g_nfa_math) return std::pow(x, y);
        if (y == (double)(int)y && y >= 0 && y <= 6 && x == std::floor(x)) {
            if (y == 6) { const double c = x * x * x; return c * c; }
            double r = 1; for (int i = 0; i < (int)y; ++i) r *= x; return r;
        }
        if (y == 1.0) return x;   // exact in any libm
        return psl_pow_pos(x, y);
    }
    static double m_sinh(double x) { return g_nfa_math ? psl_sinh_small(x) : std::sinh(x); }
    static double log_gamma_windschitl(double x) {
        return 0.918938533204673 + (x - 0.5) * m_log(x) - x + 0.5 * x * m_log(x * m_sinh(1 / x) + 1 / (810.0 * m_pow(x, 6.0)));
    }
    static double log_gamma_lanczos(double x) {
        static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
        double a = (x + 0.5) * m_log(x + 5.5) - (x + 5.5);
        double b = 0;
        for (int n = 0; n < 7; ++n) {
            a -= m_log(x + double(n));
            b += q[n] * m_pow(x, double(n));
        }
        return a + m_log(b);
    }
    static double log_gamma(double x) { return x > 15.0 ? log_gamma_windschitl(x) : log_gamma_lanczos(x); }
    static bool double_equal(double a, double b) {
        if (a == b) return true;
        const double abs_diff = std::fabs(a - b), aa = std::fabs(a), bb = std::fabs(b);
        double abs_max = aa > bb ? aa : bb;
        if (abs_max < 2.2250738585072014e-308) abs_max = 2.2250738585072014e-308;  // DBL_MIN
        return (abs_diff / abs_max) <= (100.0 * 2.2204460492503131e-16);            // RELATIVE_ERROR_FACTOR * DBL_EPSILON
    }
    double nfa(int n, int k, double p) const {
        if (n == 0 || k == 0) return -LOG_NT;
        if (n == k) return -LOG_NT - double(n) * m_log10(p);
        const double p_term = p / (1 - p);
        const double log1term = log_gamma(double(n) + 1) - log_gamma(double(k) + 1) - log_gamma(double(n - k) + 1)
                                + double(k) * m_log(p) + double(n - k) * m_log(1.0 - p);
        double term = m_exp(log1term);
        if (double_equal(term, 0)) {
            if (k > n * p) return -log1term / 2.30258509299404568402 - LOG_NT;  // M_LN10
            return -LOG_NT;
        }
        double bin_tail = term;
        const double tolerance = 0.1;
        for (int i = k + 1; i <= n; ++i) {
            const double bin_term = double(n - i + 1) / double(i);
            const double mult_term = bin_term * p_term;
            term *= mult_term;
            bin_tail += term;
            if (bin_term < 1) {
                const double err = term * ((1 - m_pow(mult_term, double(n - i + 1))) / (1 - mult_term) - 1);
                if (err < tolerance * std::fabs(-m_log10(bin_tail) - LOG_NT) * bin_tail) break;
            }
        }
        return -m_log10(bin_tail) - LOG_NT;
    }
    // The rectangle scan of OpenCV 3.x as it behaves: corners truncated to int, slopes by INTEGER division, the second
    // slopes take `tailp->p.x` where the y coordinate was meant, and a row outside the image is skipped WITHOUT
    // advancing the column bounds.
    struct Edge { int x, y; bool taken; };
    void rect_counts(const Rect& rec, int* total, int* aligned) const {
        int total_pts = 0, alg_pts = 0;
        const double half_width = rec.width / 2.0;
        const double dyhw = rec.dy * half_width, dxhw = rec.dx * half_width;
        Edge e[4] = {{int(rec.x1 - dyhw), int(rec.y1 + dxhw), false}, {int(rec.x2 - dyhw), int(rec.y2 + dxhw), false},
                     {int(rec.x2 + dyhw), int(rec.y2 - dxhw), false}, {int(rec.x1 + dyhw), int(rec.y1 - dxhw), false}};
        std::sort(e, e + 4, [](const Edge& a, const Edge& b) { return a.x == b.x ? a.y < b.y : a.x < b.x; });
        Edge *min_y = &e[0], *max_y = &e[0];
        for (int i = 1; i < 4; ++i) {
            if (min_y->y > e[i].y) min_y = &e[i];
            if (max_y->y < e[i].y) max_y = &e[i];
        }
        min_y->taken = true;
        Edge* leftmost = nullptr;
        for (int i = 0; i < 4; ++i) if (!e[i].taken) { if (!leftmost) leftmost = &e[i]; else if (leftmost->x > e[i].x) leftmost = &e[i]; }
        leftmost->taken = true;
        Edge* rightmost = nullptr;
        for (int i = 0; i < 4; ++i) if (!e[i].taken) { if (!rightmost) rightmost = &e[i]; else if (rightmost->x < e[i].x) rightmost = &e[i]; }
        rightmost->taken = true;
        Edge* tailp = nullptr;
        for (int i = 0; i < 4; ++i) if (!e[i].taken) { if (!tailp) tailp = &e[i]; else if (tailp->x > e[i].x) tailp = &e[i]; }
        tailp->taken = true;
        const double flstep = (min_y->y != leftmost->y) ? (min_y->x - leftmost->x) / (min_y->y - leftmost->y) : 0;
        const double slstep = (leftmost->y != tailp->x) ? (leftmost->x - tailp->x) / (leftmost->y - tailp->x) : 0;
        const double frstep = (min_y->y != rightmost->y) ? (min_y->x - rightmost->x) / (min_y->y - rightmost->y) : 0;
        const double srstep = (rightmost->y != tailp->x) ? (rightmost->x - tailp->x) / (rightmost->y - tailp->x) : 0;
        double lstep = flstep, rstep = frstep;
        double left_x = min_y->x, right_x = min_y->x;
        const int min_iter = min_y->y, max_iter = max_y->y;
        for (int y = min_iter; y <= max_iter; ++y) {
            if (y < 0 || y >= H) continue;
            for (int x = int(left_x); x <= int(right_x); ++x) {
                if (x < 0 || x >= W) continue;
                ++total_pts;
                if (is_aligned(x + y * W, rec.theta, rec.prec)) ++alg_pts;
            }
            if (y >= leftmost->y) lstep = slstep;
            if (y >= rightmost->y) rstep = srstep;
            left_x += lstep;
            right_x += rstep;
        }
        *total = total_pts; *aligned = alg_pts;
    }
    double rect_nfa(const Rect& rec) const {
        int n, k;
        rect_counts(rec, &n, &k);
        return nfa(n, k, rec.p);
    }
    double rect_improve(Rect& rec) const {
        const double delta = 0.5, delta_2 = delta / 2.0, LOG_EPS = 0;
        double log_nfa = rect_nfa(rec);
        if (log_nfa > LOG_EPS) return log_nfa;
        Rect r = rec;
        for (int n = 0; n < 5; ++n) {  // finer precision
            r.p /= 2;
            r.prec = r.p * kPI;
            const double v = rect_nfa(r);
            if (v > log_nfa) { log_nfa = v; rec = r; }
        }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (int n = 0; n < 5; ++n) {  // reduce width
            if ((r.width - delta) >= 0.5) {
                r.width -= delta;
                const double v = rect_nfa(r);
                if (v > log_nfa) { rec = r; log_nfa = v; }
            }
        }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (int n = 0; n < 5; ++n) {  // reduce one side
            if ((r.width - delta) >= 0.5) {
                r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2;
                r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2;
                r.width -= delta;
                const double v = rect_nfa(r);
                if (v > log_nfa) { rec = r; log_nfa = v; }
            }
        }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (int n = 0; n < 5; ++n) {  // reduce the other side
            if ((r.width - delta) >= 0.5) {
                r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2;
                r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2;
                r.width -= delta;
                const double v = rect_nfa(r);
                if (v > log_nfa) { rec = r; log_nfa = v; }
            }
        }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (int n = 0; n < 5; ++n) {  // finer precision again
            if ((r.width - delta) >= 0.5) {
                r.p /= 2;
                r.prec = r.p * kPI;
                const double v = rect_nfa(r);
                if (v > log_nfa) { rec = r; log_nfa = v; }
            }
        }
        return log_nfa;
    }

    void detect(const uint8_t* gray, int w, int h, int stride, std::vector<float>& lines) {
        const double SCALE = 0.8, QUANT = 2.0, ANG_TH = 22.5, DENSITY_TH = 0.7;
        const double prec = kPI * ANG_TH / 180, p = ANG_TH / 180, rho = QUANT / std::sin(prec);
        scale_image(gray, w, h, stride);
        ll_angle(rho);
        LOG_NT = 5 * (std::log10(double(W)) + std::log10(double(H))) / 2 + std::log10(11.0);
        const size_t min_reg_size = size_t(-LOG_NT / std::log10(p));
        used.assign((size_t)W * H, 0);
        std::vector<RegionPoint> reg((size_t)W * H);
        lines.clear();
        for (int y = 0; y < H - 1; ++y)        // list[i] by index == raster order over (W-1) x (H-1)
            for (int x = 0; x < W - 1; ++x) {
                const int adx = x + y * W;
                if (used[adx] != 0 || angles[adx] == NOTDEF) continue;
                int reg_size;
                double reg_angle;
                region_grow(x, y, reg, reg_size, reg_angle, prec);
                if ((size_t)reg_size < min_reg_size) continue;
                Rect rec;
                region2rect(reg, reg_size, reg_angle, prec, p, rec);
                if (!refine(reg, reg_size, reg_angle, prec, p, rec, DENSITY_TH)) continue;
                if (g_lsd_refine >= 2) {  // LSD_REFINE_ADV
                    if (debug_rects) { const double* f = &rec.x1; debug_rects->insert(debug_rects->end(), f, f + 12); }
                    const double log_nfa = rect_improve(rec);
                    if (log_nfa <= 0) continue;  // LOG_EPS = 0
                }
                rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
                rec.x1 /= SCALE; rec.y1 /= SCALE; rec.x2 /= SCALE; rec.y2 /= SCALE;
                lines.push_back(float(rec.x1)); lines.push_back(float(rec.y1));
                lines.push_back(float(rec.x2)); lines.push_back(float(rec.y2));
            }
    }
};

// contrib wrapper: checkLineExtremes (LSDDetector_custom.cpp:111-138); only the clamped extremes
// survive optimizeAndMergeLines_lsd, which rebuilds every KeyLine field.
void clamp_extremes(float* e, int w, int h) {
    if (e[0] < 0) e[0] = 0;
    if (e[0] >= w) e[0] = (float)w - 1.0f;
    if (e[2] < 0) e[2] = 0;
    if (e[2] >= w) e[2] = (float)w - 1.0f;
    if (e[1] < 0) e[1] = 0;
    if (e[1] >= h) e[1] = (float)h - 1.0f;
    if (e[3] < 0) e[3] = 0;
    if (e[3] >= h) e[3] = (float)h - 1.0f;
}

// ================================ MergeLines (uselongline.cpp) ================================
typedef std::vector<float> V4;  // flat x1,y1,x2,y2 per line

float point_line_distance(const float* line, float x0, float y0) {
    float x1 = line[0], y1 = line[1], x2 = line[2], y2 = line[3];
    // std::pow on floats promotes to double; the quotient is rounded to float on return
    double num = std::fabs((y2 - y1) * x0 + (x1 - x2) * y0 + ((x2 * y1) - (x1 * y2)));
    double den = std::sqrt(std::pow(y2 - y1, 2) + std::pow(x1 - x2, 2));
    return (float)(num / den);
}

float angle_diff_f(float a1, float a2) {
    float c1 = std::abs(a2 - a1);
    float c2 = (float)(M_PI + std::min(a1, a2) - std::max(a1, a2));
    return std::min(c1, c2);
}

void merge_two_lines(const float* l1, const float* l2, float* out) {  // :266-334
    float ax = l1[0], ay = l1[1], bx = l1[2], by = l1[3];
    float cx = l2[0], cy = l2[1], dx = l2[2], dy = l2[3];
    float dlix = bx - ax, dliy = by - ay, dljx = dx - cx, dljy = dy - cy;
    double li = std::sqrt((double)(dlix * dlix) + (double)(dliy * dliy));
    double lj = std::sqrt((double)(dljx * dljx) + (double)(dljy * dljy));
    double xg = (li * (double)(ax + bx) + lj * (double)(cx + dx)) / (double)(2.0 * (li + lj));
    double yg = (li * (double)(ay + by) + lj * (double)(cy + dy)) / (double)(2.0 * (li + lj));
    double thi = dlix == 0.0f ? kPI / 2.0 : std::atan(dliy / dlix);
    double thj = dljx == 0.0f ? kPI / 2.0 : std::atan(dljy / dljx);
    double thr;
    if (std::fabs(thi - thj) <= kPI / 2.0) thr = (li * thi + lj * thj) / (li + lj);
    else {
        double tmp = thj - kPI * (thj / std::fabs(thj));
        thr = li * thi + lj * tmp;
        thr /= (li + lj);
    }
    const double s = std::sin(thr), c = std::cos(thr);
    double axg = ((double)ay - yg) * s + ((double)ax - xg) * c;
    double bxg = ((double)by - yg) * s + ((double)bx - xg) * c;
    double cxg = ((double)cy - yg) * s + ((double)cx - xg) * c;
    double dxg = ((double)dy - yg) * s + ((double)dx - xg) * c;
    double d1 = std::min(axg, std::min(bxg, std::min(cxg, dxg)));
    double d2 = std::max(axg, std::max(bxg, std::max(cxg, dxg)));
    out[0] = (float)(d1 * c + xg); out[1] = (float)(d1 * s + yg);
    out[2] = (float)(d2 * c + xg); out[3] = (float)(d2 * s + yg);
}

void merge_lines(const V4& src, V4& dst, float angle_threshold, float distance_threshold, float endpoint_threshold) {
    dst.clear();
    const size_t n = src.size() / 4;
    if (n == 0) return;  // reference reads source_lines[0] of an empty vector (H12)
    std::vector<float> angles(n), length(n);
    for (size_t i = 0; i < n; ++i) {
        float dx = src[4 * i + 2] - src[4 * i], dy = src[4 * i + 3] - src[4 * i + 1];
        angles[i] = std::atan(dy / dx);               // Eigen ArrayXf atan() == atanf per element
        length[i] = std::sqrt(dx * dx + dy * dy);
    }
    std::vector<size_t> indices(n);
    for (size_t a = 0; a < n; ++a) indices[a] = a;
    std::stable_sort(indices.begin(), indices.end(), [&](size_t i1, size_t i2) { return angles[i1] < angles[i2]; });
    const float angle_thr = angle_threshold, distance_thr = distance_threshold;
    const float ep_thr = endpoint_threshold * endpoint_threshold;
    const float quater_PI = (float)(M_PI / 4.0);
    std::vector<std::vector<size_t>> neighbors(n);
    for (size_t i = 0; i < n; i++) {
        size_t idx1 = indices[i];
        float x11 = src[4 * idx1], y11 = src[4 * idx1 + 1], x12 = src[4 * idx1 + 2], y12 = src[4 * idx1 + 3];
        float angle1 = angles[idx1];
        bool to_sort_x = (std::abs(angle1) < quater_PI);
        if ((to_sort_x && (x12 < x11)) || ((!to_sort_x) && y12 < y11)) { std::swap(x11, x12); std::swap(y11, y12); }
        for (size_t j = i + 1; j < n; j++) {
            size_t idx2 = indices[j];
            float x21 = src[4 * idx2], y21 = src[4 * idx2 + 1], x22 = src[4 * idx2 + 2], y22 = src[4 * idx2 + 3];
            if ((to_sort_x && (x22 < x21)) || ((!to_sort_x) && y22 < y21)) { std::swap(x21, x22); std::swap(y21, y22); }
            float angle2 = angles[idx2];
            float d_angle = angle_diff_f(angle1, angle2);
            if (d_angle > angle_thr) {
                if (std::abs(angle1) < (M_PI_2 - angle_threshold)) break; else continue;
            }
            float mid_x1 = (float)(0.5 * (src[4 * idx1] + src[4 * idx1 + 2])), mid_y1 = (float)(0.5 * (src[4 * idx1 + 1] + src[4 * idx1 + 3]));
            float mid_x2 = (float)(0.5 * (src[4 * idx2] + src[4 * idx2 + 2])), mid_y2 = (float)(0.5 * (src[4 * idx2 + 1] + src[4 * idx2 + 3]));
            float mid1_to_line2 = point_line_distance(&src[4 * idx2], mid_x1, mid_y1);
            float mid2_to_line1 = point_line_distance(&src[4 * idx1], mid_x2, mid_y2);
            if (mid1_to_line2 > distance_thr && mid2_to_line1 > distance_thr) continue;
            float cx12, cy12, cx21, cy21;
            if ((to_sort_x && x12 > x22) || (!to_sort_x && y12 > y22)) { cx12 = x22; cy12 = y22; cx21 = x11; cy21 = y11; }
            else { cx12 = x12; cy12 = y12; cx21 = x21; cy21 = y21; }
            bool to_merge = ((to_sort_x && cx12 >= cx21) || (!to_sort_x && cy12 >= cy21));
            if (!to_merge) {
                float d_ep = (cx21 - cx12) * (cx21 - cx12) + (cy21 - cy12) * (cy21 - cy12);
                to_merge = (d_ep < ep_thr);
            }
            if (to_merge) { neighbors[idx1].push_back(idx2); neighbors[idx2].push_back(idx1); }
        }
    }
    std::vector<int> cluster_codes(n, -1);
    std::vector<std::vector<size_t>> cluster_ids;
    for (size_t i = 0; i < n; i++) {
        if (cluster_codes[i] >= 0) continue;
        size_t new_code = cluster_ids.size();
        cluster_codes[i] = (int)new_code;
        std::vector<size_t> to_check = neighbors[i], cluster;
        cluster.push_back(i);
        while (!to_check.empty()) {
            std::set<size_t> tmp;
            for (size_t j : to_check) {
                if (cluster_codes[j] < 0) { cluster_codes[j] = (int)new_code; cluster.push_back(j); }
                for (size_t k : neighbors[j]) if (cluster_codes[k] < 0) tmp.insert(k);
            }
            to_check.assign(tmp.begin(), tmp.end());
        }
        cluster_ids.push_back(cluster);
    }
    std::vector<std::vector<size_t>> new_cluster_ids;
    for (auto& cluster : cluster_ids) {
        size_t cs = cluster.size();
        if (cs <= 2) { new_cluster_ids.push_back(cluster); continue; }
        std::stable_sort(cluster.begin(), cluster.end(), [&](size_t i1, size_t i2) { return length[i1] > length[i2]; });
        std::vector<size_t> loc(n, 0);
        for (size_t i = 0; i < cs; i++) loc[cluster[i]] = i;
        std::vector<bool> clustered(cs, false);
        for (size_t j = 0; j < cs; j++) {
            if (clustered[j]) continue;
            size_t line_idx = cluster[j];
            std::vector<size_t> sub;
            sub.push_back(line_idx);
            for (size_t k : neighbors[line_idx]) { clustered[loc[k]] = true; sub.push_back(k); }
            new_cluster_ids.push_back(sub);
        }
    }
    for (auto& cluster : new_cluster_ids) {
        float nl[4] = {src[4 * cluster[0]], src[4 * cluster[0] + 1], src[4 * cluster[0] + 2], src[4 * cluster[0] + 3]};
        for (size_t i = 0; i < cluster.size(); i++) {  // the first line is merged with itself first (:247-254)
            float out[4];
            merge_two_lines(nl, &src[4 * cluster[i]], out);
            memcpy(nl, out, sizeof(nl));
        }
        dst.insert(dst.end(), nl, nl + 4);
    }
}

void filter_short(V4& lines, float length_thr) {  // :338-351
    const float thr_square = length_thr * length_thr;
    size_t keep = 0;
    for (size_t i = 0; i < lines.size() / 4; ++i) {
        float dx = lines[4 * i + 2] - lines[4 * i], dy = lines[4 * i + 3] - lines[4 * i + 1];
        float ls = dx * dx + dy * dy;
        if (ls > thr_square) { memmove(&lines[4 * keep], &lines[4 * i], 4 * sizeof(float)); ++keep; }
    }
    lines.resize(4 * keep);
}

// cv::clipLine (integer arithmetic) + cv::LineIterator(...).count, 8-connected (Appendix A.8)
bool clip_line(int w, int h, long long& x1, long long& y1, long long& x2, long long& y2) {
    const long long right = w - 1, bottom = h - 1;
    if (w <= 0 || h <= 0) return false;
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
    return (c1 | c2) == 0;
}

int line_iterator_count(int w, int h, float fx1, float fy1, float fx2, float fy2) {
    long long x1 = pso_cvround(fx1), y1 = pso_cvround(fy1), x2 = pso_cvround(fx2), y2 = pso_cvround(fy2);
    if ((unsigned long long)x1 >= (unsigned long long)w || (unsigned long long)x2 >= (unsigned long long)w ||
        (unsigned long long)y1 >= (unsigned long long)h || (unsigned long long)y2 >= (unsigned long long)h)
        if (!clip_line(w, h, x1, y1, x2, y2)) return 0;
    long long dx = x2 - x1, dy = y2 - y1;
    dx = dx < 0 ? -dx : dx; dy = dy < 0 ? -dy : dy;
    return (int)std::max(dx, dy) + 1;
}

void vec4f_to_keyline(const V4& lines, int w, int h, std::vector<PsoKeyLine>& out) {  // :411-447
    out.clear();
    for (size_t i = 0; i < lines.size() / 4; i++) {
        const float* l = &lines[4 * i];
        PsoKeyLine kl;
        const double octaveScale = 1.f;
        kl.startPointX = (float)(l[0] * octaveScale); kl.startPointY = (float)(l[1] * octaveScale);
        kl.endPointX = (float)(l[2] * octaveScale); kl.endPointY = (float)(l[3] * octaveScale);
        kl.sPointInOctaveX = l[0]; kl.sPointInOctaveY = l[1]; kl.ePointInOctaveX = l[2]; kl.ePointInOctaveY = l[3];
        kl.lineLength = (float)std::sqrt(std::pow(l[0] - l[2], 2) + std::pow(l[1] - l[3], 2));
        kl.angle = atan2f((kl.endPointY - kl.startPointY), (kl.endPointX - kl.startPointX));  // float overload (convention, header)
        kl.class_id = (int)i;
        kl.octave = 0;
        kl.size = (kl.endPointX - kl.startPointX) * (kl.endPointY - kl.startPointY);
        kl.pt_x = (kl.endPointX + kl.startPointX) / 2; kl.pt_y = (kl.endPointY + kl.startPointY) / 2;
        kl.response = kl.lineLength / std::max(w, h);
        kl.numOfPixels = line_iterator_count(w, h, l[0], l[1], l[2], l[3]);
        out.push_back(kl);
    }
}

// ================================ LBD ================================
const int kCombos[32][2] = {{0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {0, 6}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6},
                            {2, 3}, {2, 4}, {2, 5}, {2, 6}, {2, 7}, {2, 8}, {3, 4}, {3, 5}, {3, 6}, {3, 7}, {3, 8},
                            {4, 5}, {4, 6}, {4, 7}, {4, 8}, {5, 6}, {5, 7}, {5, 8}, {6, 7}, {6, 8}, {7, 8}};

struct Lbd {
    int w = 0, h = 0;
    std::vector<short> dx, dy;
    double gaussL[21], gaussG[63];

    Lbd() {  // BinaryDescriptor ctor :219-261 (integer divisions as written there)
        const int wb = 7, nb = 9;
        double u = (wb * 3 - 1) / 2;
        double sigma = (wb * 2 + 1) / 2;
        double inv = -1 / (2 * sigma * sigma);
        for (int i = 0; i < wb * 3; i++) { double d = i - u; gaussL[i] = std::exp(d * d * inv); }
        u = (nb * wb - 1) / 2;
        sigma = u;
        inv = -1 / (2 * sigma * sigma);
        for (int i = 0; i < nb * wb; i++) { double d = i - u; gaussG[i] = std::exp(d * d * inv); }
    }

    // GaussianBlur(5x5, sigma 1) on 8U (OpenCV 3.2 integer path) then Sobel 3x3 -> CV_16S, REFLECT_101
    void prepare(const uint8_t* gray, int ww, int hh, int stride) {
        w = ww; h = hh;
        int K[5];
        {
            float cf[5]; double s2 = -0.5, sum = 0;
            for (int i = 0; i < 5; ++i) { double x = i - 2.0; cf[i] = (float)std::exp(s2 * x * x); sum += cf[i]; }
            sum = 1. / sum;
            for (int i = 0; i < 5; ++i) { cf[i] = (float)(cf[i] * sum); K[i] = pso_cvround((double)cf[i] * 256.0); }
        }
        std::vector<int> tmp((size_t)w * h);
        std::vector<uint8_t> bl((size_t)w * h);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                int s = 0;
                for (int k = -2; k <= 2; ++k) s += K[k + 2] * gray[(size_t)y * stride + reflect101(x + k, w)];
                tmp[(size_t)y * w + x] = s;
            }
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                int s = 0;
                for (int k = -2; k <= 2; ++k) s += K[k + 2] * tmp[(size_t)reflect101(y + k, h) * w + x];
                int v = (s + (1 << 15)) >> 16;
                bl[(size_t)y * w + x] = (uint8_t)(v > 255 ? 255 : v);
            }
        dx.assign((size_t)w * h, 0); dy.assign((size_t)w * h, 0);
        auto B = [&](int y, int x) { return (int)bl[(size_t)reflect101(y, h) * w + reflect101(x, w)]; };
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                dx[(size_t)y * w + x] = (short)((B(y - 1, x + 1) + 2 * B(y, x + 1) + B(y + 1, x + 1)) - (B(y - 1, x - 1) + 2 * B(y, x - 1) + B(y + 1, x - 1)));
                dy[(size_t)y * w + x] = (short)((B(y + 1, x - 1) + 2 * B(y + 1, x) + B(y + 1, x + 1)) - (B(y - 1, x - 1) + 2 * B(y - 1, x) + B(y - 1, x + 1)));
            }
    }

    void describe(const PsoKeyLine& kl, float* desVec /*72*/, uint8_t* bin /*32*/) const {  // computeLBD :1027-1373
        const int NB = 9, WB = 7;
        const short heightOfLSP = (short)(WB * NB);
        float pL[9] = {0}, nL[9] = {0}, pL2[9] = {0}, nL2[9] = {0}, pO[9] = {0}, nO[9] = {0}, pO2[9] = {0}, nO2[9] = {0};
        const short realWidth = (short)w, imageWidth = (short)(realWidth - 1), imageHeight = (short)(h - 1);
        const short lengthOfLSP = (short)kl.numOfPixels;
        const short halfHeight = (short)((heightOfLSP - 1) / 2), halfWidth = (short)((lengthOfLSP - 1) / 2);
        const float midX = (float)(0.5 * (kl.sPointInOctaveX + kl.ePointInOctaveX));
        const float midY = (float)(0.5 * (kl.sPointInOctaveY + kl.ePointInOctaveY));
        float dL[2], dO[2];
        dL[0] = pso_cosf(kl.angle); dL[1] = pso_sinf(kl.angle);  // cos(float) -> float overload == libm cosf/sinf
        dO[0] = -dL[1]; dO[1] = dL[0];
        float sCorX0 = -dL[0] * halfWidth + dL[1] * halfHeight + midX;
        float sCorY0 = -dL[1] * halfWidth - dL[0] * halfHeight + midY;
        for (short hID = 0; hID < heightOfLSP; hID++) {
            float sCorX = sCorX0, sCorY = sCorY0;
            float pgdLRowSum = 0, ngdLRowSum = 0, pgdORowSum = 0, ngdORowSum = 0;
            for (short wID = 0; wID < lengthOfLSP; wID++) {
                short t = (short)std::round(sCorX);
                short xCor = (t < 0) ? 0 : (t > imageWidth) ? imageWidth : t;
                t = (short)std::round(sCorY);
                short yCor = (t < 0) ? 0 : (t > imageHeight) ? imageHeight : t;
                short gx = dx[(size_t)yCor * realWidth + xCor], gy = dy[(size_t)yCor * realWidth + xCor];
                float gDL = gx * dL[0] + gy * dL[1];
                float gDO = gx * dO[0] + gy * dO[1];
                if (gDL > 0) pgdLRowSum += gDL; else ngdLRowSum -= gDL;
                if (gDO > 0) pgdORowSum += gDO; else ngdORowSum -= gDO;
                sCorX += dL[0];
                sCorY += dL[1];
            }
            sCorX0 -= dL[1];
            sCorY0 += dL[0];
            float coef = (float)gaussG[hID];
            pgdLRowSum = coef * pgdLRowSum; ngdLRowSum = coef * ngdLRowSum;
            float pgdL2RowSum = pgdLRowSum * pgdLRowSum, ngdL2RowSum = ngdLRowSum * ngdLRowSum;
            pgdORowSum = coef * pgdORowSum; ngdORowSum = coef * ngdORowSum;
            float pgdO2RowSum = pgdORowSum * pgdORowSum, ngdO2RowSum = ngdORowSum * ngdORowSum;
            auto add = [&](int band, float c) {
                pL[band] += c * pgdLRowSum; nL[band] += c * ngdLRowSum;
                pL2[band] += c * c * pgdL2RowSum; nL2[band] += c * c * ngdL2RowSum;
                pO[band] += c * pgdORowSum; nO[band] += c * ngdORowSum;
                pO2[band] += c * c * pgdO2RowSum; nO2[band] += c * c * ngdO2RowSum;
            };
            short bandID = (short)(hID / WB);
            add(bandID, (float)gaussL[hID % WB + WB]);
            bandID--;
            if (bandID >= 0) add(bandID, (float)gaussL[hID % WB + 2 * WB]);
            bandID = bandID + 2;
            if (bandID < NB) add(bandID, (float)gaussL[hID % WB]);
        }
        const float invN2 = (float)(1.0 / (WB * 2.0)), invN3 = (float)(1.0 / (WB * 3.0));
        for (short b = 0; b < NB; b++) {
            const float invN = (b == 0 || b == NB - 1) ? invN2 : invN3;
            const int d = b * 8;
            float temp = pL[b] * invN;
            desVec[d] = temp; desVec[d + 4] = std::sqrt(pL2[b] * invN - temp * temp);
            temp = nL[b] * invN;
            desVec[d + 1] = temp; desVec[d + 5] = std::sqrt(nL2[b] * invN - temp * temp);
            temp = pO[b] * invN;
            desVec[d + 2] = temp; desVec[d + 6] = std::sqrt(pO2[b] * invN - temp * temp);
            temp = nO[b] * invN;
            desVec[d + 3] = temp; desVec[d + 7] = std::sqrt(nO2[b] * invN - temp * temp);
        }
        float tempM = 0, tempS = 0;
        for (int b = 0; b < NB; ++b) {
            const float* v = desVec + 8 * b;
            tempM += v[0] * v[0]; tempM += v[1] * v[1]; tempM += v[2] * v[2]; tempM += v[3] * v[3];
            tempS += v[4] * v[4]; tempS += v[5] * v[5]; tempS += v[6] * v[6]; tempS += v[7] * v[7];
        }
        tempM = 1 / std::sqrt(tempM);
        tempS = 1 / std::sqrt(tempS);
        for (int b = 0; b < NB; ++b) {
            float* v = desVec + 8 * b;
            v[0] *= tempM; v[1] *= tempM; v[2] *= tempM; v[3] *= tempM;
            v[4] *= tempS; v[5] *= tempS; v[6] *= tempS; v[7] *= tempS;
        }
        for (int i = 0; i < 72; i++) if (desVec[i] > 0.4) desVec[i] = (float)0.4;
        float temp = 0;
        for (int i = 0; i < 72; i++) temp += desVec[i] * desVec[i];
        temp = 1 / std::sqrt(temp);
        for (int i = 0; i < 72; i++) desVec[i] = desVec[i] * temp;
        for (int c = 0; c < 32; ++c) {  // binaryConversion :402-413
            const float* f1 = &desVec[8 * kCombos[c][0]];
            const float* f2 = &desVec[8 * kCombos[c][1]];
            uint8_t r = 0;
            for (int i = 0; i < 8; i++) if (f1[i] > f2[i]) r = (uint8_t)(r + (1 << i));
            bin[c] = r;
        }
    }
};

// ================================ pairing ================================
struct RotRect { float cx, cy, sw, sh, angle; };

double det2(float a, float b, float c, float d) { return (double)a * d - (double)b * c; }  // cv::determinant 2x2 CV_32F returns double

}  // namespace

extern "C" {

// 1 = LSD_REFINE_STD, 2 = LSD_REFINE_ADV (default); returns the previous mode
int pso_set_lsd_refine(int mode) { const int old = g_lsd_refine; if (mode == 1 || mode == 2) g_lsd_refine = mode; return old; }
// 0 = host libm in nfa() (default, what the reference calls), 1 = the product's restated functions; returns the previous mode
int pso_set_nfa_math(int restated) { const int old = g_nfa_math; g_nfa_math = restated ? 1 : 0; return old; }
// nfa(n, k, p) for an image of W x H (scaled) pixels, and the rectangle counts of rect_nfa: taps for the unit tests
double pso_lsd_nfa(int n, int k, double p, int W, int H) {
    Lsd l;
    l.LOG_NT = 5 * (std::log10(double(W)) + std::log10(double(H))) / 2 + std::log10(11.0);
    return l.nfa(n, k, p);
}
// the same with log(NT) given, and log_gamma alone: compared with the reference tree's twin (ED_Lib/NFA.cpp) in tests/test_oracle_ref_cpu.py
double pso_lsd_nfa_lognt(int n, int k, double p, double logNT) {
    Lsd l;
    l.LOG_NT = logNT;
    return l.nfa(n, k, p);
}
double pso_lsd_log_gamma(double x) { return Lsd::log_gamma(x); }
// rectangles that reach rect_improve (12 doubles each: x1 y1 x2 y2 width x y theta dx dy prec p); returns their number
int pso_lsd_rects(const uint8_t* gray, int w, int h, int stride, double* rects, int cap) {
    Lsd lsd;
    std::vector<double> r;
    std::vector<float> v;
    lsd.debug_rects = &r;
    const int keep = g_lsd_refine;
    g_lsd_refine = 2;
    lsd.detect(gray, w, h, stride, v);
    g_lsd_refine = keep;
    const int n = (int)r.size() / 12;
    for (int i = 0; i < n && i < cap; ++i) memcpy(rects + 12 * i, &r[12 * (size_t)i], 96);
    return n;
}

// growth log of every region_grow call (analysis of the queue order, tools/grow_stats.py); returns the int32 count
int pso_lsd_growlog(const uint8_t* gray, int w, int h, int stride, int32_t* out, int cap) {
    Lsd lsd;
    std::vector<int32_t> g;
    std::vector<float> v;
    lsd.debug_grow = &g;
    lsd.detect(gray, w, h, stride, v);
    g.push_back(-2); g.push_back((int32_t)lsd.debug_cands);
    const int n = (int)std::min<size_t>(g.size(), (size_t)cap);
    memcpy(out, g.data(), sizeof(int32_t) * (size_t)n);
    return (int)g.size();
}

// what the refinement of a frame's regions costs (analysis, tools/grow_stats.py): Lsd::debug_refine
void pso_lsd_refine_stats(const uint8_t* gray, int w, int h, int stride, long long* out) {
    Lsd lsd;
    std::vector<float> v;
    lsd.detect(gray, w, h, stride, v);
    memcpy(out, lsd.debug_refine, sizeof lsd.debug_refine);
}

// LSD + contrib wrapper clamp: returns number of segments, lines = x1,y1,x2,y2 floats
int pso_lsd_detect(const uint8_t* gray, int w, int h, int stride, float* lines, int cap) {
    if (!gray || w <= 0 || h <= 0) return 0;
    Lsd lsd;
    std::vector<float> v;
    lsd.detect(gray, w, h, stride, v);
    int n = (int)v.size() / 4;
    for (int i = 0; i < n; ++i) clamp_extremes(&v[4 * i], w, h);
    for (int i = 0; i < n && i < cap; ++i) memcpy(lines + 4 * i, &v[4 * i], 16);
    return n;
}

// LSD stage taps (scaled image, angle in degrees-as-double*DEG_TO_RADS, modgrad)
int pso_lsd_gradient(const uint8_t* gray, int w, int h, int stride, double* scaled, double* angles, double* modgrad, int* W, int* H) {
    Lsd lsd;
    lsd.scale_image(gray, w, h, stride);
    lsd.ll_angle(2.0 / std::sin(kPI * 22.5 / 180));
    *W = lsd.W; *H = lsd.H;
    if (scaled) memcpy(scaled, lsd.scaled.data(), lsd.scaled.size() * 8);
    if (angles) memcpy(angles, lsd.angles.data(), lsd.angles.size() * 8);
    if (modgrad) memcpy(modgrad, lsd.modgrad.data(), lsd.modgrad.size() * 8);
    return 0;
}

int pso_merge_lines(const float* src, int n, float ang, float dist, float ep, float* dst, int cap) {
    V4 s(src, src + 4 * (size_t)n), d;
    merge_lines(s, d, ang, dist, ep);
    int m = (int)d.size() / 4;
    for (int i = 0; i < m && i < cap; ++i) memcpy(dst + 4 * i, &d[4 * i], 16);
    return m;
}

// optimizeAndMergeLines_lsd (:449-485): lines in/out as KeyLines
int pso_optimize_and_merge(const float* src, int n, int w, int h, PsoKeyLine* out, int cap) {
    V4 s(src, src + 4 * (size_t)n), t, d;
    merge_lines(s, t, 0.05f, 5, 15);
    filter_short(t, 30);
    merge_lines(t, d, 0.03f, 3, 30);
    filter_short(d, 50);
    std::vector<PsoKeyLine> kls;
    vec4f_to_keyline(d, w, h, kls);
    for (int i = 0; i < (int)kls.size() && i < cap; ++i) out[i] = kls[i];
    return (int)kls.size();
}

int pso_line_iterator_count(int w, int h, float x1, float y1, float x2, float y2) { return line_iterator_count(w, h, x1, y1, x2, y2); }

// BinaryDescriptor::compute on given keylines: desc n x 32, optional float descriptors n x 72
int pso_lbd_compute(const uint8_t* gray, int w, int h, int stride, const PsoKeyLine* kls, int n, uint8_t* desc, float* fdesc) {
    if (n <= 0) return 0;
    Lbd lbd;
    lbd.prepare(gray, w, h, stride);
    for (int i = 0; i < n; ++i) {
        float v[72];
        lbd.describe(kls[i], v, desc + (size_t)i * 32);
        if (fdesc) memcpy(fdesc + (size_t)i * 72, v, sizeof(v));
    }
    return n;
}

int pso_lbd_sobel(const uint8_t* gray, int w, int h, int stride, short* dx, short* dy) {
    Lbd lbd;
    lbd.prepare(gray, w, h, stride);
    memcpy(dx, lbd.dx.data(), (size_t)w * h * 2);
    memcpy(dy, lbd.dy.data(), (size_t)w * h * 2);
    return 0;
}

// LINEextractor::operator() (add_src/LineExtractor.cpp:325-366): keylines, descriptors, 2-D line equations
int pso_line_extract(const uint8_t* gray, int w, int h, int stride, int nLSDFeature, PsoKeyLine* kls, uint8_t* desc,
                     double* lineEq, int cap) {
    if (!gray || w <= 0 || h <= 0) return 0;
    std::vector<float> seg((size_t)4 * 65536);
    int n = pso_lsd_detect(gray, w, h, stride, seg.data(), 65536);
    std::vector<PsoKeyLine> k(65536);
    int m = pso_optimize_and_merge(seg.data(), n, w, h, k.data(), 65536);
    k.resize(m);
    if ((int)k.size() > nLSDFeature) {
        std::stable_sort(k.begin(), k.end(), [](const PsoKeyLine& a, const PsoKeyLine& b) { return a.response > b.response; });
        k.resize(nLSDFeature);
        for (int i = 0; i < nLSDFeature; i++) k[i].class_id = i;
    }
    m = (int)k.size();
    if (m > cap) return -1;
    if (m > 0) pso_lbd_compute(gray, w, h, stride, k.data(), m, desc, nullptr);
    for (int i = 0; i < m; ++i) {
        kls[i] = k[i];
        const double sx = k[i].startPointX, sy = k[i].startPointY, ex = k[i].endPointX, ey = k[i].endPointY;
        const double lx = sy * 1.0 - 1.0 * ey, ly = 1.0 * ex - sx * 1.0, lz = sx * ey - sy * ex;  // sp x ep
        const double nrm = std::sqrt(lx * lx + ly * ly);
        lineEq[3 * i] = lx / nrm; lineEq[3 * i + 1] = ly / nrm; lineEq[3 * i + 2] = lz / nrm;
    }
    return m;
}

// CPartiallyRecoverConnectivity(mLines, radius, fans, img, fanThr): fans rows (x, y, i, j)
int pso_lil_pair(const float* L, int rows, float radius, float fanThr, int imgCols, int imgRows, float* fans, int cap) {
    std::vector<float> out;
    const int npts = 2 * rows;
    for (int i = 0; i < rows; i++) {
        const float* pdat = L + 4 * i;
        float cenx = (pdat[0] + pdat[2]) / 2, ceny = (pdat[1] + pdat[3]) / 2;
        float dy = pdat[3] - pdat[1], dx = pdat[2] - pdat[0];
        float degAng = pso_fast_atan2(dy, dx);
        float arcAng = (float)(degAng / 180 * kPI);
        float length = std::abs(std::tan(arcAng)) > 1 ? std::abs(dy) : std::abs(dx);
        int th = (int)(radius * 2), tw = (int)(length + 2 * radius);  // CvSize is integer
        RotRect rr = {cenx, ceny, (float)tw, (float)th, degAng};
        // ptsDropInRotatedRect: evaluated by cv::addWeighted after MatExpr folds the scalars:
        //   fposx = x*dcos + y*dsin + (float)(-cx*dcos - cy*dsin),  fposy = x*dsin + y*(-dcos) + (float)(-cx*dsin + cy*dcos)
        float hafW = rr.sw / 2, hafH = rr.sh / 2;
        float angle = (float)(rr.angle * kPI / 180);
        float dsin = pso_sinf(angle), dcos = pso_cosf(angle);
        const float gx = (float)(-(double)cenx * (double)dcos - (double)ceny * (double)dsin);
        const float gy = (float)(-(double)cenx * (double)dsin + (double)ceny * (double)dcos);
        const float ndcos = (float)(-(double)dcos);
        for (int pj = 0; pj < npts; ++pj) {
            const float px = pj < rows ? L[4 * pj] : L[4 * (pj - rows) + 2];
            const float py = pj < rows ? L[4 * pj + 1] : L[4 * (pj - rows) + 3];
            const float fposx = (px * dcos + py * dsin) + gx;
            const float fposy = (px * dsin + py * ndcos) + gy;
            if (!(-hafW <= fposx && fposx < hafW && -hafH <= fposy && fposy < hafH)) continue;
            int curSer = pj >= rows ? pj - rows : pj;
            if (curSer == i) continue;
            const float* pdat1 = L + 4 * curSer;
            float dy1 = pdat1[3] - pdat1[1], dx1 = pdat1[2] - pdat1[0];
            float degAng1 = pso_fast_atan2(dy1, dx1);
            float arcAng1 = (float)(degAng1 / 180 * kPI);
            float tmp = (float)kPI;
            float tmpa = std::fmod(std::abs(arcAng - arcAng1), tmp);
            if (tmpa < fanThr || kPI - tmpa < fanThr) continue;
            // intersectionOfLines
            float A1 = pdat[1] - pdat[3], B1 = pdat[2] - pdat[0], C1 = pdat[3] * pdat[0] - pdat[1] * pdat[2];
            float A2 = pdat1[1] - pdat1[3], B2 = pdat1[2] - pdat1[0], C2 = pdat1[3] * pdat1[0] - pdat1[1] * pdat1[2];
            float D = (float)det2(A1, B1, A2, B2);              // float D = determinant(tmat1)
            float X = (float)(det2(-C1, B1, -C2, B2) / D);      // double / float -> double -> float
            float Y = (float)(det2(A1, -C1, A2, -C2) / D);
            // isPtInRotatedRect (scalar float arithmetic)
            float fx = dcos * (X - rr.cx) + dsin * (Y - rr.cy);
            float fy = dsin * (X - rr.cx) - dcos * (Y - rr.cy);
            bool inr = (-hafW <= fx && fx < hafW && -hafH <= fy && fy < hafH);
            if (inr && (X >= 4 && X < imgCols - 4 && Y >= 4 && Y < imgRows - 4)) {
                out.push_back(X); out.push_back(Y); out.push_back((float)i); out.push_back((float)curSer);
            }
        }
    }
    int nfan = (int)out.size() / 4, kept = 0;
    for (int i = 0; i < nfan; i++) {  // keep the LAST occurrence of every unordered pair (:109-131)
        int s1 = (int)out[4 * i + 2], s2 = (int)out[4 * i + 3];
        bool flag = true;
        for (int j = i + 1; j < nfan; j++) {
            int s3 = (int)out[4 * j + 2], s4 = (int)out[4 * j + 3];
            if ((s1 == s3 && s2 == s4) || (s1 == s4 && s2 == s3)) { flag = false; break; }
        }
        if (flag) {
            if (kept < cap) memcpy(fans + 4 * kept, &out[4 * i], 16);
            ++kept;
        }
    }
    return kept;
}

}  // extern "C"
