// LSD_REFINE_ADV for the LSD of the line front-end (gfx950, wave64). Product code.
// Reference behaviour reproduced: the `doRefine >= LSD_REFINE_ADV` step of OpenCV 3.x lsd.cpp flsd() -
// rect_improve(), rect_nfa(), nfa(), log_gamma() - which the stock contrib LSDDetector the reference links
// (add_src/LineExtractor.cpp:336-337, CMakeLists.txt:96) selects; lsd.cpp is not in the reference tree, the
// nfa() / log_gamma() arithmetic has a twin there: Thirdparty/line_descriptor/src/ED_Lib/NFA.cpp:106-240.
//
// Decomposition.  The validation reads the level-line angles and the rectangle, never the `used` map, and a
// rejected rectangle releases nothing: it does not feed back into the region growing.  k_lsd_grow4 therefore only
// records the rectangles (in seed order) and this file validates them afterwards with ONE WAVE PER RECTANGLE -
// thousands of independent waves per launch instead of a longer serial chain per frame:
//   k_lsd_nfa_count / k_lsd_nfa_eval  one pair of launches per rect_improve phase (see below): pixel scans with one wave per
//               (rectangle, trial), nfa() with one THREAD per (rectangle, trial);
//   k_lsd_emit  workgroup = frame: ordered compaction of the accepted segments.
// rect_nfa() is reproduced as OpenCV 3.x behaves, not as the paper describes it: corners truncated to int, edge
// slopes by INTEGER division, the second slopes use `tailp->p.x` where the y coordinate was meant, and a row outside
// the image is skipped without advancing the column bounds (the bounds are therefore closed forms of the row index).
#ifndef PSL_LINE_KERNELS3_H
#define PSL_LINE_KERNELS3_H

#include "line_kernels.h"

struct LsdnRect { double x1, y1, x2, y2, width, theta, dx, dy, prec, p; };
struct LsdnGeom {
    int ya, yb;            // in-image rows of the scan
    int lefty, righty;     // rows from which the second slopes apply
    long long minx, fl, sl, fr, sr;
};

__device__ __forceinline__ void lsdn_cswap(long long& a, long long& b) { const long long lo = a < b ? a : b, hi = a < b ? b : a; a = lo; b = hi; }

// the edge bookkeeping of rect_nfa() up to the row loop
__device__ __forceinline__ void lsdn_geom(const LsdnRect& r, int H, LsdnGeom* G) {
    const double half_width = r.width / 2.0;
    const double dyhw = PSL_DMUL(r.dy, half_width), dxhw = PSL_DMUL(r.dx, half_width);
    int cx[4] = {(int)PSL_DSUB(r.x1, dyhw), (int)PSL_DSUB(r.x2, dyhw), (int)PSL_DADD(r.x2, dyhw), (int)PSL_DADD(r.x1, dyhw)};
    int cy[4] = {(int)PSL_DADD(r.y1, dxhw), (int)PSL_DADD(r.y2, dxhw), (int)PSL_DSUB(r.y2, dxhw), (int)PSL_DSUB(r.y1, dxhw)};
    // std::sort by (x, y): lexicographic order on one 64-bit key per corner (identical corners are interchangeable)
    long long k[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) k[i] = (long long)cx[i] * 4294967296ll + ((long long)cy[i] + 2147483648ll);
    lsdn_cswap(k[0], k[1]); lsdn_cswap(k[2], k[3]); lsdn_cswap(k[0], k[2]); lsdn_cswap(k[1], k[3]); lsdn_cswap(k[1], k[2]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long q = k[i] >> 32;  // floor: the low part is non-negative
        cx[i] = (int)q;
        cy[i] = (int)((k[i] - q * 4294967296ll) - 2147483648ll);
    }
    // corner i by a runtime index, without dynamic register indexing
    auto X = [&](int i) { return i == 0 ? cx[0] : i == 1 ? cx[1] : i == 2 ? cx[2] : cx[3]; };
    auto Y = [&](int i) { return i == 0 ? cy[0] : i == 1 ? cy[1] : i == 2 ? cy[2] : cy[3]; };
    int mn = 0, mx = 0;      // first corner with the smallest / largest y
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        if (Y(mn) > cy[i]) mn = i;
        if (Y(mx) < cy[i]) mx = i;
    }
    unsigned taken = 1u << mn;
    int left = -1;           // leftmost of the rest (first on ties), then the rightmost (first on ties), then the last one
#pragma unroll
    for (int i = 0; i < 4; ++i) if (!((taken >> i) & 1u)) { if (left < 0) left = i; else if (X(left) > cx[i]) left = i; }
    taken |= 1u << left;
    int right = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (!((taken >> i) & 1u)) { if (right < 0) right = i; else if (X(right) < cx[i]) right = i; }
    taken |= 1u << right;
    int tail = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (!((taken >> i) & 1u)) tail = i;
    const int mnx = X(mn), mny = Y(mn), lx = X(left), ly = Y(left), rx = X(right), ry = Y(right), tx = X(tail), mxy = Y(mx);
    G->fl = (mny != ly) ? (mnx - lx) / (mny - ly) : 0;
    G->sl = (ly != tx) ? (lx - tx) / (ly - tx) : 0;     // `tailp->p.x`, as upstream
    G->fr = (mny != ry) ? (mnx - rx) / (mny - ry) : 0;
    G->sr = (ry != tx) ? (rx - tx) / (ry - tx) : 0;
    G->minx = mnx;
    G->lefty = ly; G->righty = ry;
    G->ya = mny < 0 ? 0 : mny;
    G->yb = mxy >= H ? H - 1 : mxy;
}

// column bounds of in-image row y: the steps added after the in-image rows ya .. y-1 (first slope before the row of the
// left / right corner, second slope from it on), clipped to the image
__device__ __forceinline__ void lsdn_row_span(const LsdnGeom& G, int y, int W, int* xa, int* xb) {
    const int n = y - G.ya;
    int c1 = (y < G.lefty ? y : G.lefty) - G.ya; c1 = c1 < 0 ? 0 : c1;
    int d1 = (y < G.righty ? y : G.righty) - G.ya; d1 = d1 < 0 ? 0 : d1;
    const long long L = G.minx + G.fl * c1 + G.sl * (n - c1);
    const long long R = G.minx + G.fr * d1 + G.sr * (n - d1);
    *xa = (int)(L < 0 ? 0 : (L > W ? W : L));
    *xb = (int)(R >= W ? W - 1 : (R < -1 ? -1 : R));
}

// sums / maxima over aligned groups of GL lanes (GL = 16: four scans per wave)
// From how many rows on a lane of a scan group takes a whole row (fewer rows: the pixels as slots dealt round-robin / row after row with the lanes
// sharing it).  Until round 3 both were GL = 16 ("enough rows to occupy the group"): but a row's spans cost ~25 instructions per trial and are paid once per
// row by the lane that owns it, against once per row and step by every lane of the group - the rectangles of the later phases are small (a few rows of
// 5 - 20 pixels) and run better with most lanes idle.  12288 dense frames, A/B in one session (profiles/r03z_ab_byrow.log): trials 16: 45.9 ms,
// 6: 41.3, 3: 40.8; single-geometry scans 16: 45.9, 6: 45.0, 3: 44.9.
#ifndef PSL_NFA_BYROW_MIN
#define PSL_NFA_BYROW_MIN 4   // lsdn_count
#endif
#ifndef PSL_NFA_BYROW_T
#define PSL_NFA_BYROW_T 3     // lsdn_count_trials
#endif
template <int GL>
__device__ __forceinline__ int lsdn_group_sum(int v) {
#pragma unroll
    for (int o = GL / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int GL>
__device__ __forceinline__ int lsdn_group_max(int v) {
#pragma unroll
    for (int o = GL / 2; o > 0; o >>= 1) { const int u = __shfl_xor(v, o); v = u > v ? u : v; }
    return v;
}

// total_pts and, for NP tolerances, alg_pts of rect_nfa().  The scan is a few hundred pixels whose angles sit in L2 / L1: what
// it costs is memory latency, so the pixels are enumerated as slots (row, offset < widest span) dealt to the lanes round-robin
// and every lane has four loads in flight - not one dependent load per pixel in a per-row or per-column loop.  A scan is done by
// a group of GL lanes (`lane` = the lane's index in its group); the groups of a wave scan different rectangles, each lane running
// its own loops, and only the final sums cross lanes (inside the group).
template <int NP, int GL>
__device__ __forceinline__ void lsdn_count(const float* __restrict__ ang, int W, const LsdnGeom& G, double theta, const double* prec, int lane, int* total, int* alg) {
    int al[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) al[j] = 0;
    // isAligned(x, y, theta, prec) per pixel is eight double-precision operations (half rate).  The angle map holds f32 degrees
    // and the double value is exactly (double)deg * pi/180, so the folded difference is first formed in f32 degrees: its error
    // against the exact double evaluation is below 1e-3 degrees (theta <= 3 pi: 6e-5 from the conversion, 3e-5 per f32 operation),
    // and only a pixel within that margin of a tolerance is decided by the exact arithmetic.
    const float theta_deg = (float)(theta * 57.295779513082320877);
    float pdeg[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) pdeg[j] = (float)(prec[j] * 57.295779513082320877);
    auto pixel = [&](float a) {
        if (a == PSL_LSD_NOTDEF) return;
        float d = __builtin_fabsf(theta_deg - a);
        const float d2 = __builtin_fabsf(d - 360.0f);
        d = d > 270.0f ? d2 : d;
        bool close = __builtin_fabsf(d - 270.0f) < 2e-3f;   // the fold itself is decided on the exact value near its switch point
#pragma unroll
        for (int j = 0; j < NP; ++j) close = close || __builtin_fabsf(d - pdeg[j]) < 2e-3f;
        if (!close) {
#pragma unroll
            for (int j = 0; j < NP; ++j) al[j] += d <= pdeg[j] ? 1 : 0;
        } else {
            const double f = lsdg_fold(PSL_DMUL((double)a, PSL_DEG2RAD), theta);
#pragma unroll
            for (int j = 0; j < NP; ++j) al[j] += f <= prec[j] ? 1 : 0;
        }
    };
    const int nrows = G.yb - G.ya + 1;
    int tot = 0;
    if (nrows >= PSL_NFA_BYROW_MIN) {   // lane = row: the span is computed once per row, four loads of the row in flight
        for (int y = G.ya + lane; y <= G.yb; y += GL) {
            int xa, xb;
            lsdn_row_span(G, y, W, &xa, &xb);
            const int c = xb - xa + 1;
            tot += c > 0 ? c : 0;
            const float* row = ang + y * W;
            for (int x = xa; x <= xb; x += 4) {
                float a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = x + u <= xb ? row[x + u] : PSL_LSD_NOTDEF;
#pragma unroll
                for (int u = 0; u < 4; ++u) pixel(a[u]);
            }
        }
        *total = lsdn_group_sum<GL>(tot);
    } else {             // few rows: the pixels as slots (row, offset < widest span) dealt to the lanes round-robin
        int wmax = 0;
        for (int y = G.ya + lane; y <= G.yb; y += GL) {
            int xa, xb;
            lsdn_row_span(G, y, W, &xa, &xb);
            const int c = xb - xa + 1;
            tot += c > 0 ? c : 0;
            wmax = c > wmax ? c : wmax;
        }
        *total = lsdn_group_sum<GL>(tot);
        wmax = lsdn_group_max<GL>(wmax);
        if (nrows > 0 && wmax > 0) {
            const int nslots = nrows * wmax, row_step = GL / wmax, dx_step = GL - row_step * wmax;
            int row = lane / wmax, dx = lane - row * wmax;
            for (int s = lane; s < nslots; s += 4 * GL) {
                float a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a[u] = PSL_LSD_NOTDEF;
                    if (s + GL * u < nslots) {
                        int xa, xb;
                        const int y = G.ya + row;
                        lsdn_row_span(G, y, W, &xa, &xb);
                        if (dx <= xb - xa) a[u] = ang[y * W + xa + dx];
                    }
                    dx += dx_step; row += row_step;
                    if (dx >= wmax) { dx -= wmax; ++row; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) pixel(a[u]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) alg[j] = lsdn_group_sum<GL>(al[j]);
}

// The five trial rectangles of a width-reduction / side-reduction phase of rect_improve() differ by quarter- and half-pixel steps:
// their pixel scans cover nearly the same pixels, with the same theta and tolerance.  ONE pass over the union of their row spans:
// a pixel's angle is loaded and tested for alignment once, and counted for every trial whose span of that row holds it (the spans
// are rect_nfa's own, per trial: corners truncated to int, integer slopes - membership is not a function of the distance to the
// axis).  total_pts needs no pixel at all: it is the sum of the span lengths.  The five geometries live in LDS as nine 32-bit
// integers each (corner coordinates are a few hundred, slopes and row counts likewise: the closed form of a row's bounds stays
// far inside 32 bits), written by lanes 0 - 4 of the group - lane t builds trial t - so that the scan itself runs in ~60
// registers: with the geometries in registers the kernel took 122 and half the waves per SIMD, and was slower than five scans.
struct LsdnGeomI { int ya, yb, lefty, righty, minx, fl, sl, fr, sr; };   // yb < ya: no in-image row; yb == -2: excluded by the width guard
typedef __attribute__((address_space(3))) const int lds_cint;

__device__ __forceinline__ void lsdn_row_span_i(lds_cint* g, int y, int W, int* xa, int* xb) {
    const int ya = g[0], lefty = g[2], righty = g[3], minx = g[4];
    const int n = y - ya;
    int c1 = (y < lefty ? y : lefty) - ya; c1 = c1 < 0 ? 0 : c1;
    int d1 = (y < righty ? y : righty) - ya; d1 = d1 < 0 ? 0 : d1;
    const int L = minx + g[5] * c1 + g[6] * (n - c1);
    const int R = minx + g[7] * d1 + g[8] * (n - d1);
    *xa = L < 0 ? 0 : (L > W ? W : L);
    *xb = R >= W ? W - 1 : (R < -1 ? -1 : R);
}

template <int GL>
__device__ __forceinline__ void lsdn_count_trials(const float* __restrict__ ang, int W, lds_cint* geo, double theta, double prec, int lane, int* total, int* alg) {
    const float theta_deg = (float)(theta * 57.295779513082320877), pdeg = (float)(prec * 57.295779513082320877);
    auto aligned = [&](float a) -> bool {   // isAligned: f32 degrees first, the reference's f64 arithmetic inside the margin (see lsdn_count)
        if (a == PSL_LSD_NOTDEF) return false;
        float d = __builtin_fabsf(theta_deg - a);
        const float d2 = __builtin_fabsf(d - 360.0f);
        d = d > 270.0f ? d2 : d;
        if (__builtin_fabsf(d - 270.0f) < 2e-3f || __builtin_fabsf(d - pdeg) < 2e-3f) return lsdg_fold(PSL_DMUL((double)a, PSL_DEG2RAD), theta) <= prec;
        return d <= pdeg;
    };
    int ya = 0, yb = -1;   // union of the trials' in-image rows (empty: no trial has one)
    bool any = false;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int a = geo[9 * t], b = geo[9 * t + 1];
        if (b >= a) {
            ya = !any || a < ya ? a : ya;
            yb = !any || b > yb ? b : yb;
            any = true;
        }
    }
    int tot[5], al[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) { tot[t] = 0; al[t] = 0; }
    const bool by_row = yb - ya + 1 >= PSL_NFA_BYROW_T;   // lane = row (four loads of the row in flight), or - few, long rows - row after row with the lanes sharing it
    for (int y = by_row ? ya + lane : ya; y <= yb; y += by_row ? GL : 1) {
        int xa[5], xb[5], ua = 0x7fffffff, ub = -1;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            xa[t] = 1; xb[t] = 0;
            if (y >= geo[9 * t] && y <= geo[9 * t + 1]) lsdn_row_span_i(geo + 9 * t, y, W, &xa[t], &xb[t]);
            if (xb[t] >= xa[t]) {
                if (by_row || lane == 0) tot[t] += xb[t] - xa[t] + 1;
                ua = xa[t] < ua ? xa[t] : ua; ub = xb[t] > ub ? xb[t] : ub;
            }
        }
        if (ub < ua) continue;   // no trial has a pixel in this row (ua is still the sentinel: never form an address from it)
        const float* row = ang + y * W;
        const int step = by_row ? 1 : GL;
        for (int x = by_row ? ua : ua + lane; x <= ub; x += 4 * step) {
            float a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = x + step * u <= ub ? row[x + step * u] : PSL_LSD_NOTDEF;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (aligned(a[u])) {
#pragma unroll
                    for (int t = 0; t < 5; ++t) al[t] += (x + step * u >= xa[t] && x + step * u <= xb[t]) ? 1 : 0;
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 5; ++t) { total[t] = lsdn_group_sum<GL>(tot[t]); alg[t] = lsdn_group_sum<GL>(al[t]); }
}

// ---- nfa() ------------------------------------------------------------------------------------------------------------
// pow(x, n) for the integer-valued arguments log_gamma sees: exact products where libm's pow is exact as well (x <= 15,
// n <= 6); x^6 = (x^3)^2 with x^3 exact, i.e. one rounding - what a pow with < 1 ulp of error returns - for the Windschitl term
__host__ __device__ static inline double lsdn_log_gamma(double x) {
    if (x > 15.0) {
        const double c = PSL_DMUL(PSL_DMUL(x, x), x), x6 = PSL_DMUL(c, c);
        const double inner = PSL_DADD(PSL_DMUL(x, psl_sinh_small(1 / x)), 1 / PSL_DMUL(810.0, x6));
        return PSL_DADD(PSL_DSUB(PSL_DADD(0.918938533204673, PSL_DMUL(PSL_DSUB(x, 0.5), psl_log(x))), x), PSL_DMUL(PSL_DMUL(0.5, x), psl_log(inner)));
    }
    const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
    double a = PSL_DSUB(PSL_DMUL(PSL_DADD(x, 0.5), psl_log(PSL_DADD(x, 5.5))), PSL_DADD(x, 5.5));
    double b = 0, xn = 1;
#pragma unroll
    for (int n = 0; n < 7; ++n) {
        a = PSL_DSUB(a, psl_log(PSL_DADD(x, (double)n)));
        b = PSL_DADD(b, PSL_DMUL(q[n], xn));
        xn = PSL_DMUL(xn, x);
    }
    return PSL_DADD(a, psl_log(b));
}

__device__ __forceinline__ bool lsdn_double_equal0(double a) {  // double_equal(a, 0)
    if (a == 0.0) return true;
    const double aa = fabs(a);
    const double abs_max = aa < 2.2250738585072014e-308 ? 2.2250738585072014e-308 : aa;
    return (aa / abs_max) <= PSL_DMUL(100.0, 2.2204460492503131e-16);
}

// What nfa() needs of its transcendental functions, tabulated once per geometry by the host WITH THE SAME FUNCTIONS (plain IEEE
// operations, no contraction: host and device results are bit-identical, pslfe_line.hip: prepare): log_gamma of every integer
// argument up to lg_n - 1 (the arguments are pixel counts + 1) and log(p), log(1 - p), log10(p) for p = p0 / 2^j, the only
// probabilities rect_improve can reach (j <= 10).  An nfa() then costs three loads and an exp instead of eight logarithms.
#define PSL_NFA_NP 11
struct LsdnTables {
    const double* lg;     // [lg_n]
    const double* logs;   // [3][PSL_NFA_NP]: log(p_j), log(1 - p_j), log10(p_j) - in HBM, not in the kernel argument: a lane indexes them by its own j
    int lg_n;
    double p0, log_nt;
    const double* inv;    // [PSL_RATIO_BMAX]: 1 / i, correctly rounded (psl_ratio_inv)
};

__device__ __forceinline__ double lsdn_lg(const LsdnTables& T, int i) { return i < T.lg_n ? T.lg[i] : lsdn_log_gamma((double)i); }

// The truncation test of the binomial tail, `err < tolerance * |-log10(bin_tail) - logNT| * bin_tail` with
// err = term * ((1 - m^q) / (1 - m) - 1) = term * (m - m^q) / (1 - m), is what an iteration of nfa() costs (a pow and a log10 in
// double: ~500 instructions beside ~15 for the recurrence).  It only decides WHERE the series stops, so it is decided from
// rigorous bounds first, in two stages, and evaluated exactly only when they leave it open:
//   stage 1 (no transcendental): m <= (m - m^q) / (1 - m) <= m / (1 - m) for q >= 2 (m < 1/7), and log10(bin_tail) lies between
//           its binade's end points; all but the last few terms before the stop are decided here;
//   stage 2: m^q and log2 of the mantissa from the hardware's f32 exp2 / log2: err within 1e-3 relative (m - m^q and 1 - m do
//           not cancel), the logarithm within 1e-5 absolute.
// A comparison that the margins decide is the comparison the exact arithmetic makes.  returns 1 = stop, 0 = go on, -1 = undecided
__device__ __forceinline__ int lsdn_tail_test_fast(double term, double bin_tail, double mult_term, int q, double log_nt) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(bin_tail);
    const int e = (int)((bits >> 52) & 0x7ff) - 1023;
    if (e <= -1022 || e >= 1024 || q < 2) return -1;  // subnormal / non-finite sums, and q == 1 (err is an exact 0 there): exact path
    const double tb = 0.1 * bin_tail;
    {   // stage 1
        const double a0 = -(double)e * 0.30102999566398120 - log_nt, a1 = -(double)(e + 1) * 0.30102999566398120 - log_nt;  // -log10(bt) - logNT in (a1, a0]
        const double f0 = fabs(a0), f1 = fabs(a1);
        const double Lhi = (f0 > f1 ? f0 : f1) + 1e-9, Llo = (a0 > 0) == (a1 > 0) ? (f0 < f1 ? f0 : f1) - 1e-9 : 0.0;
        // the reference's `(1 - pow) / (1 - m) - 1` cancels to about m with an ABSOLUTE rounding error of a few 1e-16: the slack
        // 5e-16 covers it (it matters for tiny m); 1 / (1 - m) <= 1 + 1.2 m for m <= 1/7
        const double errL = term * (mult_term - 5e-16), errU = term * (mult_term * (1.0 + 1.2 * mult_term) + 5e-16);
        if (errU * 1.000001 < tb * (Llo > 0 ? Llo : 0.0)) return 1;
        if (errL * 0.999999 >= tb * Lhi) return 0;
    }
    const float m = (float)mult_term;
    const float pw = q == 2 ? m * m : __builtin_amdgcn_exp2f((float)q * __builtin_amdgcn_logf(m));
    const float A = (m - pw) / (1.0f - m);
    const double errE = term * (double)A, slack = term * 5e-16;
    const float fm = (float)__longlong_as_double((long long)((bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull));
    const double l10 = ((double)e + (double)__builtin_amdgcn_logf(fm)) * 0.30102999566398120;
    const double Lv = fabs(-l10 - log_nt);
    const double lo = Lv - 1e-5, hi = Lv + 1e-5;
    if (errE * 1.001 + slack < tb * (lo > 0 ? lo : 0.0)) return 1;
    if (errE * 0.999 - slack >= tb * hi) return 0;
    return -1;
}

__device__ __forceinline__ double lsdn_nfa(const LsdnTables& T, int n, int k, double p) {
    const double log_nt = T.log_nt;
    if (n == 0 || k == 0) return -log_nt;
    // p = p0 / 2^j: the table row, or -1 (a probability rect_improve cannot produce: evaluated directly)
    int j = (int)((__double_as_longlong(T.p0) >> 52) & 0x7ff) - (int)((__double_as_longlong(p) >> 52) & 0x7ff);
    if (j < 0 || j >= PSL_NFA_NP || __longlong_as_double(__double_as_longlong(T.p0) - ((long long)j << 52)) != p) j = -1;
    if (n == k) return PSL_DSUB(-log_nt, PSL_DMUL((double)n, j >= 0 ? T.logs[2 * PSL_NFA_NP + j] : psl_log10(p)));
    const double p_term = p / PSL_DSUB(1.0, p);
    double log1term = PSL_DSUB(PSL_DSUB(lsdn_lg(T, n + 1), lsdn_lg(T, k + 1)), lsdn_lg(T, n - k + 1));
    log1term = PSL_DADD(PSL_DADD(log1term, PSL_DMUL((double)k, j >= 0 ? T.logs[j] : psl_log(p))),
                        PSL_DMUL((double)(n - k), j >= 0 ? T.logs[PSL_NFA_NP + j] : psl_log(PSL_DSUB(1.0, p))));
    double term = psl_exp(log1term);
    if (lsdn_double_equal0(term)) {
        if ((double)k > PSL_DMUL((double)n, p)) return PSL_DSUB(-log1term / 2.30258509299404568402, log_nt);
        return -log_nt;
    }
    double bin_tail = term;
    for (int i = k + 1; i <= n; ++i) {
        const double bin_term = (double)(n - i + 1) / (double)i;
        const double mult_term = PSL_DMUL(bin_term, p_term);
        term = PSL_DMUL(term, mult_term);
        bin_tail = PSL_DADD(bin_tail, term);
        if (bin_term < 1) {
            const int q = n - i + 1;
            int stop = lsdn_tail_test_fast(term, bin_tail, mult_term, q, log_nt);
            if (stop < 0) {
                const double pw = q == 1 ? mult_term : psl_pow_pos(mult_term, (double)q);  // pow(x, 1) is exact in any libm
                const double err = PSL_DMUL(term, PSL_DSUB(PSL_DSUB(1.0, pw) / PSL_DSUB(1.0, mult_term), 1.0));
                stop = err < PSL_DMUL(PSL_DMUL(0.1, fabs(PSL_DSUB(-psl_log10(bin_tail), log_nt))), bin_tail) ? 1 : 0;
            }
            if (stop) break;
        }
    }
    return PSL_DSUB(-psl_log10(bin_tail), log_nt);
}

// one step of the cumulative change a rect_improve phase applies to its trial rectangle (phase 0: narrower, 1 / 2: one side)
__device__ __forceinline__ void lsdn_shrink(LsdnRect& r, int phase) {
    const double delta = 0.5, delta_2 = 0.25;
    if (phase == 1) {
        r.x1 = PSL_DADD(r.x1, PSL_DMUL(-r.dy, delta_2)); r.y1 = PSL_DADD(r.y1, PSL_DMUL(r.dx, delta_2));
        r.x2 = PSL_DADD(r.x2, PSL_DMUL(-r.dy, delta_2)); r.y2 = PSL_DADD(r.y2, PSL_DMUL(r.dx, delta_2));
    } else if (phase == 2) {
        r.x1 = PSL_DSUB(r.x1, PSL_DMUL(-r.dy, delta_2)); r.y1 = PSL_DSUB(r.y1, PSL_DMUL(r.dx, delta_2));
        r.x2 = PSL_DSUB(r.x2, PSL_DMUL(-r.dy, delta_2)); r.y2 = PSL_DSUB(r.y2, PSL_DMUL(r.dx, delta_2));
    }
    r.width = PSL_DSUB(r.width, delta);
}

// ---- rect_improve() as a sequence of launches ---------------------------------------------------------------------------
// rect_improve() is: first test; then five phases (finer precision; narrower; one side; the other side; finer precision again),
// each of five trial rectangles, returning as soon as a phase ends with log_nfa > 0.  The trials of a phase do not depend on
// each other's outcome (r changes cumulatively, rec only receives copies), and the two halves of a trial parallelise in
// opposite ways, so every phase is TWO launches over all rectangles that are still undecided:
//   k_lsd_nfa_count<PH>  16 lanes = (rectangle, trial): the pixel scan (for the finer-precision phases 16 lanes = rectangle, one
//                        scan for all five tolerances) -> (n, k) per trial;
//   k_lsd_nfa_setup<PH> / k_lsd_nfa_series<PH>  THREAD = (rectangle, trial): nfa(n, k, p) is scalar code - thousands of
//                        independent evaluations per launch instead of one per wave; the series are drawn from a shared
//                        counter, because their lengths differ widely;
//   k_lsd_nfa_select<PH> thread = rectangle: the reference's sequential `if (v > log_nfa)` selection over the trials, the
//                        chosen trial rebuilt by replaying its steps, accept / keep going / reject.
// State per rectangle: PSL_LSD_RECT_F64 doubles in `rects` (x1 y1 x2 y2 width theta dx dy | prec p log_nfa) and one byte in
// `keep`: 0 = rejected, 1 = accepted, 2 = undecided.  PH: -2 first test, -1 finer, 0 narrower, 1 / 2 one side, 3 finer again.
#define PSL_NFA_FIRST (-2)

__device__ __forceinline__ void lsdn_load(const double* __restrict__ r, LsdnRect* rec) {
    rec->x1 = r[0]; rec->y1 = r[1]; rec->x2 = r[2]; rec->y2 = r[3]; rec->width = r[4]; rec->theta = r[5]; rec->dx = r[6]; rec->dy = r[7];
    rec->prec = r[8]; rec->p = r[9];
}

#ifndef PSL_NFA_GL
#define PSL_NFA_GL 16   // lanes per pixel scan: four (rectangle, trial) scans per wave (8: nfa_count 48.7 ms per 12288 dense frames, 16: 39.9, 32: 40.2 - profiles/r03z_ab_nfa_gl.log)
#endif

#ifndef PSL_NFA_COUNT_WAVES
#define PSL_NFA_COUNT_WAVES 8   // register bound: 64 VGPRs (78 at 4): 39.9 -> 38.4 ms per 12288 dense frames (profiles/r03z_ab_nfa_grid.log)
#endif
template <int PH>
__global__ __launch_bounds__(256, PSL_NFA_COUNT_WAVES) void k_lsd_nfa_count(LineParams P, const float* __restrict__ angdeg, double* __restrict__ rects,
                                                          const int* __restrict__ nrect, const uint8_t* __restrict__ keep, int2* __restrict__ counts) {
    constexpr int TR = 1;                               // scans per rectangle: the five trials of a phase share ONE pass (lsdn_count_trials / lsdn_count<5>)
    constexpr int GPB = 256 / PSL_NFA_GL;               // scans in flight per workgroup
    __shared__ int s_geo[GPB * 45];                     // per scan group: five trial geometries (LsdnGeomI)
    const int frame = blockIdx.y, lane = threadIdx.x & (PSL_NFA_GL - 1);
    const int cnt = nrect[frame] < P.maxseg ? nrect[frame] : P.maxseg;
    const float* ang = angdeg + (size_t)frame * P.W * P.H;
    // every lane runs its own trip count (its group's items); nothing inside the loop crosses groups
    for (int item = (int)blockIdx.x * GPB + (int)(threadIdx.x / PSL_NFA_GL); item < cnt * TR; item += (int)gridDim.x * GPB) {
        const int idx = item / TR, j = item - idx * TR;
        const size_t o = (size_t)frame * P.maxseg + idx;
        if (PH != PSL_NFA_FIRST && keep[o] != 2) continue;
        double* rs = rects + o * PSL_LSD_RECT_F64;
        LsdnRect rec;
        lsdn_load(rs, &rec);
        if (PH == PSL_NFA_FIRST) {
            rec.prec = P.prec; rec.p = P.p;
            if (lane == 0) { rs[8] = rec.prec; rs[9] = rec.p; }
        }
        LsdnGeom G;
        int2* out = counts + o * 5;
        if (PH == -1 || PH == 3) {       // five tolerances p / 2^(j+1), one geometry, ONE pass over the pixels
            if (PH == 3 && !(PSL_DSUB(rec.width, 0.5) >= 0.5)) {   // the guard holds for all five trials or for none
                if (lane < 5) out[lane] = make_int2(-1, 0);
                continue;
            }
            double pr[5], pp = rec.p;
#pragma unroll
            for (int t = 0; t < 5; ++t) { pp = pp / 2; pr[t] = PSL_DMUL(pp, PSL_PI); }
            int tot, kk[5];
            lsdn_geom(rec, P.H, &G);
            lsdn_count<5, PSL_NFA_GL>(ang, P.W, G, rec.theta, pr, lane, &tot, kk);
            if (lane == 0) {
#pragma unroll
                for (int t = 0; t < 5; ++t) out[t] = make_int2(tot, kk[t]);
            }
        } else if (PH >= 0) {            // trial t = t + 1 cumulative steps, each under the width guard (it only ever turns false)
            int* geo = &s_geo[(threadIdx.x / PSL_NFA_GL) * 45];
            if (lane < 5) {              // lane t builds trial t's geometry
                LsdnRect r = rec;
                bool valid = true;
                for (int t = 0; t <= lane; ++t) {
                    if (!(PSL_DSUB(r.width, 0.5) >= 0.5)) { valid = false; break; }
                    lsdn_shrink(r, PH);
                }
                int* g = geo + 9 * lane;
                g[0] = 0; g[1] = -2;
                if (valid) {
                    lsdn_geom(r, P.H, &G);
                    g[1] = -1;               // no in-image row: a valid, empty scan (0, 0)
                    if (G.yb >= G.ya) {
                        g[0] = G.ya; g[1] = G.yb; g[2] = G.lefty; g[3] = G.righty; g[4] = (int)G.minx;
                        g[5] = (int)G.fl; g[6] = (int)G.sl; g[7] = (int)G.fr; g[8] = (int)G.sr;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            int nn[5], kk[5];
            lsdn_count_trials<PSL_NFA_GL>(ang, P.W, (lds_cint*)geo, rec.theta, rec.prec, lane, nn, kk);
            if (lane == 0) {
#pragma unroll
                for (int t = 0; t < 5; ++t) out[t] = geo[9 * t + 1] == -2 ? make_int2(-1, 0) : make_int2(nn[t], kk[t]);
            }
            __builtin_amdgcn_wave_barrier();   // the group's geometries are rewritten by its next item
        } else {                         // the first test: one scan
            int nn = -1, kk = 0;
            lsdn_geom(rec, P.H, &G);
            lsdn_count<1, PSL_NFA_GL>(ang, P.W, G, rec.theta, &rec.prec, lane, &nn, &kk);
            if (lane == 0) out[j] = make_int2(nn, kk);
        }
    }
}

// nfa() for every (rectangle, trial) of a frame, THREAD = evaluation, in two launches because its two halves behave differently:
//   k_lsd_nfa_setup   everything before the binomial tail - the trivial cases, log_gamma from the table, the first term (an
//                     exp) - uniform work, one item per thread; leaves either the value or the series' start (term, p_term);
//                     and lists the frame's series by predicted length;
//   k_lsd_nfa_series  the tail itself, 1 .. thousands of iterations per evaluation (mean ~40, heavy tail): a wave that mixes
//                     lengths waits for its slowest lane, so a workgroup sums the series of ONE length class of several frames
//                     (see the kernel).  -log10(sum) - logNT is left to k_lsd_nfa_select.
// sstate[item] = (term, p_term), term == 0: no series (the value is in vals[item]; -inf for a trial the width guard excludes);
// after the series: sstate[item].x = the binomial tail.
// One binomial tail to be summed (32 B).  `stop`: rect_improve only ever asks of a trial's value v whether v > log_nfa, and log_nfa is
// at least the value the rectangle carries into the phase; v = -log10(tail) - logNT can only fall while terms (all positive) are added,
// so once the partial tail reaches `stop` = 10^(-log_nfa_in - logNT) (1 + 1e-6) the trial has lost whatever the remaining terms are,
// and the series ends there (k_lsd_nfa_series stores +inf: -log10 gives -inf, never selected).  The margin 1e-6 (4e-7 in v) covers
// the last-ulp non-monotonicity of the restated log10 (~2e-14) many times over; a trial that could still win is summed to the
// reference's own truncation point, bit for bit.  The first test (its value BECOMES log_nfa) has stop = +inf.  `stop` is kept as the next
// float ABOVE the threshold (never below FLT_MIN): a larger threshold only stops later.  `pad`: the length class, 0xffff = no series.
struct LsdnSeries { double term, p_term; int n, i; float stop; uint16_t slot, pad; };
static_assert(sizeof(LsdnSeries) == 32, "LsdnSeries is 32 bytes");
#define PSL_NFA_NCLS 12   // length classes of the series: class c holds predicted lengths in [2^c, 2^(c+1)) (c = 11: all longer ones)
#ifndef PSL_NFA_FG
#define PSL_NFA_FG 16     // frames whose series of one class are summed by one workgroup (8: nfa_eval 29.0 ms per 12288 dense frames, 16: 27.3, 32: 42.6, 64: 50.2 - profiles/r03z_ab_nfa_fg.log)
#endif

// predicted number of terms of the tail of B(n, p) from k + 1: the climb to the mode n p (if k is below it) plus a few standard
// deviations, never more than n - k.  Only a scheduling hint: series of similar predicted length share a wave.
__device__ __forceinline__ int lsdn_series_class(int n, int k, double p) {
    const float np = (float)n * (float)p;
    float len = fmaxf(0.0f, np - (float)k) + 4.0f * __builtin_sqrtf(np * (1.0f - (float)p)) + 4.0f;
    len = fminf(len, (float)(n - k));
    const int c = 31 - __clz((int)len | 1);
    return c < PSL_NFA_NCLS - 1 ? c : PSL_NFA_NCLS - 1;
}

// workgroup = frame (all items of the frame, so that its series can be bucketed by predicted length in LDS without global
// atomics).  list: [F][maxseg * 5] entries, class-major; coff: [F][PSL_NFA_NCLS + 1] offsets of the classes
template <int PH>
__global__ __launch_bounds__(256) void k_lsd_nfa_setup(LineParams P, LsdnTables T, const double* __restrict__ rects, const int* __restrict__ nrect,
                                                       const uint8_t* __restrict__ keep, const int2* __restrict__ counts, double* __restrict__ vals,
                                                       double2* __restrict__ sstate, LsdnSeries* __restrict__ tmp, LsdnSeries* __restrict__ list,
                                                       int* __restrict__ coff) {
    __shared__ int s_hist[PSL_NFA_NCLS], s_cur[PSL_NFA_NCLS];
    const int frame = blockIdx.x, tid = threadIdx.x;
    const int cnt = nrect[frame] < P.maxseg ? nrect[frame] : P.maxseg;
    constexpr int TR = PH == PSL_NFA_FIRST ? 1 : 5;
    const double log_nt = T.log_nt;
    const int lcap = P.maxseg * 5;
    LsdnSeries* TMP = tmp + (size_t)frame * lcap;
    LsdnSeries* L = list + (size_t)frame * lcap;
    if (tid < PSL_NFA_NCLS) s_hist[tid] = 0;
    __syncthreads();
    for (int item = tid; item < cnt * TR; item += 256) {
        const int idx = item / TR, j = item - idx * TR;
        const size_t o = (size_t)frame * P.maxseg + idx;
        LsdnSeries e = {};
        e.pad = 0xffffu;   // no series
        if (PH == PSL_NFA_FIRST || keep[o] == 2) {
            const size_t slot = o * 5 + j;
            double v = 0, term = 0, p_term = 0;
            const int2 c = counts[slot];
            if (c.x < 0) v = -__builtin_inf();
            else {
                double p = rects[o * PSL_LSD_RECT_F64 + 9];
                if (PH == -1 || PH == 3) for (int t = 0; t <= j; ++t) p = p / 2;
                const int n = c.x, k = c.y;
                // p = p0 / 2^jp: the table row, or -1 (a probability rect_improve cannot produce: evaluated directly)
                int jp = (int)((__double_as_longlong(T.p0) >> 52) & 0x7ff) - (int)((__double_as_longlong(p) >> 52) & 0x7ff);
                if (jp < 0 || jp >= PSL_NFA_NP || __longlong_as_double(__double_as_longlong(T.p0) - ((long long)jp << 52)) != p) jp = -1;
                if (n == 0 || k == 0) v = -log_nt;
                else if (n == k) v = PSL_DSUB(-log_nt, PSL_DMUL((double)n, jp >= 0 ? T.logs[2 * PSL_NFA_NP + jp] : psl_log10(p)));
                else {
                    p_term = p / PSL_DSUB(1.0, p);
                    double log1term = PSL_DSUB(PSL_DSUB(lsdn_lg(T, n + 1), lsdn_lg(T, k + 1)), lsdn_lg(T, n - k + 1));
                    log1term = PSL_DADD(PSL_DADD(log1term, PSL_DMUL((double)k, jp >= 0 ? T.logs[jp] : psl_log(p))),
                                        PSL_DMUL((double)(n - k), jp >= 0 ? T.logs[PSL_NFA_NP + jp] : psl_log(PSL_DSUB(1.0, p))));
                    term = psl_exp(log1term);
                    if (lsdn_double_equal0(term)) {
                        v = (double)k > PSL_DMUL((double)n, p) ? PSL_DSUB(-log1term / 2.30258509299404568402, log_nt) : -log_nt;
                        term = 0;
                    } else {   // (k + 1 <= n here: the series has at least one term)
                        e.term = term; e.p_term = p_term; e.n = n; e.i = k + 1; e.slot = (uint16_t)(slot - (size_t)frame * lcap);   // < maxseg * 5 <= 65535
                        e.stop = __builtin_inf();
                        if (PH != PSL_NFA_FIRST) {
                            const double lin = rects[o * PSL_LSD_RECT_F64 + 10];   // log_nfa the rectangle brings into this phase
                            const double ex = (-lin - log_nt) * 2.30258509299404568402;
                            if (ex < 0.0 && ex > -700.0) {   // tail <= 1: a threshold >= 1 never triggers
                                const double th = psl_exp(ex) * (1.0 + 1e-6);
                                float tf = (float)th;
                                if ((double)tf < th) tf = __uint_as_float(__float_as_uint(tf) + 1u);   // round up (th > 0)
                                e.stop = tf < 1.17549435e-38f ? 1.17549435e-38f : tf;
                            }
                        }
                        e.pad = (uint16_t)lsdn_series_class(n, k, p);
                        atomicAdd(&s_hist[e.pad], 1);
                    }
                }
            }
            vals[slot] = v;
            sstate[slot] = make_double2(term, p_term);
        }
        TMP[item] = e;
    }
    __syncthreads();
    if (tid == 0) {   // classes in DESCENDING length order: the longest series are listed (and started) first
        int run = 0;
        for (int c = PSL_NFA_NCLS - 1; c >= 0; --c) { s_cur[c] = run; coff[frame * (PSL_NFA_NCLS + 1) + (PSL_NFA_NCLS - 1 - c)] = run; run += s_hist[c]; }
        coff[frame * (PSL_NFA_NCLS + 1) + PSL_NFA_NCLS] = run;
    }
    __syncthreads();
    for (int item = tid; item < cnt * TR; item += 256) {
        const LsdnSeries e = TMP[item];
        if (e.pad != 0xffffu) L[atomicAdd(&s_cur[e.pad], 1)] = e;
    }
}

// The binomial tails, one per THREAD.  Workgroup = (length class, PSL_NFA_FG frames): the series of that class of those frames
// are dealt to its threads round-robin, so the lanes of a wave sum series of similar length (a wave that mixes lengths waits
// for its longest: measured 7 % lane use with one workgroup per frame).  A thread works through its entries one after the
// other and has the next one loaded before it needs it; a lane whose series has stopped idles until an eighth of its wave is
// idle - then all such lanes store their sums and start their next entries together.  The hot loop between two such points is
// the recurrence, the division and the pre-test.
#ifndef PSL_NFA_SERIES_WAVES
#define PSL_NFA_SERIES_WAVES 6   // waves per SIMD the register bound allows: the dependent f64 chains leave the vector unit half idle, so more waves help until they spill -
                                 // without a bound 122 VGPRs (4 waves): nfa_eval 28.8 ms per 12288 dense frames, 5: 27.8, 6 (78 VGPRs): 26.8, 8 (64 + scratch): 41.2 (profiles/r03z_ab_series_waves.log)
#endif
template <int PH>
__global__ __launch_bounds__(256, PSL_NFA_SERIES_WAVES) void k_lsd_nfa_series(LineParams P, LsdnTables T, int nframes, const LsdnSeries* __restrict__ list,
                                                        const int* __restrict__ coff, double2* __restrict__ sstate) {
    __shared__ int s_begin[PSL_NFA_FG], s_pref[PSL_NFA_FG + 1];
    const int cls = blockIdx.x, f0 = (int)blockIdx.y * PSL_NFA_FG, tid = threadIdx.x;
    const int lcap = P.maxseg * 5;
    if (tid == 0) {
        int run = 0;
        for (int f = 0; f < PSL_NFA_FG; ++f) {
            int b = 0, c = 0;
            if (f0 + f < nframes) { b = coff[(f0 + f) * (PSL_NFA_NCLS + 1) + cls]; c = coff[(f0 + f) * (PSL_NFA_NCLS + 1) + cls + 1] - b; }
            s_begin[f] = b; s_pref[f] = run; run += c;
        }
        s_pref[PSL_NFA_FG] = run;
    }
    __syncthreads();
    const int total = s_pref[PSL_NFA_FG];
    if (total == 0) return;
    // virtual index t in the concatenation of the frames' class segments -> the entry and its frame
    auto locate = [&](int t, int* fr) {
        int f = 0;
#pragma unroll
        for (int g = 1; g < PSL_NFA_FG; ++g) f += t >= s_pref[g] ? 1 : 0;
        *fr = f;
        return list + (size_t)(f0 + f) * lcap + s_begin[f] + (t - s_pref[f]);
    };
    const double log_nt = T.log_nt;
    bool running = false, done = false;
    int mine = tid;
    bool have_next = mine < total;
    LsdnSeries nxt = {};
    int nxt_f = 0;
    if (have_next) nxt = *locate(mine, &nxt_f);
    int n = 0, i = 0;
    double2* out = nullptr;
    double term = 0, bin_tail = 0, p_term = 0, stop_at = __builtin_inf();
    for (;;) {
        if (__popcll(__ballot(!running)) >= 8) {
            if (done) { out->x = bin_tail; done = false; }
            if (!running && have_next) {
                n = nxt.n; i = nxt.i; term = nxt.term; bin_tail = nxt.term; p_term = nxt.p_term; stop_at = (double)nxt.stop;
                out = sstate + (size_t)(f0 + nxt_f) * lcap + nxt.slot;
                running = true;
                mine += 256;
                have_next = mine < total;
                if (have_next) nxt = *locate(mine, &nxt_f);   // in flight while this series runs
            }
            if (!__ballot(running)) break;
        }
        if (running && i + 7 <= n && 2 * (i + 7) <= n + 1) {
            // eight terms at once while bin_term >= 1 (i <= (n + 1) / 2: the reference does not test there - 98 % of all terms): the
            // divisions are independent of the running product, only the multiply-add chain is serial
            double bt[8];
            if (i + 7 < PSL_RATIO_BMAX && n < PSL_RATIO_AMAX) {   // the quotients from the reciprocal table: 3 operations each instead of a division sequence
                double rv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) rv[u] = T.inv[i + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) bt[u] = psl_ratio_inv((double)(n - i - u + 1), (double)(i + u), rv[u]);
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) bt[u] = (double)(n - i - u + 1) / (double)(i + u);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                term = PSL_DMUL(term, PSL_DMUL(bt[u], p_term));
                bin_tail = PSL_DADD(bin_tail, term);
            }
            i += 8;
            if (i > n) { running = false; done = true; }
            if (bin_tail >= stop_at) { bin_tail = __builtin_inf(); running = false; done = true; }   // the trial has lost (LsdnSeries)
        } else if (running) {   // one term of the binomial tail
            const double bin_term = (double)(n - i + 1) / (double)i;
            const double mult_term = PSL_DMUL(bin_term, p_term);
            term = PSL_DMUL(term, mult_term);
            bin_tail = PSL_DADD(bin_tail, term);
            bool stop = false;
            if (bin_term < 1) {
                const int q = n - i + 1;
                int st = lsdn_tail_test_fast(term, bin_tail, mult_term, q, log_nt);
                if (st < 0) {
                    const double pw = q == 1 ? mult_term : psl_pow_pos(mult_term, (double)q);  // pow(x, 1) is exact in any libm
                    const double err = PSL_DMUL(term, PSL_DSUB(PSL_DSUB(1.0, pw) / PSL_DSUB(1.0, mult_term), 1.0));
                    st = err < PSL_DMUL(PSL_DMUL(0.1, fabs(PSL_DSUB(-psl_log10(bin_tail), log_nt))), bin_tail) ? 1 : 0;
                }
                stop = st != 0;
            }
            ++i;
            if (stop || i > n) { running = false; done = true; }
            if (bin_tail >= stop_at) { bin_tail = __builtin_inf(); running = false; done = true; }
        }
    }
}

// thread = rectangle: the reference's selection over the trials of the phase, in trial order - `if (v > log_nfa) { log_nfa = v;
// rec = r; }` - the chosen trial rebuilt by replaying its steps, and the verdict if there is one
template <int PH>
__global__ __launch_bounds__(256) void k_lsd_nfa_select(LineParams P, double log_nt, double* __restrict__ rects, const int* __restrict__ nrect,
                                                        uint8_t* __restrict__ keep, const double* __restrict__ vals, const double2* __restrict__ sstate,
                                                        float* __restrict__ segtmp) {
    const int frame = blockIdx.y;
    const int cnt = nrect[frame] < P.maxseg ? nrect[frame] : P.maxseg;
    for (int idx = (int)(blockIdx.x * 256 + threadIdx.x); idx < cnt; idx += (int)gridDim.x * 256) {
        const size_t o = (size_t)frame * P.maxseg + idx;
        if (PH != PSL_NFA_FIRST && keep[o] != 2) continue;
        double* rs = rects + o * PSL_LSD_RECT_F64;
        LsdnRect rec;
        lsdn_load(rs, &rec);
        double logn;
        int best = -1;
        // nfa() of trial t: the value k_lsd_nfa_setup left, or -log10(binomial tail) - logNT of its series
        auto value = [&](int t) {
            const double bt = sstate[o * 5 + t].x;
            if (bt == __builtin_inf()) return -__builtin_inf();   // a series that ended at its `stop`: the trial cannot be selected
            return bt != 0 ? PSL_DSUB(-psl_log10(bt), log_nt) : vals[o * 5 + t];
        };
        if (PH == PSL_NFA_FIRST) logn = value(0);
        else {
            logn = rs[10];
#pragma unroll 1
            for (int t = 0; t < 5; ++t) {
                const double vt = value(t);
                if (vt > logn) { logn = vt; best = t; }
            }
        }
        if (best >= 0) {
            if (PH == -1 || PH == 3) {
                double pp = rec.p;
                for (int t = 0; t <= best; ++t) pp = pp / 2;
                rec.p = pp; rec.prec = PSL_DMUL(pp, PSL_PI);
                rs[8] = rec.prec; rs[9] = rec.p;
            } else if (PH >= 0) {
                for (int t = 0; t <= best; ++t) lsdn_shrink(rec, PH);
                rs[0] = rec.x1; rs[1] = rec.y1; rs[2] = rec.x2; rs[3] = rec.y2; rs[4] = rec.width;
            }
        }
        rs[10] = logn;
        if (logn > 0) {   // LOG_EPS = 0: accepted, with the rectangle as it stands now
            keep[o] = 1;
            psl_lsd_store_segment(P, rec.x1, rec.y1, rec.x2, rec.y2, segtmp + 4 * o);
        } else keep[o] = PH == 3 ? 0 : 2;
    }
}

// workgroup = frame: the accepted segments in seed order
__global__ __launch_bounds__(256) void k_lsd_emit(LineParams P, const int* __restrict__ nrect, const float* __restrict__ segtmp,
                                                  const uint8_t* __restrict__ keep, float* __restrict__ seg, int* __restrict__ nseg) {
    __shared__ int s_w[4];
    const int frame = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cnt = nrect[frame] < P.maxseg ? nrect[frame] : P.maxseg;
    const size_t o = (size_t)frame * P.maxseg;
    int running = 0;
    for (int base = 0; base < cnt; base += 256) {
        const int i = base + tid;
        const bool f = i < cnt && keep[o + i] != 0;
        const unsigned long long m = __ballot(f);
        if (lane == 0) s_w[wave] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { before += q < wave ? s_w[q] : 0; total += s_w[q]; }
        if (f) {
            const int pos = running + before + __popcll(m & ((1ull << lane) - 1ull));
            const float4 v = *reinterpret_cast<const float4*>(segtmp + 4 * (o + i));
            *reinterpret_cast<float4*>(seg + 4 * (o + pos)) = v;
        }
        running += total;
        __syncthreads();
    }
    if (tid == 0) nseg[frame] = running;
}

#endif
