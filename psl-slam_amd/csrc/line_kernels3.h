// LSD_REFINE_ADV for the LSD of the line front-end (gfx950, wave64). Product code.
// Reference behaviour reproduced: the `doRefine >= LSD_REFINE_ADV` step of OpenCV 3.x lsd.cpp flsd() -
// rect_improve(), rect_nfa(), nfa(), log_gamma() - which the stock contrib LSDDetector the reference links
// (add_src/LineExtractor.cpp:336-337, CMakeLists.txt:96) selects; lsd.cpp is not in the reference tree, the
// nfa() / log_gamma() arithmetic has a twin there: Thirdparty/line_descriptor/src/ED_Lib/NFA.cpp:106-240.
//
// Decomposition.  The validation reads the level-line angles and the rectangle, never the `used` map, and a
// rejected rectangle releases nothing: it does not feed back into the region growing.  k_lsd_grow3 therefore only
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

__device__ __forceinline__ int lsdn_wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ int lsdn_wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int u = __shfl_xor(v, o); v = u > v ? u : v; }
    return v;
}

// total_pts and, for NP tolerances, alg_pts of rect_nfa().  The scan is a few hundred pixels whose angles sit in L2 / L1: what
// it costs is memory latency, so the pixels are enumerated as slots (row, offset < widest span) dealt to the lanes round-robin
// and every lane has four loads in flight - not one dependent load per pixel in a per-row or per-column loop.
template <int NP>
__device__ __forceinline__ void lsdn_count(const float* __restrict__ ang, int W, const LsdnGeom& G, double theta, const double* prec, int lane, int* total, int* alg) {
    int al[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) al[j] = 0;
    const int nrows = G.yb - G.ya + 1;
    int tot = 0, wmax = 0;
    for (int y = G.ya + lane; y <= G.yb; y += 64) {  // lane = row: pixels and widest span
        int xa, xb;
        lsdn_row_span(G, y, W, &xa, &xb);
        const int c = xb - xa + 1;
        tot += c > 0 ? c : 0;
        wmax = c > wmax ? c : wmax;
    }
    *total = lsdn_wave_sum(tot);
    wmax = lsdn_wave_max_i(wmax);
    if (nrows > 0 && wmax > 0) {
        const int nslots = nrows * wmax, row_step = 64 / wmax, dx_step = 64 - row_step * wmax;
        int row = lane / wmax, dx = lane - row * wmax;
        for (int s = lane; s < nslots; s += 256) {
            float a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = PSL_LSD_NOTDEF;
                if (s + 64 * u < nslots) {
                    int xa, xb;
                    const int y = G.ya + row;
                    lsdn_row_span(G, y, W, &xa, &xb);
                    if (dx <= xb - xa) a[u] = ang[y * W + xa + dx];
                }
                dx += dx_step; row += row_step;
                if (dx >= wmax) { dx -= wmax; ++row; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (a[u] != PSL_LSD_NOTDEF) {
                    const double f = lsdg_fold(PSL_DMUL((double)a[u], PSL_DEG2RAD), theta);
#pragma unroll
                    for (int j = 0; j < NP; ++j) al[j] += f <= prec[j] ? 1 : 0;
                }
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) alg[j] = lsdn_wave_sum(al[j]);
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
//   k_lsd_nfa_count<PH>  wave = (rectangle, trial): the pixel scan (for the finer-precision phases wave = rectangle, one scan
//                        for all five tolerances) -> (n, k) per trial;
//   k_lsd_nfa_eval<PH>   THREAD = (rectangle, trial): nfa(n, k, p) is scalar code - thousands of independent evaluations
//                        per launch instead of one per wave - drawn from a shared counter, because their lengths differ widely;
//   k_lsd_nfa_select<PH> thread = rectangle: the reference's sequential `if (v > log_nfa)` selection over the trials, the
//                        chosen trial rebuilt by replaying its steps, accept / keep going / reject.
// State per rectangle: PSL_LSD_RECT_F64 doubles in `rects` (x1 y1 x2 y2 width theta dx dy | prec p log_nfa) and one byte in
// `keep`: 0 = rejected, 1 = accepted, 2 = undecided.  PH: -2 first test, -1 finer, 0 narrower, 1 / 2 one side, 3 finer again.
#define PSL_NFA_FIRST (-2)

__device__ __forceinline__ void lsdn_load(const double* __restrict__ r, LsdnRect* rec) {
    rec->x1 = r[0]; rec->y1 = r[1]; rec->x2 = r[2]; rec->y2 = r[3]; rec->width = r[4]; rec->theta = r[5]; rec->dx = r[6]; rec->dy = r[7];
    rec->prec = r[8]; rec->p = r[9];
}

template <int PH>
__global__ __launch_bounds__(256, 4) void k_lsd_nfa_count(LineParams P, const float* __restrict__ angdeg, double* __restrict__ rects,
                                                          const int* __restrict__ nrect, const uint8_t* __restrict__ keep, int2* __restrict__ counts) {
    constexpr int TR = (PH >= 0 && PH <= 2) ? 5 : 1;   // waves per rectangle
    const int frame = blockIdx.y, lane = threadIdx.x & 63;
    const int cnt = nrect[frame] < P.maxseg ? nrect[frame] : P.maxseg;
    const float* ang = angdeg + (size_t)frame * P.W * P.H;
    for (int item = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); item < cnt * TR; item += (int)gridDim.x * 4) {
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
            lsdn_count<5>(ang, P.W, G, rec.theta, pr, lane, &tot, kk);
            if (lane == 0) {
#pragma unroll
                for (int t = 0; t < 5; ++t) out[t] = make_int2(tot, kk[t]);
            }
        } else {
            bool valid = true;
            if (PH >= 0) {               // trial j = j + 1 cumulative steps, each under the width guard (it only ever turns false)
                for (int t = 0; t <= j; ++t) {
                    if (!(PSL_DSUB(rec.width, 0.5) >= 0.5)) { valid = false; break; }
                    lsdn_shrink(rec, PH);
                }
            }
            int nn = -1, kk = 0;
            if (valid) {
                lsdn_geom(rec, P.H, &G);
                lsdn_count<1>(ang, P.W, G, rec.theta, &rec.prec, lane, &nn, &kk);
            }
            if (lane == 0) out[j] = make_int2(nn, kk);
        }
    }
}

// nfa() for every (rectangle, trial) of a frame: THREAD = evaluation, scheduled dynamically.  The binomial tail of nfa() runs
// for 1 .. ~900 iterations (mean ~40, heavy tail), so a wave that gives every lane ONE evaluation waits for its slowest lane
// (measured: 10x the mean).  Here the lanes of a workgroup draw evaluations from a shared counter: a lane whose series has
// stopped idles at most 15 iterations, until the next refill point, where all such lanes finish (log10, store), draw the next
// item and set it up (three table loads, an exp) together - the hot loop between refill points is only the recurrence, the
// division and the f32 pre-test.  Results overwrite the (n, k) pair of the trial with the value (a double, -inf for a trial
// the width guard excludes); k_lsd_nfa_select reads them.
template <int PH>
__global__ __launch_bounds__(256) void k_lsd_nfa_eval(LineParams P, LsdnTables T, const double* __restrict__ rects, const int* __restrict__ nrect,
                                                      const uint8_t* __restrict__ keep, int2* __restrict__ counts) {
    __shared__ int s_next;
    const int frame = blockIdx.y;
    const int cnt = nrect[frame] < P.maxseg ? nrect[frame] : P.maxseg;
    constexpr int TR = PH == PSL_NFA_FIRST ? 1 : 5;
    const int nitems = cnt * TR;
    const int per = (nitems + (int)gridDim.x - 1) / (int)gridDim.x;
    const int begin = (int)blockIdx.x * per, end = min(begin + per, nitems);
    if (begin >= end) return;
    if (threadIdx.x == 0) s_next = begin;
    __syncthreads();
    const double log_nt = T.log_nt;
    double* vals = reinterpret_cast<double*>(counts);
    // lane state
    bool running = false, done = false;   // done: the series has stopped, the value is still to be written
    int n = 0, i = 0;
    size_t slot = 0;
    double term = 0, bin_tail = 0, p_term = 0;
    for (int iter = 0;; ++iter) {
        if ((iter & 15) == 0) {   // refill point
            if (done) { vals[slot] = PSL_DSUB(-psl_log10(bin_tail), log_nt); done = false; }
            for (int tries = 0; tries < 1 && !running; ++tries) {
                const int item = atomicAdd(&s_next, 1);
                if (item >= end) break;
                const int idx = item / TR, j = item - idx * TR;
                const size_t o = (size_t)frame * P.maxseg + idx;
                if (PH != PSL_NFA_FIRST && keep[o] != 2) continue;
                slot = o * 5 + j;
                const int2 c = counts[slot];
                if (c.x < 0) { vals[slot] = -__builtin_inf(); continue; }
                double p = rects[o * PSL_LSD_RECT_F64 + 9];
                if (PH == -1 || PH == 3) for (int t = 0; t <= j; ++t) p = p / 2;
                // nfa(): everything before the series
                n = c.x;
                const int k = c.y;
                if (n == 0 || k == 0) { vals[slot] = -log_nt; continue; }
                int jp = (int)((__double_as_longlong(T.p0) >> 52) & 0x7ff) - (int)((__double_as_longlong(p) >> 52) & 0x7ff);
                if (jp < 0 || jp >= PSL_NFA_NP || __longlong_as_double(__double_as_longlong(T.p0) - ((long long)jp << 52)) != p) jp = -1;
                if (n == k) { vals[slot] = PSL_DSUB(-log_nt, PSL_DMUL((double)n, jp >= 0 ? T.logs[2 * PSL_NFA_NP + jp] : psl_log10(p))); continue; }
                p_term = p / PSL_DSUB(1.0, p);
                double log1term = PSL_DSUB(PSL_DSUB(lsdn_lg(T, n + 1), lsdn_lg(T, k + 1)), lsdn_lg(T, n - k + 1));
                log1term = PSL_DADD(PSL_DADD(log1term, PSL_DMUL((double)k, jp >= 0 ? T.logs[jp] : psl_log(p))),
                                    PSL_DMUL((double)(n - k), jp >= 0 ? T.logs[PSL_NFA_NP + jp] : psl_log(PSL_DSUB(1.0, p))));
                term = psl_exp(log1term);
                if (lsdn_double_equal0(term)) {
                    vals[slot] = (double)k > PSL_DMUL((double)n, p) ? PSL_DSUB(-log1term / 2.30258509299404568402, log_nt) : -log_nt;
                    continue;
                }
                bin_tail = term;
                i = k + 1;
                if (i > n) { vals[slot] = PSL_DSUB(-psl_log10(bin_tail), log_nt); continue; }
                running = true;
            }
            if (!__syncthreads_or(running ? 1 : 0)) break;   // nobody in the workgroup has work left (the counter is exhausted)
        }
        if (running) {   // one term of the binomial tail
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
        }
    }
}

// thread = rectangle: the reference's selection over the trials of the phase, in trial order - `if (v > log_nfa) { log_nfa = v;
// rec = r; }` - the chosen trial rebuilt by replaying its steps, and the verdict if there is one
template <int PH>
__global__ __launch_bounds__(256) void k_lsd_nfa_select(LineParams P, double* __restrict__ rects, const int* __restrict__ nrect,
                                                        uint8_t* __restrict__ keep, const int2* __restrict__ counts, float* __restrict__ segtmp) {
    const int frame = blockIdx.y;
    const int cnt = nrect[frame] < P.maxseg ? nrect[frame] : P.maxseg;
    const double* vals = reinterpret_cast<const double*>(counts);
    for (int idx = (int)(blockIdx.x * 256 + threadIdx.x); idx < cnt; idx += (int)gridDim.x * 256) {
        const size_t o = (size_t)frame * P.maxseg + idx;
        if (PH != PSL_NFA_FIRST && keep[o] != 2) continue;
        double* rs = rects + o * PSL_LSD_RECT_F64;
        LsdnRect rec;
        lsdn_load(rs, &rec);
        double logn;
        int best = -1;
        if (PH == PSL_NFA_FIRST) logn = vals[o * 5];
        else {
            logn = rs[10];
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const double vt = vals[o * 5 + t];
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
