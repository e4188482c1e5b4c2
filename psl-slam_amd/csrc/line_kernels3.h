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
//   k_lsd_nfa   wave = rectangle: pixel counts by the lanes (rows or columns of the scan, whichever is longer),
//               the five trial rectangles of one rect_improve phase evaluated together (their geometry does
//               not depend on each other's outcome; the two "finer precision" phases share ONE pixel pass for
//               all five tolerances), nfa() of trial j on lane j;
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

// The truncation test of the binomial tail, `err < tolerance * |-log10(bin_tail) - logNT| * bin_tail`, is what an iteration of
// nfa() costs (a pow and a log10 in double: ~500 instructions beside ~40 for the recurrence).  It only decides WHERE the
// series stops, so it is first evaluated with the hardware's f32 log2 / exp2 and explicit error margins: the estimate of err is
// within 1e-3 relative (m - m^q and 1 - m do not cancel for m < 1/7) and the logarithm within 1e-5 absolute; a comparison that
// these margins decide is the comparison the exact arithmetic makes, anything closer falls back to the exact test.
// returns 1 = stop, 0 = go on, -1 = undecided
__device__ __forceinline__ int lsdn_tail_test_fast(double term, double bin_tail, double mult_term, int q, double log_nt) {
    const float m = (float)mult_term;
    const float pw = q == 1 ? m : (q == 2 ? m * m : __builtin_amdgcn_exp2f((float)q * __builtin_amdgcn_logf(m)));
    const float A = (m - pw) / (1.0f - m);
    const double errE = term * (double)A;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(bin_tail);
    const int e = (int)((bits >> 52) & 0x7ff) - 1023;
    if (e <= -1022 || e >= 1024) return -1;  // subnormal / non-finite: exact path
    const float fm = __longlong_as_double((long long)((bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull));
    const double l10 = ((double)e + (double)__builtin_amdgcn_logf(fm)) * 0.30102999566398120;
    const double Lv = fabs(-l10 - log_nt);
    const double lo = Lv - 1e-5, hi = Lv + 1e-5;
    if (errE * 1.001 < 0.1 * (lo > 0 ? lo : 0.0) * bin_tail) return 1;
    if (errE * 0.999 >= 0.1 * hi * bin_tail) return 0;
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

// nfa() of up to five (n, k, p) triples, trial j on lane j (lanes >= ntr idle); lane j keeps its value
__device__ __forceinline__ double lsdn_nfa_lanes(const LsdnTables& T, int nj, int kj, double pj, int ntr, int lane) {
    double mine = 0;
    if (lane < ntr) mine = lsdn_nfa(T, nj, kj, pj);
    return mine;
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

// rect_improve(): returns log_nfa, *rec = the improved rectangle.  All lanes hold the same values.  The five trial rectangles
// of a phase do not depend on each other's outcome (r is changed cumulatively, rec only receives copies), so their pixel counts
// are taken one after the other (wave-parallel each), their nfa() values together (trial j on lane j), and the chosen trial is
// rebuilt by replaying its steps - nothing but (n, k) per trial is kept, in LDS (sc: 10 ints of this wave).
__device__ __forceinline__ double lsdn_rect_improve(const float* __restrict__ ang, int W, int H, const LsdnTables& T, LsdnRect* rec, int lane, int* sc) {
    LsdnGeom G;
    double log_nfa;
    {
        int n1, k1;
        lsdn_geom(*rec, H, &G);
        lsdn_count<1>(ang, W, G, rec->theta, &rec->prec, lane, &n1, &k1);
        log_nfa = lsdw_lane_f64(lsdn_nfa_lanes(T, n1, k1, rec->p, 1, lane), 0);
    }
    if (log_nfa > 0) return log_nfa;
#pragma unroll 1
    for (int phase = -1; phase <= 3; ++phase) {
        int ntr = 0;
        double pj = rec->p;   // this lane's trial probability
        if (phase == -1 || phase == 3) {
            // finer precision: one geometry, five tolerances p / 2^(j+1), ONE pass over the pixels
            if (phase == 3 && !(PSL_DSUB(rec->width, 0.5) >= 0.5)) break;   // the guard holds for all five trials or for none
            double pr[5], pp = rec->p;
#pragma unroll
            for (int j = 0; j < 5; ++j) { pp = pp / 2; pr[j] = PSL_DMUL(pp, PSL_PI); if (lane == j) pj = pp; }
            int tot, kk[5];
            lsdn_geom(*rec, H, &G);
            lsdn_count<5>(ang, W, G, rec->theta, pr, lane, &tot, kk);
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < 5; ++j) { sc[j] = tot; sc[5 + j] = kk[j]; }
            }
            ntr = 5;
        } else {
            LsdnRect r = *rec;
#pragma unroll 1
            for (int j = 0; j < 5; ++j) {
                if (!(PSL_DSUB(r.width, 0.5) >= 0.5)) break;   // the guard only ever turns false: the width shrinks monotonically
                lsdn_shrink(r, phase);
                int nn, kk;
                lsdn_geom(r, H, &G);
                lsdn_count<1>(ang, W, G, r.theta, &r.prec, lane, &nn, &kk);
                if (lane == 0) { sc[j] = nn; sc[5 + j] = kk; }
                ntr = j + 1;
            }
        }
        if (ntr) {
            __builtin_amdgcn_wave_barrier();
            const int jl = lane < 5 ? lane : 0;
            const double mine = lsdn_nfa_lanes(T, sc[jl], sc[5 + jl], pj, ntr, lane);
            __builtin_amdgcn_wave_barrier();
            int best = -1;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const double vj = lsdw_lane_f64(mine, j);
                if (j < ntr && vj > log_nfa) { log_nfa = vj; best = j; }
            }
            if (best >= 0) {
                if (phase == -1 || phase == 3) {
                    double pp = rec->p;
                    for (int j = 0; j <= best; ++j) pp = pp / 2;
                    rec->p = pp; rec->prec = PSL_DMUL(pp, PSL_PI);
                } else {
                    LsdnRect r = *rec;
                    for (int j = 0; j <= best; ++j) lsdn_shrink(r, phase);
                    *rec = r;
                }
            }
        }
        if (log_nfa > 0) return log_nfa;
    }
    return log_nfa;
}

// wave = rectangle; grid (chunks, frames), a workgroup strides over the rectangles of its frame.
// rects: [F][maxseg][PSL_LSD_RECT_F64] from k_lsd_grow3, nrect: [F]; segtmp: [F][maxseg][4], keep: [F][maxseg]
__global__ __launch_bounds__(256, 4) void k_lsd_nfa(LineParams P, LsdnTables T, const float* __restrict__ angdeg, const double* __restrict__ rects,
                                                 const int* __restrict__ nrect, float* __restrict__ segtmp, uint8_t* __restrict__ keep) {
    __shared__ int s_counts[4][10];
    const int frame = blockIdx.y, lane = threadIdx.x & 63;
    const int cnt = nrect[frame] < P.maxseg ? nrect[frame] : P.maxseg;
    for (int idx = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); idx < cnt; idx += (int)gridDim.x * 4) {
        const size_t o = (size_t)frame * P.maxseg + idx;
        const double* r = rects + o * PSL_LSD_RECT_F64;
        LsdnRect rec;
        rec.x1 = r[0]; rec.y1 = r[1]; rec.x2 = r[2]; rec.y2 = r[3]; rec.width = r[4]; rec.theta = r[5]; rec.dx = r[6]; rec.dy = r[7];
        rec.prec = P.prec; rec.p = P.p;
        const double log_nfa = lsdn_rect_improve(angdeg + (size_t)frame * P.W * P.H, P.W, P.H, T, &rec, lane, s_counts[threadIdx.x >> 6]);
        if (lane == 0) {
            const bool ok = log_nfa > 0;  // LOG_EPS = 0
            keep[o] = ok ? 1 : 0;
            if (ok) psl_lsd_store_segment(P, rec.x1, rec.y1, rec.x2, rec.y2, segtmp + 4 * o);
        }
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
