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
__device__ void lsdn_geom(const LsdnRect& r, int H, LsdnGeom* G) {
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

// total_pts and, for NP tolerances, alg_pts of rect_nfa()
template <int NP>
__device__ void lsdn_count(const float* __restrict__ ang, int W, const LsdnGeom& G, double theta, const double* prec, int lane, int* total, int* alg) {
    int tot = 0, al[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) al[j] = 0;
    auto pixel = [&](int x, int y) {
        const float a = ang[y * W + x];
        ++tot;
        if (a != PSL_LSD_NOTDEF) {
            const double f = lsdg_fold(PSL_DMUL((double)a, PSL_DEG2RAD), theta);
#pragma unroll
            for (int j = 0; j < NP; ++j) al[j] += f <= prec[j] ? 1 : 0;
        }
    };
    const int nrows = G.yb - G.ya + 1;
    if (nrows >= 24) {  // steep rectangles: lane = row
        for (int y = G.ya + lane; y <= G.yb; y += 64) {
            int xa, xb;
            lsdn_row_span(G, y, W, &xa, &xb);
            for (int x = xa; x <= xb; ++x) pixel(x, y);
        }
    } else {            // flat rectangles: lane = column
        for (int y = G.ya; y <= G.yb; ++y) {
            int xa, xb;
            lsdn_row_span(G, y, W, &xa, &xb);
            for (int x = xa + lane; x <= xb; x += 64) pixel(x, y);
        }
    }
    *total = lsdn_wave_sum(tot);
#pragma unroll
    for (int j = 0; j < NP; ++j) alg[j] = lsdn_wave_sum(al[j]);
}

// ---- nfa() ------------------------------------------------------------------------------------------------------------
// pow(x, n) for the integer-valued arguments log_gamma sees: exact products where libm's pow is exact as well (x <= 15,
// n <= 6); x^6 = (x^3)^2 with x^3 exact, i.e. one rounding - what a pow with < 1 ulp of error returns - for the Windschitl term
__device__ __forceinline__ double lsdn_log_gamma(double x) {
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

__device__ double lsdn_nfa(int n, int k, double p, double log_nt) {
    if (n == 0 || k == 0) return -log_nt;
    if (n == k) return PSL_DSUB(-log_nt, PSL_DMUL((double)n, psl_log10(p)));
    const double p_term = p / PSL_DSUB(1.0, p);
    double log1term = PSL_DSUB(PSL_DSUB(lsdn_log_gamma(PSL_DADD((double)n, 1.0)), lsdn_log_gamma(PSL_DADD((double)k, 1.0))),
                               lsdn_log_gamma(PSL_DADD((double)(n - k), 1.0)));
    log1term = PSL_DADD(PSL_DADD(log1term, PSL_DMUL((double)k, psl_log(p))), PSL_DMUL((double)(n - k), psl_log(PSL_DSUB(1.0, p))));
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
            const double pw = psl_pow_pos(mult_term, (double)(n - i + 1));
            const double err = PSL_DMUL(term, PSL_DSUB(PSL_DSUB(1.0, pw) / PSL_DSUB(1.0, mult_term), 1.0));
            if (err < PSL_DMUL(PSL_DMUL(0.1, fabs(PSL_DSUB(-psl_log10(bin_tail), log_nt))), bin_tail)) break;
        }
    }
    return PSL_DSUB(-psl_log10(bin_tail), log_nt);
}

// nfa() of up to five (n, k, p) triples, trial j on lane j; v[j] is returned to every lane
__device__ __forceinline__ void lsdn_nfa5(const int* n, const int* k, const double* p, int ntr, double log_nt, int lane, double* v) {
    const int nj = lane == 0 ? n[0] : lane == 1 ? n[1] : lane == 2 ? n[2] : lane == 3 ? n[3] : n[4];
    const int kj = lane == 0 ? k[0] : lane == 1 ? k[1] : lane == 2 ? k[2] : lane == 3 ? k[3] : k[4];
    const double pj = lane == 0 ? p[0] : lane == 1 ? p[1] : lane == 2 ? p[2] : lane == 3 ? p[3] : p[4];
    double mine = 0;
    if (lane < ntr) mine = lsdn_nfa(nj, kj, pj, log_nt);
#pragma unroll
    for (int j = 0; j < 5; ++j) v[j] = lsdw_lane_f64(mine, j);
}

// rect_improve(): returns log_nfa, *rec = the improved rectangle.  All lanes hold the same values.
__device__ double lsdn_rect_improve(const float* __restrict__ ang, int W, int H, double log_nt, LsdnRect* rec, int lane) {
    const double delta = 0.5, delta_2 = 0.25;
    LsdnGeom G;
    int n1, k1;
    lsdn_geom(*rec, H, &G);
    lsdn_count<1>(ang, W, G, rec->theta, &rec->prec, lane, &n1, &k1);
    double v1 = 0;
    if (lane == 0) v1 = lsdn_nfa(n1, k1, rec->p, log_nt);
    double log_nfa = lsdw_lane_f64(v1, 0);
    if (log_nfa > 0) return log_nfa;

    int n[5], k[5];
    double p[5], pr[5], v[5];
    // phase 1 and phase 5: finer precision - one geometry, five tolerances, ONE pass over the pixels
    auto finer = [&]() {
        double pp = rec->p;
#pragma unroll
        for (int j = 0; j < 5; ++j) { pp = pp / 2; p[j] = pp; pr[j] = PSL_DMUL(pp, PSL_PI); }
        int tot;
        lsdn_geom(*rec, H, &G);
        lsdn_count<5>(ang, W, G, rec->theta, pr, lane, &tot, k);
#pragma unroll
        for (int j = 0; j < 5; ++j) n[j] = tot;
        lsdn_nfa5(n, k, p, 5, log_nt, lane, v);
#pragma unroll
        for (int j = 0; j < 5; ++j)
            if (v[j] > log_nfa) { log_nfa = v[j]; rec->p = p[j]; rec->prec = pr[j]; }
    };
    finer();
    if (log_nfa > 0) return log_nfa;

    // phases 2-4: reduce the width (centred / one side / the other side).  The five trial rectangles of a phase do not
    // depend on each other's outcome: r is changed cumulatively, rec only receives copies.
#pragma unroll 1
    for (int phase = 0; phase < 3; ++phase) {
        LsdnRect r = *rec, tr[5];
        int ntr = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            tr[j] = r;
            if (PSL_DSUB(r.width, delta) >= 0.5) {
                if (phase == 1) {
                    r.x1 = PSL_DADD(r.x1, PSL_DMUL(-r.dy, delta_2)); r.y1 = PSL_DADD(r.y1, PSL_DMUL(r.dx, delta_2));
                    r.x2 = PSL_DADD(r.x2, PSL_DMUL(-r.dy, delta_2)); r.y2 = PSL_DADD(r.y2, PSL_DMUL(r.dx, delta_2));
                } else if (phase == 2) {
                    r.x1 = PSL_DSUB(r.x1, PSL_DMUL(-r.dy, delta_2)); r.y1 = PSL_DSUB(r.y1, PSL_DMUL(r.dx, delta_2));
                    r.x2 = PSL_DSUB(r.x2, PSL_DMUL(-r.dy, delta_2)); r.y2 = PSL_DSUB(r.y2, PSL_DMUL(r.dx, delta_2));
                }
                r.width = PSL_DSUB(r.width, delta);
                tr[j] = r;
                ntr = j + 1;  // the guard only ever turns false (the width shrinks monotonically)
            }
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            n[j] = 0; k[j] = 0; p[j] = r.p;
            if (j < ntr) {
                lsdn_geom(tr[j], H, &G);
                lsdn_count<1>(ang, W, G, tr[j].theta, &tr[j].prec, lane, &n[j], &k[j]);
            }
        }
        if (ntr) {
            lsdn_nfa5(n, k, p, ntr, log_nt, lane, v);
#pragma unroll
            for (int j = 0; j < 5; ++j)
                if (j < ntr && v[j] > log_nfa) { log_nfa = v[j]; *rec = tr[j]; }
        }
        if (log_nfa > 0) return log_nfa;
    }
    // phase 5: finer precision again, under the same width guard (all five trials or none)
    if (PSL_DSUB(rec->width, delta) >= 0.5) finer();
    return log_nfa;
}

// wave = rectangle; grid (chunks, frames), a workgroup strides over the rectangles of its frame.
// rects: [F][maxseg][PSL_LSD_RECT_F64] from k_lsd_grow3, nrect: [F]; segtmp: [F][maxseg][4], keep: [F][maxseg]
__global__ __launch_bounds__(256) void k_lsd_nfa(LineParams P, const float* __restrict__ angdeg, const double* __restrict__ rects,
                                                 const int* __restrict__ nrect, float* __restrict__ segtmp, uint8_t* __restrict__ keep) {
    const int frame = blockIdx.y, lane = threadIdx.x & 63;
    const int cnt = nrect[frame] < P.maxseg ? nrect[frame] : P.maxseg;
    for (int idx = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); idx < cnt; idx += (int)gridDim.x * 4) {
        const size_t o = (size_t)frame * P.maxseg + idx;
        const double* r = rects + o * PSL_LSD_RECT_F64;
        LsdnRect rec;
        rec.x1 = r[0]; rec.y1 = r[1]; rec.x2 = r[2]; rec.y2 = r[3]; rec.width = r[4]; rec.theta = r[5]; rec.dx = r[6]; rec.dy = r[7];
        rec.prec = P.prec; rec.p = P.p;
        const double log_nfa = lsdn_rect_improve(angdeg + (size_t)frame * P.W * P.H, P.W, P.H, P.log_nt, &rec, lane);
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
