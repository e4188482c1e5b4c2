// HIP kernels of the line front-end (gfx950, wave64). Product code.
// Reference behaviour reproduced:
//   cv::createLineSegmentDetector(LSD_REFINE_ADV | LSD_REFINE_STD) (OpenCV 3.x lsd.cpp; not in the
//   reference tree, restated from the published algorithm — SURVEY Appendix A.9) behind the stock contrib
//   wrapper cv::line_descriptor::LSDDetector the reference links (add_src/LineExtractor.cpp:336-337; its
//   vendored twin: Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:166-251).  ADV (rect_improve + NFA,
//   line_kernels3.h) is the default, STD is selected with pslfe_line_set_refine
//   optimizeAndMergeLines_lsd            add_src/uselongline.cpp:24-485
//   BinaryDescriptor::compute (LBD)      Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp:351-413, 1027-1373
//   CPartiallyRecoverConnectivity        add_src/PartiallyRecoverConnectivity.cpp:14-247
// Floating point: every operation whose rounding can reach an output is written with explicit,
// never-contracted single operations in the order the reference evaluates them.
#ifndef PSL_LINE_KERNELS_H
#define PSL_LINE_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pslfe.h"
#include "psl_device_math.h"

#define PSL_LSD_NOTDEF (-1024.0f)   // angle map label for "gradient undefined" (stored as f32 degrees)
#define PSL_SC64_QUAL __host__ __device__ static inline
#include "psl_sincos64.h"
#include "psl_sincos_glibc.h"
#define PSL_F64_QUAL __host__ __device__ static inline
#include "psl_f64math.h"
#define PSL_PI 3.1415926535897932384626433832795
#define PSL_DEG2RAD (PSL_PI / 180)

struct LineParams {
    int w, h, stride_pad;     // input image size
    int W, H;                 // LSD working image: cvRound(w*0.8) x cvRound(h*0.8)
    int maxseg;               // capacity of the raw segment list per frame
    int maxkl;                // capacity of the keyline list per frame (after merging)
    int nfeatures;            // nLSDFeature (top-N by response)
    double gk[7];             // Gaussian kernel sigma 0.75 (f64)
    double rho, prec, p;      // gradient threshold, angle tolerance (rad), p = ANG_TH/180
    double rho_q;             // largest q with sqrt(q) <= rho: `norm <= rho` decided on the squared magnitude
    int min_reg_size;
    const double* sctab;      // psl_sincostab.inc in HBM: the table of the glibc-exact double sin / cos (psl_sincos_glibc.h)
    int refine;               // 1 = LSD_REFINE_STD (segments leave k_lsd_grow4), 2 = LSD_REFINE_ADV (rectangles -> k_lsd_nfa -> k_lsd_emit)
    double log_nt;            // LOG_NT of the NFA: 5 (log10 W + log10 H) / 2 + log10 11
    int full_grad;            // k_lsd_grad stores the gradient magnitude of every pixel (debug tap), not only where the angle is defined
    int singles;              // k_lsd_grad flags static singletons in the `used` map (few-frame launches: it pays in the chain's latency, not in throughput)
    int lbdK[5];              // integer Gaussian 5x5 sigma 1 (OpenCV 3.2 8-bit path)
    float gaussG[63], gaussL[21];
};

// Tile -> (tx, ty, frame).  Many-frames launches use the XCD-aware grid (8, tiles, ceil(frames / 8)) so that all tiles
// of a frame share one L2 (see orb_kernels.h: psl_item_frame); otherwise the plain grid (tiles_x, tiles_y, frames).
__device__ __forceinline__ bool psl_tile_frame(int tiles_x, int nframes, int xcd, int* tx, int* ty, int* frame) {
    if (xcd) {
        *frame = (int)(blockIdx.z * 8 + blockIdx.x);
        *ty = (int)blockIdx.y / tiles_x;
        *tx = (int)blockIdx.y - *ty * tiles_x;
        return *frame < nframes;
    }
    *tx = (int)blockIdx.x; *ty = (int)blockIdx.y; *frame = (int)blockIdx.z;
    return true;
}

__device__ __forceinline__ int psl_reflect101i(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// ---------------------------------------------------------------------------------------------
// LSD step 1: GaussianBlur(CV_64F, 7x7, sigma 0.75, REFLECT_101) + resize(0.8, INTER_LINEAR) on
// doubles, in the summation order of OpenCV's RowFilter / SymmColumnFilter.  Scale 0.8 -> 1/scale =
// 1.25 and the bilinear weights {0.125,0.375,0.625,0.875} are exact, so the tables are computed
// inline exactly as cv::resize computes them.  A 64 x 16 tile of the working image needs <= 82 x 22
// blurred samples, i.e. <= 88 x 28 input pixels; they are staged once in LDS (reflect-101 applied
// while loading), row sums and column sums are formed once per sample, and the bilinear step reads
// the blurred tile.
// ---------------------------------------------------------------------------------------------
#define PSL_LS_IC 88
#define PSL_LS_IR 28
#define PSL_LS_BC 82
#define PSL_LS_BR 22
__device__ __forceinline__ void psl_lsd_src(int d, int ssize, int* s, float* f, bool clamp_coef) {
    const double sc = 1. / 0.8;
    float ff = (float)((d + 0.5) * sc - 0.5);
    int ss = (int)__builtin_floorf(ff);
    ff -= ss;
    if (clamp_coef) {
        if (ss < 0) { ff = 0; ss = 0; }
        if (ss >= ssize - 1) { ff = 0; ss = ssize - 1; }
    }
    *s = ss; *f = ff;
}

__global__ __launch_bounds__(256) void k_lsd_scale_tiled(LineParams P, const uint8_t* __restrict__ gray, int stride, size_t fstride,
                                                          double* __restrict__ scaled, int nframes, int xcd) {
    __shared__ __attribute__((aligned(16))) uint8_t s_in[PSL_LS_IR * PSL_LS_IC + 16];
    __shared__ double s_rs[PSL_LS_IR * PSL_LS_BC];
    __shared__ double s_bl[PSL_LS_BR * PSL_LS_BC];
    const int tid = threadIdx.x;
    int tx, ty, frame;
    if (!psl_tile_frame((P.W + 63) / 64, nframes, xcd, &tx, &ty, &frame)) return;
    const int dx0 = tx * 64, dy0 = ty * 16;
    const uint8_t* img = gray + (size_t)frame * fstride;
    // blurred sample range of this tile
    int s, bx0, bx1, by0, by1;
    float f;
    psl_lsd_src(dx0, P.w, &bx0, &f, true);
    psl_lsd_src(min(dx0 + 63, P.W - 1), P.w, &s, &f, true);
    bx1 = min(s + 1, P.w - 1);
    psl_lsd_src(dy0, P.h, &s, &f, false);
    by0 = min(max(s, 0), P.h - 1);
    psl_lsd_src(min(dy0 + 15, P.H - 1), P.h, &s, &f, false);
    by1 = min(max(s + 1, 0), P.h - 1);
    const int nbc = bx1 - bx0 + 1, nbr = by1 - by0 + 1;  // <= 82, <= 22 for scale 0.8
    const int nic = nbc + 6, nir = nbr + 6;
    if (bx0 - 3 >= 0 && bx0 - 3 + nic <= P.w && bx0 - 3 + PSL_LS_IC <= stride) {
        // no column reflection in this tile (6 of 8 tile columns): whole rows as 22 dwords (unaligned in HBM, aligned in LDS)
        for (int k = tid; k < nir * (PSL_LS_IC / 4); k += 256) {
            const int r = k / (PSL_LS_IC / 4), d = k - r * (PSL_LS_IC / 4);
            uint32_t w;
            __builtin_memcpy(&w, img + (size_t)psl_reflect101i(by0 - 3 + r, P.h) * stride + (bx0 - 3) + 4 * d, 4);
            *reinterpret_cast<uint32_t*>(&s_in[r * PSL_LS_IC + 4 * d]) = w;
        }
    } else {
        const uint32_t mic = (1048576u + (uint32_t)nic - 1u) / (uint32_t)nic;  // k / nic == (k * mic) >> 20 for k < 4096
        for (int k = tid; k < nir * nic; k += 256) {
            const int r = (int)(((uint32_t)k * mic) >> 20), c = k - r * nic;
            s_in[r * PSL_LS_IC + c] = img[(size_t)psl_reflect101i(by0 - 3 + r, P.h) * stride + psl_reflect101i(bx0 - 3 + c, P.w)];
        }
    }
    __syncthreads();
    // RowFilter: s = k0*S0; s += k1*S1; ...  A thread makes 4 adjacent row sums from 10 input bytes (three aligned dwords)
    // instead of reading 7 bytes per sum (the kernel was LDS-bound); 4 rather than 8 per thread so that the <= 588 work items
    // fill the 256 threads in 3 even trips (8 per thread: 2 trips, the second one 20 % occupied).
    {
        const int ngr = (nbc + 3) >> 2;  // <= 21
        const uint32_t mgr = (1048576u + (uint32_t)ngr - 1u) / (uint32_t)ngr;
        for (int k = tid; k < nir * ngr; k += 256) {
            const int r = (int)(((uint32_t)k * mgr) >> 20), g = k - r * ngr;
            const uint32_t* in32 = reinterpret_cast<const uint32_t*>(&s_in[r * PSL_LS_IC + 4 * g]);
            const uint32_t w0 = in32[0], w1 = in32[1], w2 = in32[2];
            double v[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const uint32_t w = j < 4 ? w0 : (j < 8 ? w1 : w2);
                v[j] = (double)((w >> (8 * (j & 3))) & 0xffu);
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (4 * g + o < nbc) {
                    double acc = PSL_DMUL(P.gk[0], v[o]);
#pragma unroll
                    for (int j = 1; j < 7; ++j) acc = PSL_DADD(acc, PSL_DMUL(P.gk[j], v[o + j]));
                    s_rs[r * PSL_LS_BC + 4 * g + o] = acc;
                }
            }
        }
    }
    __syncthreads();
    // SymmColumnFilter: centre, then (S[k] + S[-k]) * ky[k].  A thread makes 8 vertically adjacent samples of one
    // column from 14 row sums read once.
    {
        const int nst = (nbr + 7) >> 3;  // <= 3
        for (int k = tid; k < nst * nbc; k += 256) {
            const int st = k / nbc, c = k - st * nbc;
            const int r0 = st * 8;
            double v[14];
#pragma unroll
            for (int j = 0; j < 14; ++j) v[j] = s_rs[min(r0 + j, nir - 1) * PSL_LS_BC + c];
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                if (r0 + o < nbr) {
                    double acc = PSL_DMUL(P.gk[3], v[o + 3]);
#pragma unroll
                    for (int j = 1; j <= 3; ++j) acc = PSL_DADD(acc, PSL_DMUL(P.gk[3 + j], PSL_DADD(v[o + 3 + j], v[o + 3 - j])));
                    s_bl[(r0 + o) * PSL_LS_BC + c] = acc;
                }
            }
        }
    }
    __syncthreads();
    int sx;
    float fx;
    psl_lsd_src(min(dx0 + (tid & 63), P.W - 1), P.w, &sx, &fx, true);  // a thread's four pixels share the column
    const double a0 = (double)(1.f - fx), a1 = (double)fx;
    for (int k = tid; k < 64 * 16; k += 256) {
        const int dx = dx0 + (k & 63), dy = dy0 + (k >> 6);
        if (dx >= P.W || dy >= P.H) continue;
        int sy;
        float fy;
        psl_lsd_src(dy, P.h, &sy, &fy, false);
        const double b0 = (double)(1.f - fy), b1 = (double)fy;
        const int sy0 = min(max(sy, 0), P.h - 1) - by0, sy1 = min(max(sy + 1, 0), P.h - 1) - by0;
        const int cx = sx - bx0;
        double h0, h1;
        if (sx + 1 < P.w) {
            h0 = PSL_DADD(PSL_DMUL(s_bl[sy0 * PSL_LS_BC + cx], a0), PSL_DMUL(s_bl[sy0 * PSL_LS_BC + cx + 1], a1));
            h1 = PSL_DADD(PSL_DMUL(s_bl[sy1 * PSL_LS_BC + cx], a0), PSL_DMUL(s_bl[sy1 * PSL_LS_BC + cx + 1], a1));
        } else {
            h0 = s_bl[sy0 * PSL_LS_BC + cx];
            h1 = s_bl[sy1 * PSL_LS_BC + cx];
        }
        scaled[(size_t)frame * P.W * P.H + (size_t)dy * P.W + dx] = PSL_DADD(PSL_DMUL(h0, b0), PSL_DMUL(h1, b1));
    }
}

// LSD step 2 (ll_angle): 2x2 gradient, norm (f64, correctly rounded sqrt), level-line angle by the
// f32 fastAtan2 polynomial.  The angle is stored as f32 degrees (the reference's double is exactly
// double(deg) * DEG_TO_RADS, recomputed where it is used); NOTDEF = -1024.
// Also tabulates, per pixel with a defined angle a = (double)deg * DEG_TO_RADS, the four values the
// region-growing chain needs: cosf((float)a), sinf((float)a) (every pixel that joins a region) and
// (float)cos(a), (float)sin(a) (the seed pixel), so that the serial chain contains no trigonometry.
#ifndef PSL_LSD_SINGLES
#define PSL_LSD_SINGLES 1   // 0: no static-singleton flags (A/B, tools/ab_build.sh)
#endif
#define PSL_GRAD_TH 16  // tile height of k_lsd_grad (64 x 16 pixels per workgroup, 4 per thread)
// squared gradient magnitude of pixel (x, y); false where the reference leaves the angle undefined by construction
__device__ __forceinline__ bool psl_lsd_norm(const LineParams& P, const double* __restrict__ img, int x, int y, double* q) {
    if (x < 0 || y < 0 || x >= P.W - 1 || y >= P.H - 1) return false;
    const double* r0 = img + (size_t)y * P.W;
    const double* r1 = r0 + P.W;
    const double DA = PSL_DSUB(r1[x + 1], r0[x]), BC = PSL_DSUB(r0[x + 1], r1[x]);
    const double gx = PSL_DADD(DA, BC), gy = PSL_DSUB(DA, BC);
    *q = PSL_DADD(PSL_DMUL(gx, gx), PSL_DMUL(gy, gy)) / 4;
    return true;
}

// trig[o] = (cosf, sinf) of the pixel's angle, (0, 0) where the angle is undefined: what a round of k_lsd_grow4 needs of a neighbour
// (the angle itself only inside the margin of the decision, read from angdeg then); seedt[o] = (float)cos / sin of the double
// angle, read once per seed.  A record is only ever read for a pixel with a defined angle or for a neighbour of one, so it is written
// only for those - about a third of the pixels (16-byte records for every pixel made this kernel HBM-write-bound, 9.4 - 11.5 ms).
// Which pixels have a defined neighbour is known from a flag tile in LDS: 64 x 16 pixels per workgroup plus a one-pixel ring whose
// magnitudes are recomputed.  8-byte records: a window row of the growing is 64 bytes, 1 - 2 HBM sectors instead of 2 - 3.
// |theta - ad| folded into [0, pi] and compared with prec, as lsdw_aligned but without branches: both differences are formed
// and one is selected (the serial acceptance chain below executes this once per accepted pixel).
__device__ __forceinline__ double lsdg_fold(double ad, double theta) {
    const double d = __builtin_fabs(PSL_DSUB(theta, ad));
    const double d2 = __builtin_fabs(PSL_DSUB(d, 2 * PSL_PI));
    return d > (3 * PSL_PI) / 2 ? d2 : d;
}
__device__ __forceinline__ bool lsdg_aligned(double ad, double theta, double prec) { return lsdg_fold(ad, theta) <= prec; }

// The heavy part of a defined pixel (f64 square root, fastAtan2, two sine / cosine pairs: ~160 vector instructions) runs on the
// COMPACTED list of the tile's defined pixels: ~15 % of the pixels have a defined angle and they are scattered, so that run in place
// nearly every wave executed it for each of its four pixel slots with a handful of lanes active (7.7 ms per 6144 frames).  Phase A
// (thread = 4 pixels of a column): gradient and squared magnitude, the threshold, NOTDEF angles, the flag tile, and the defined
// pixels appended to an LDS list (any order: every output is a per-pixel store).  Phase B (thread = list entry): magnitude, angle,
// records.  Phase C: the neighbour records of the pixels that have a defined pixel in their 3 x 3.
#ifndef PSL_GRAD_WAVES
#define PSL_GRAD_WAVES 1
#endif
__global__ __launch_bounds__(256, PSL_GRAD_WAVES) void k_lsd_grad(LineParams P, const double* __restrict__ scaled, float* __restrict__ angdeg,
                                                   double* __restrict__ modgrad, float2* __restrict__ trig, float2* __restrict__ seedt, uint8_t* __restrict__ used,
                                                   int* __restrict__ weight, int nframes, int xcd) {
    __shared__ uint8_t s_def[PSL_GRAD_TH + 2][68];
    __shared__ float2 s_cs[PSL_GRAD_TH * 64];    // (cosf, sinf) of the tile's pixels, (0, 0) = undefined
    __shared__ float s_deg[PSL_GRAD_TH * 64];    // their angles: written to HBM as whole rows in phase C
    __shared__ uint8_t s_single[PSL_GRAD_TH * 64];  // 1: no neighbour's angle is aligned with this pixel's (phase B2)
    __shared__ uint16_t s_px[PSL_GRAD_TH * 64];  // list: pixel of the tile (row * 64 + column); its gradient is formed again from the
                                                 // four values (L1 hits): keeping it in LDS cost a third of the resident workgroups
    __shared__ int s_n;
    int tile_x, tile_y, frame;  // XCD-aware grid for many frames: the tiles of a frame share one L2 (their halos overlap)
    if (!psl_tile_frame((P.W + 63) / 64, nframes, xcd, &tile_x, &tile_y, &frame)) return;
    const int tid = threadIdx.x;
    const int x0 = tile_x * 64, y0 = tile_y * PSL_GRAD_TH;
    const int tx = tid & 63, ty = tid >> 6, x = x0 + tx;
    const size_t fo = (size_t)frame * P.W * P.H;
    const double* img = scaled + fo;
    if (tid == 0) s_n = 0;
    // all sixteen loads of the thread's four pixels first (clamped addresses), then the arithmetic: one memory round trip
    double w00[4], w01[4], w10[4], w11[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int yy = min(y0 + ty + 4 * i, P.H - 2), xx = min(x, P.W - 2);
        const double* r0 = img + (size_t)yy * P.W + xx;
        w00[i] = r0[0]; w01[i] = r0[1]; w10[i] = r0[P.W]; w11[i] = r0[P.W + 1];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty + 4 * i, y = y0 + r;
        bool def = false;
        double gx = 0.0, gy = 0.0, q = 0.0;
        if (x < P.W && y < P.H) {
            const size_t o = fo + (size_t)y * P.W + x;
            const bool inner = x < P.W - 1 && y < P.H - 1;  // last column / row: magnitude 0, angle undefined by construction
            if (inner) {
                const double DA = PSL_DSUB(w11[i], w00[i]), BC = PSL_DSUB(w01[i], w10[i]);
                gx = PSL_DADD(DA, BC); gy = PSL_DSUB(DA, BC);
                q = PSL_DADD(PSL_DMUL(gx, gx), PSL_DMUL(gy, gy)) / 4;
            }
            // norm = sqrt(q) <= rho  <=>  q <= rho_q (the square root is correctly rounded, hence monotone): the ~85 % of the
            // pixels below the threshold need no square root.  The magnitude is read back only for pixels of regions (defined
            // angle): many-frames launches skip the other stores.
            def = inner && !(q <= P.rho_q);
            if (!def && P.full_grad) modgrad[o] = __dsqrt_rn(q);
        }
        const unsigned long long dmask = __ballot(def);
        if (dmask) {  // one LDS atomic per wave, positions by rank
            int base = 0;
            if (tx == 0) base = atomicAdd(&s_n, __popcll(dmask));
            base = __builtin_amdgcn_readfirstlane(base);
            if (def) {
                const int k = base + __popcll(dmask & ((1ull << tx) - 1ull));
                s_px[k] = (uint16_t)(r * 64 + tx);
            }
        }
        s_def[r + 1][tx + 1] = def;
        s_cs[r * 64 + tx] = make_float2(0.f, 0.f);
        s_deg[r * 64 + tx] = PSL_LSD_NOTDEF;
        s_single[r * 64 + tx] = 0;
    }
    // the ring around the tile: 2 x 66 + 2 x 16 = 164 pixels
    if (tid < 2 * 66 + 2 * PSL_GRAD_TH) {
        int r, c;
        if (tid < 132) { r = tid < 66 ? 0 : PSL_GRAD_TH + 1; c = tid < 66 ? tid : tid - 66; }
        else { const int k = tid - 132; r = 1 + (k >> 1); c = (k & 1) ? 65 : 0; }
        double q = 0.0;
        const bool inner = psl_lsd_norm(P, img, x0 - 1 + c, y0 - 1 + r, &q);
        s_def[r][c] = inner && !(q <= P.rho_q);
    }
    __syncthreads();
    const int n = s_n;
    if (weight && tid == 0 && n > 0) atomicAdd(weight + frame, n);   // the frame's defined pixels: what the region growing's work scales with (k_frame_order)
    for (int k = tid; k < n; k += 256) {
        const int px = s_px[k], r = px >> 6, c = px & 63;
        const size_t o = fo + (size_t)(y0 + r) * P.W + (x0 + c);
        const double* r0 = img + (size_t)(y0 + r) * P.W + (x0 + c);
        const double DA = PSL_DSUB(r0[P.W + 1], r0[0]), BC = PSL_DSUB(r0[1], r0[P.W]);
        const double gx = PSL_DADD(DA, BC), gy = PSL_DSUB(DA, BC);
        modgrad[o] = __dsqrt_rn(PSL_DADD(PSL_DMUL(gx, gx), PSL_DMUL(gy, gy)) / 4);
        const float deg = psl_fast_atan2((float)gx, (float)(-gy));
        const double ad = PSL_DMUL((double)deg, PSL_DEG2RAD);
        float sn, cs;
        psl_sincosf((float)ad, &sn, &cs);
        float cd, sd;  // (float)cos(ad), (float)sin(ad): restricted-range evaluation, pinned exhaustively (psl_sincos64.h)
        psl_cos_sin_2pi_f32(ad, &cd, &sd);
        seedt[o] = make_float2(cd, sd);
        s_deg[px] = deg;
        s_cs[px] = make_float2(cs, sn);
    }
    __syncthreads();
    // Phase B2, STATIC SINGLETONS: a pixel none of whose eight neighbours has an angle aligned with ITS angle (the reference's
    // isAligned with the detector's tolerance) grows, as a seed, into a region of exactly one pixel whatever has been used by then -
    // a region the size test drops.  Such a seed is marked in the `used` map (value 2) and k_lsd_grow4's scan only takes it (used = 1)
    // instead of growing it: 40 % of the region starts of the headline scene.  Measured A/B in one session: one frame at a time -1 %
    // (structure scene) to -7 % (textured scene, 35.0 -> 32.6 ms); launches of thousands of frames gain nothing (the launch lasts as
    // long as its heaviest frames) and pay 0.3 ms here: P.singles is set for launches of at most 64 frames.  (It can still be absorbed by a neighbour's region
    // before the scan reaches it: the test there is against the REGION's angle.)  Only pixels whose neighbours all lie in the tile
    // are examined; the others go the ordinary way.
    for (int k = tid; PSL_LSD_SINGLES && P.singles && k < n; k += 256) {
        const int px = s_px[k], r = px >> 6, c = px & 63;
        if (r < 1 || r > PSL_GRAD_TH - 2 || c < 1 || c > 62) continue;
        const double ap = PSL_DMUL((double)s_deg[px], PSL_DEG2RAD);
        bool alone = true;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            if (j == 4) continue;
            const float an = s_deg[px + (j / 3 - 1) * 64 + (j % 3 - 1)];
            if (an != PSL_LSD_NOTDEF && lsdg_aligned(PSL_DMUL((double)an, PSL_DEG2RAD), ap, P.prec)) alone = false;
        }
        if (alone) s_single[px] = 1;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty + 4 * i, y = y0 + r;
        if (x >= P.W || y >= P.H) continue;
        if (s_single[r * 64 + tx]) used[fo + (size_t)y * P.W + x] = 2;
        const uint8_t* d0 = &s_def[r][tx];
        const uint8_t* d1 = &s_def[r + 1][tx];
        const uint8_t* d2 = &s_def[r + 2][tx];
        const bool any = (d0[0] | d0[1] | d0[2] | d1[0] | d1[1] | d1[2] | d2[0] | d2[1] | d2[2]) != 0;
        angdeg[fo + (size_t)y * P.W + x] = s_deg[r * 64 + tx];
        if (any) trig[fo + (size_t)y * P.W + x] = s_cs[r * 64 + tx];
    }
}

// HEAVIEST FRAMES FIRST.  k_lsd_grow4 is one wave per frame and the hardware starts workgroups in index order as wave slots free up
// - greedy list scheduling.  A launch of 12288 frames on 8192 slots (8 waves per SIMD) lasted 245 ms while its average wave lived
// 127 ms (profiles/r03e: 22 % of the slot-time idle): the frames started last ran alone at the end.  Ordered by decreasing weight
// (longest processing time first) the tail is made of the lightest frames.  weight = number of pixels with a defined gradient,
// counted by k_lsd_grad (one atomic per tile); the order is a counting sort over 1024 weight classes by one workgroup (the order
// inside a class is whatever the atomics give: it only affects the schedule, never a result).
#ifndef PSL_LSD_SUBBATCH
#define PSL_LSD_SUBBATCH 2048   // frames whose f64 working image is resident between k_lsd_scale_tiled and k_lsd_grad (pslfe_line.hip: run_lsd)
#endif
#ifndef PSL_FRAME_ORDER
#define PSL_FRAME_ORDER 1   // 0: frames in index order (A/B, tools/ab_build.sh)
#endif
#define PSL_ORDER_CLASSES 1024
__global__ __launch_bounds__(1024) void k_frame_order(const int* __restrict__ weight, int nframes, int wmax, int* __restrict__ order) {
    __shared__ int s_cnt[PSL_ORDER_CLASSES];
    __shared__ int s_w[17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    s_cnt[tid] = 0;
    __syncthreads();
    const long long scale = wmax > 0 ? wmax : 1;
    auto cls = [&](int w) {   // class 0 = heaviest
        long long c = (long long)(w < 0 ? 0 : (w > wmax ? wmax : w)) * (PSL_ORDER_CLASSES - 1) / scale;
        return PSL_ORDER_CLASSES - 1 - (int)c;
    };
    for (int f = tid; f < nframes; f += 1024) atomicAdd(&s_cnt[cls(weight[f])], 1);
    __syncthreads();
    const int mine = s_cnt[tid];
    int inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int k = 0; k < 16; ++k) { const int t = s_w[k]; s_w[k] = acc; acc += t; }
    }
    __syncthreads();
    s_cnt[tid] = inc - mine + s_w[wave];   // exclusive start of the class
    __syncthreads();
    for (int f = tid; f < nframes; f += 1024) order[atomicAdd(&s_cnt[cls(weight[f])], 1)] = f;
}

struct LsdRect { double x1, y1, x2, y2, width, theta, dx, dy; };

__device__ __forceinline__ double psl_angle_diff_signed(double a, double b) {
    double diff = PSL_DSUB(a, b);
    while (diff <= -PSL_PI) diff = PSL_DADD(diff, 2 * PSL_PI);
    while (diff > PSL_PI) diff = PSL_DSUB(diff, 2 * PSL_PI);
    return diff;
}

__device__ __forceinline__ double psl_dist_sq(double x1, double y1, double x2, double y2) {
    const double dx = PSL_DSUB(x2, x1), dy = PSL_DSUB(y2, y1);
    return PSL_DADD(PSL_DMUL(dx, dx), PSL_DMUL(dy, dy));
}

__device__ __forceinline__ double psl_lsd_density(int reg_size, const LsdRect& r) {
    return (double)reg_size / PSL_DMUL(__dsqrt_rn(psl_dist_sq(r.x1, r.y1, r.x2, r.y2)), r.width);
}

// ---------------------------------------------------------------------------------------------
// LSD steps 3-6: seeds in raster order, region growing (8-connected, running mean angle), rectangle by
// inertia axes, density refinement - wave-parallel and still exact (k_lsd_grow4).  The algorithm is a serial
// chain through `used` and the running angle; one wave owns one frame and parallelism comes from the frames
// of the batch.
//  * seeds: 256 pixels per step (4 coalesced loads in flight), ballot -> first candidate; the `used` words of the step are looked
//    at again only after a region that took pixels in front of the scan;
//  * region growing: window rounds (lsdg_region_grow4 below) - the records of an 8 x 8 pixel window in ONE round trip, then queue
//    entries are popped for as long as their neighbours are lanes of the wave; the running sums are added in the reference's order,
//    decisions come from a rigorous cross / dot test and, within its margin, from the reference's own arithmetic; the last 1024 queue
//    entries live in an LDS ring, older ones in HBM;
//  * sums whose rounding depends on the order (centroid, inertia, refine statistics) are formed as
//    "terms in parallel, additions in series" (64 terms staged in LDS per step); min/max extents are
//    order independent and use wave reductions;
//  * the `used` flag is the fourth word of the pixel's 16-byte record in HBM, written and read by the same wave only (workgroup
//    scope: the CU's L1 is coherent for its own stores); stores that may still be in flight when the next window is loaded are
//    known from the previous round's acceptance mask instead of being waited for.
// Wave-uniform logic runs on the scalar unit (masks, counters) or redundantly in all lanes (the float sums).
// ---------------------------------------------------------------------------------------------
#ifndef PSL_LSD_RING
#define PSL_LSD_RING 512    // queue entries kept in LDS (a power of two >= 128; 1024 until round 3: 512 x 4 B lets 32 waves per CU fit the 160 KB of LDS).  A frontier that lags more than half of this behind the queue's end
                            // does not occur on 8-bit images (the gradient threshold of 5.2 grey levels per pixel limits a region to ~50 pixels along
                            // its gradient, hence the breadth-first frontier to ~100 entries): that path is exercised by building with
                            // -DPSL_LSD_RING=128 and running tests/test_line_gpu.py (its "band" image reaches a lag of 85)
#endif
#define PSL_LSD_HALF (PSL_LSD_RING / 2)
#ifndef PSL_REDUCE_SERIAL
#define PSL_REDUCE_SERIAL 0   // 1: reduce_region_radius walked entry by entry as the reference writes it (A/B, tools/ab_build.sh)
#endif

struct LsdW {
    int W, H, lane;
    const float* ang;
    const double* mod;
    const float2* trig;    // (cosf, sinf) per pixel, (0, 0) = angle undefined
    uint8_t* used;         // the reference's `used` map, one byte per pixel
    const float2* seedt;   // (float)cos, (float)sin of the double angle (seed pixels)
    const double* sctab;
    uint32_t* ring;   // LDS mirror of reg[idx & (RING-1)]
    uint32_t* map;    // LDS [64]: window lane -> queue index + 1 (lsdg_region_grow4)
    unsigned long long interior;  // lanes of the 8 x 8 window's 6 x 6 interior
    double* term;     // LDS [3][64]
    uint32_t* reg;    // HBM queue
    uint32_t* ubits;  // LU variants: the `used` map's "used" bit per pixel, in LDS
};

__device__ __forceinline__ uint32_t lsdw_reg(const LsdW& F, int idx, int reg_size) {
    return (reg_size - idx <= PSL_LSD_RING) ? F.ring[idx & (PSL_LSD_RING - 1)] : F.reg[idx];
}

// the `used` map: 0 free, 1 used, 2 free and a static singleton (k_lsd_grad, phase B2)
// LU = 1 (launches of a few frames: one workgroup per frame has the CU's LDS to itself): "used" is a BIT PER PIXEL IN LDS (24 KB at 640x480, 96 KB at
// 1280x960) on top of the map in memory, which then only carries k_lsd_grad's static-singleton flags and is never written.  Vector-memory operations
// return in issue order, so with the marks in memory every window load of the chain waited for the mark stores issued in front of it - ~0.8 us per
// round on an otherwise idle chip, where an L2 hit is ~0.1 us.  With thousands of frames in flight other waves cover that wait (and the bitmaps of 32 frames
// would not fit a CU's LDS): LU = 0 there.
typedef __attribute__((address_space(3))) uint32_t lds_ubits;
template <int LU>
__device__ __forceinline__ bool lsdg_used(const LsdW& F, int a) {
    if (LU) return (((lds_ubits*)F.ubits)[a >> 5] >> (a & 31)) & 1u;
    return __hip_atomic_load(F.used + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 1u;
}
template <int LU>
__device__ __forceinline__ uint32_t lsdg_state(const LsdW& F, int a) {
    const uint32_t g = __hip_atomic_load(F.used + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (LU) return lsdg_used<1>(F, a) ? 1u : g;
    return g;
}
template <int LU>
__device__ __forceinline__ void lsdg_mark(const LsdW& F, int a, uint8_t v) {   // v = 1: used, v = 0: released
    if (LU) {
        if (v) __hip_atomic_fetch_or((lds_ubits*)F.ubits + (a >> 5), 1u << (a & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_and((lds_ubits*)F.ubits + (a >> 5), ~(1u << (a & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        __hip_atomic_store(F.used + a, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
// pixel index x + y W with a 24-bit multiply (a full-rate instruction; the 64-bit multiply-add the compiler forms for a pointer index is quarter rate): coordinates and W are
// far below 2^24 / H (the scaled image of a 4096 x 4096 frame would be the limit)
__device__ __forceinline__ int lsdw_pix(const LsdW& F, int x, int y) { return __mul24(y, F.W) + x; }

// ---------------------------------------------------------------------------------------------
// Region growing in WINDOW ROUNDS.  The queue order of the reference (breadth first, 3 x 3 neighbours in raster order, the
// region angle brought up to date after every pixel) is kept exactly; what changes is how much of it one memory round trip serves.
// A round loads the 16-byte records of an 8 x 8 pixel window (lane = pixel, row major; the first entry to pop sits at column 3,
// row 1: a region's seed is its first pixel in raster order, growth goes down and sideways) and then pops queue entries for as long
// as the popped pixel lies in the window's 6 x 6 interior, i.e. all of its neighbours are lanes of the wave - including entries
// pushed in this very round.  On a line 2 - 3 pixels wide a round advances the frontier by 5 - 6 pixels instead of one (rounds per
// 640x480 frame: 12.5 k with the 7-entries-per-round scheme before, 4.6 k now; tools/grow_stats.py replays the oracle's queue).
//  * which lane holds queue entry i: `seq` (lane -> queue index): entries that were in the queue when the round began are mapped
//    through a 64-word LDS table (up to 64 of them), entries pushed in the round get theirs when they are accepted;
//  * neighbours of the popped lane le: the bit pattern 0x070707 << (le - 9) of a wave-uniform 64-bit mask, ANDed with `live`
//    (defined angle, not used); ascending bit order is the reference's visiting order; all of this is scalar-unit work;
//  * decision: the reference compares fold(|fastAtan2(sum) - a|) with prec.  The angle phi between the running sum S and the
//    pixel's unit vector u = (cosf a, sinf a) - both are at hand - differs from that quantity by < 4e-4 rad (fastAtan2's error
//    < 0.02 deg, tests/test_oracle_line_cpu.py; float rounding ~1e-6), so with a margin m = 2e-3 rad
//        |S x u| <= tan(prec - m) (S . u)  =>  joins,      |S x u| >= tan(prec + m) (S . u)  =>  does not
//    (8 vector instructions for all 64 lanes, against ~45 for fastAtan2 + the f64 compare); only a lane in between - 0.5 % of the
//    decisions - takes the exact path, which is the reference's arithmetic.  Valid while prec + m < 1.5 rad (then every added
//    vector has a positive projection on S and |S| >= 1); a refinement tolerance beyond that uses the exact path only;
//  * the sums are added in acceptance order (bit-identical), the region angle is evaluated only when the exact path or the end of
//    the region needs it;
//  * marks: one store instruction per round (the accepted lanes mark their own records) and one LDS store for the queue.  The next
//    round's load is issued right behind that store: the pixels of the previous round are therefore recognised from its
//    acceptance mask (PA, window origin pox / poy), everything older has completed before the previous round's load returned
//    (vector-memory operations complete in issue order), so the `used` word needs no wait of its own.
// ---------------------------------------------------------------------------------------------
#define PSL_G4_MARGIN 2.0e-3f
#ifdef PSL_GROW_STATS   // diagnostic build only (PSLFE_EXTRA_FLAGS=-DPSL_GROW_STATS): loop trip counts of frame 0, printed by the kernel
__device__ unsigned long long g_gstats[16];
#define GS(k) (++gs[k])
#else
#define GS(k)
#endif
// a wave-uniform 64-bit value the compiler no longer knows to be uniform (after inline asm / volatile reloads): back into SGPRs
__device__ __forceinline__ unsigned long long lsdg_uniform64(unsigned long long v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
struct LsdgFast { float t_hi, t_lo; int ok; unsigned long long th2; };
__device__ __forceinline__ unsigned long long lsdg_pack2(float lo, float hi) {
    return ((unsigned long long)__float_as_uint(hi) << 32) | (unsigned long long)__float_as_uint(lo);
}
__device__ __forceinline__ float lsdg_lo(unsigned long long v) { return __uint_as_float((uint32_t)v); }
__device__ __forceinline__ float lsdg_hi(unsigned long long v) { return __uint_as_float((uint32_t)(v >> 32)); }
__device__ __forceinline__ LsdgFast lsdg_fast_setup(double prec) {
    LsdgFast f;
    const float p = (float)prec;
    f.ok = p + PSL_G4_MARGIN < 1.5f;
    // cr < t_hi max(dot, 0) => joins, cr >= t_lo max(dot, 0) => does not; -1 / +inf: never true (inf * 0 = NaN compares false)
    f.t_hi = f.ok && p - PSL_G4_MARGIN > 0.f ? psl_tanf(p - PSL_G4_MARGIN) * (1.f - 1e-5f) : -1.f;
    f.t_lo = f.ok ? psl_tanf(p + PSL_G4_MARGIN) * (1.f + 1e-5f) : __builtin_inff();
    // lsdg_pops2 compares with t dot instead of t max(dot, 0): |cr| < t_hi dot is false for dot <= 0 as it is (t_hi >= 0), |cr| >= t_lo dot true;
    // "never": t_hi = 0 (|cr| < +-0 is false), t_lo = NaN (every comparison false)
    f.th2 = lsdg_pack2(f.t_hi > 0.f ? f.t_hi : 0.f, f.ok ? f.t_lo : __builtin_nanf(""));
    return f;
}
// The marks of the last round that ran (of this region or of the one before): possibly still on their way to memory.
struct LsdgPend { unsigned long long PA; int pox, poy; };

// The decision loop over the candidates of one popped entry, hand-scheduled: hipcc spends ~31 scalar instructions per accepted pixel on
// this wave-uniform mask logic (every uniform bool becomes a 64-bit mask, a compare and a branch); written out it is 15.  `cand`:
// candidates still to decide (absolute lane positions, ascending = visiting order).  Per iteration: masks of sure joins / sure
// rejects against the current sums unless still valid (`fresh`), skip the sure rejects in front, take the first other candidate c:
// a sure join is added (sums in order, its queue index, `live`), anything else ends the block with c returned (it is removed from
// cand; the caller runs the reference's arithmetic for it).  Returns -1 when all candidates are decided.
// gfx950 wait states observed: an SGPR written by a VALU instruction (v_readlane) is read by a VALU instruction no sooner than the third
// instruction after it.
__device__ __forceinline__ int lsdg_decide(unsigned long long& cand, unsigned long long& live, unsigned long long& RA, unsigned long long& RN, int& fresh,
                                           int& reg_size, float& sumdx, float& sumdy, int& seq, float cs, float sn, float t_hi, float t_lo) {
    int cx, c, sa, sb;
    float t0, t1, t2;
    unsigned long long tm;
    asm volatile(
        "s_mov_b32 %[cx], -1\n\t"
        ".Ltop%=:\n\t"
        "s_cmp_lg_u32 %[fresh], 0\n\t"
        "s_cbranch_scc1 .Lhave%=\n\t"
        "v_mul_f32 %[t0], %[sy], %[sn]\n\t"
        "v_mul_f32 %[t1], %[sy], %[cs]\n\t"
        "v_fmac_f32 %[t0], %[sx], %[cs]\n\t"
        "v_fma_f32 %[t1], %[sx], %[sn], -%[t1]\n\t"
        "v_max_f32 %[t0], 0, %[t0]\n\t"
        "v_mul_f32 %[t2], %[thi], %[t0]\n\t"
        "v_mul_f32 %[t0], %[tlo], %[t0]\n\t"
        "v_cmp_lt_f32 %[RA], |%[t1]|, %[t2]\n\t"
        "v_cmp_ge_f32 %[RN], |%[t1]|, %[t0]\n\t"
        "s_mov_b32 %[fresh], 1\n\t"
        ".Lhave%=:\n\t"
        "s_andn2_b64 %[tm], %[cand], %[RN]\n\t"
        "s_cbranch_scc0 .Lnone%=\n\t"
        "s_ff1_i32_b64 %[c], %[tm]\n\t"
        "s_lshl_b64 %[tm], -2, %[c]\n\t"
        "s_and_b64 %[cand], %[cand], %[tm]\n\t"
        "s_bitcmp1_b64 %[RA], %[c]\n\t"
        "s_cbranch_scc0 .Lamb%=\n\t"
        "s_mov_b32 m0, %[c]\n\t"
        "v_readlane_b32 %[sa], %[cs], %[c]\n\t"
        "v_readlane_b32 %[sb], %[sn], %[c]\n\t"
        "v_writelane_b32 %[seq], %[rs], m0\n\t"
        "s_nop 0\n\t"
        "v_add_f32 %[sx], %[sa], %[sx]\n\t"
        "v_add_f32 %[sy], %[sb], %[sy]\n\t"
        "s_bitset0_b64 %[live], %[c]\n\t"
        "s_add_i32 %[rs], %[rs], 1\n\t"
        "s_mov_b32 %[fresh], 0\n\t"
        "s_cmp_lg_u64 %[cand], 0\n\t"
        "s_cbranch_scc1 .Ltop%=\n\t"
        "s_branch .Ldone%=\n\t"
        ".Lamb%=:\n\t"
        "s_mov_b32 %[cx], %[c]\n\t"
        "s_branch .Ldone%=\n\t"
        ".Lnone%=:\n\t"
        "s_mov_b64 %[cand], 0\n\t"
        ".Ldone%=:\n\t"
        : [cand] "+s"(cand), [live] "+s"(live), [RA] "+s"(RA), [RN] "+s"(RN), [fresh] "+s"(fresh), [rs] "+s"(reg_size), [sx] "+v"(sumdx), [sy] "+v"(sumdy),
          [seq] "+v"(seq), [cx] "=&s"(cx), [c] "=&s"(c), [sa] "=&s"(sa), [sb] "=&s"(sb), [tm] "=&s"(tm), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
        : [cs] "v"(cs), [sn] "v"(sn), [thi] "v"(t_hi), [tlo] "v"(t_lo)
        : "scc", "m0");
    return cx;
}

#ifndef PSL_GROW_ASM_POPS
#define PSL_GROW_ASM_POPS 2   // 2: lsdg_pops2, 1: lsdg_pops, 0: the pop loop as compiled C++ around lsdg_decide (A/B, tools/ab_build.sh)
#endif
// lsdg_decide with the loop over the popped entries around it (round 3): per popped entry the compiler spent ~23 scalar instructions on
// "which lane holds entry i, is it in the window's interior, which of its neighbours are live" (every uniform bool a 64-bit mask, a
// compare, a select and a branch); written out it is 9.  Runs pops and decisions until no lane of the window's interior holds entry i
// (returns -1: the round is over) or a candidate c falls inside the decision margin (returns c with the popped entry's remaining
// candidates in `cand`: the caller runs the reference's arithmetic for c and calls again, which resumes with `cand`).
__device__ __forceinline__ int lsdg_pops(unsigned long long& cand, unsigned long long& live, unsigned long long& RA, unsigned long long& RN, int& fresh,
                                         int& reg_size, int& i, float& sumdx, float& sumdy, int& seq, float cs, float sn, float t_hi, float t_lo,
                                         unsigned long long interior) {
    int cx, c, sa, sb;
    float t0, t1, t2;
    unsigned long long tm;
    const unsigned long long k3x3 = 0x070707ull;
    asm volatile(
        "s_mov_b32 %[cx], -1\n\t"
        "s_cmp_lg_u64 %[cand], 0\n\t"
        "s_cbranch_scc1 .Ltop%=\n\t"
        ".Lpop%=:\n\t"
        "v_cmp_eq_u32 vcc, %[i], %[seq]\n\t"
        "s_and_b64 %[tm], vcc, %[inter]\n\t"
        "s_cbranch_scc0 .Ldone%=\n\t"
        "s_ff1_i32_b64 %[c], %[tm]\n\t"
        "s_add_i32 %[c], %[c], -9\n\t"
        "s_lshl_b64 %[tm], %[k3], %[c]\n\t"
        "s_add_i32 %[i], %[i], 1\n\t"
        "s_and_b64 %[cand], %[tm], %[live]\n\t"
        "s_cbranch_scc0 .Lpop%=\n\t"
        ".Ltop%=:\n\t"
        "s_cmp_lg_u32 %[fresh], 0\n\t"
        "s_cbranch_scc1 .Lhave%=\n\t"
        "v_mul_f32 %[t0], %[sy], %[sn]\n\t"
        "v_mul_f32 %[t1], %[sy], %[cs]\n\t"
        "v_fmac_f32 %[t0], %[sx], %[cs]\n\t"
        "v_fma_f32 %[t1], %[sx], %[sn], -%[t1]\n\t"
        "v_max_f32 %[t0], 0, %[t0]\n\t"
        "v_mul_f32 %[t2], %[thi], %[t0]\n\t"
        "v_mul_f32 %[t0], %[tlo], %[t0]\n\t"
        "v_cmp_lt_f32 %[RA], |%[t1]|, %[t2]\n\t"
        "v_cmp_ge_f32 %[RN], |%[t1]|, %[t0]\n\t"
        "s_mov_b32 %[fresh], 1\n\t"
        ".Lhave%=:\n\t"
        "s_andn2_b64 %[tm], %[cand], %[RN]\n\t"
        "s_cbranch_scc0 .Lnone%=\n\t"
        "s_ff1_i32_b64 %[c], %[tm]\n\t"
        "s_lshl_b64 %[tm], -2, %[c]\n\t"
        "s_and_b64 %[cand], %[cand], %[tm]\n\t"
        "s_bitcmp1_b64 %[RA], %[c]\n\t"
        "s_cbranch_scc0 .Lamb%=\n\t"
        "s_mov_b32 m0, %[c]\n\t"
        "v_readlane_b32 %[sa], %[cs], %[c]\n\t"
        "v_readlane_b32 %[sb], %[sn], %[c]\n\t"
        "v_writelane_b32 %[seq], %[rs], m0\n\t"
        "s_nop 0\n\t"
        "v_add_f32 %[sx], %[sa], %[sx]\n\t"
        "v_add_f32 %[sy], %[sb], %[sy]\n\t"
        "s_bitset0_b64 %[live], %[c]\n\t"
        "s_add_i32 %[rs], %[rs], 1\n\t"
        "s_mov_b32 %[fresh], 0\n\t"
        "s_cmp_lg_u64 %[cand], 0\n\t"
        "s_cbranch_scc1 .Ltop%=\n\t"
        "s_branch .Lpop%=\n\t"
        ".Lamb%=:\n\t"
        "s_mov_b32 %[cx], %[c]\n\t"
        "s_branch .Ldone%=\n\t"
        ".Lnone%=:\n\t"
        "s_mov_b64 %[cand], 0\n\t"
        "s_branch .Lpop%=\n\t"
        ".Ldone%=:\n\t"
        : [cand] "+s"(cand), [live] "+s"(live), [RA] "+s"(RA), [RN] "+s"(RN), [fresh] "+s"(fresh), [rs] "+s"(reg_size), [i] "+s"(i), [sx] "+v"(sumdx),
          [sy] "+v"(sumdy), [seq] "+v"(seq), [cx] "=&s"(cx), [c] "=&s"(c), [sa] "=&s"(sa), [sb] "=&s"(sb), [tm] "=&s"(tm), [t0] "=&v"(t0), [t1] "=&v"(t1),
          [t2] "=&v"(t2)
        : [cs] "v"(cs), [sn] "v"(sn), [thi] "v"(t_hi), [tlo] "v"(t_lo), [inter] "s"(interior), [k3] "s"(k3x3)
        : "scc", "m0", "vcc");
    return cx;
}

// lsdg_pops with packed f32 arithmetic and without the `fresh` flag (round 3, second pass; PSL_GROW_ASM_POPS == 2).  The kernel's time is its
// instruction count (7 waves per SIMD take as long as 8: profiles/r03t_ab_waves.log), and an accepted pixel cost 15 vector + 25 scalar
// instructions.  Now: the sums live in one register pair S = (sum dx, sum dy), the pixel's vector in U = (cos, sin), the thresholds in
// TH = (t_hi, t_lo);   (S.x U.x, S.x U.y)  ->  (S.y U.y + S.x U.x, -S.y U.x + S.x U.y) = (dot, cross)  ->  (t_hi dot, t_lo dot)  is three
// packed instructions, the two compares make five (nine before; max(dot, 0) is not needed: see lsdg_fast_setup), the sums take one
// packed add.  Whether the masks belong to the current sums is known from the place in the code (two copies of the pop sequence) instead of
// a flag that is set, reset and tested.  10 vector + ~21 scalar instructions per accepted pixel.  D, PT, U and the pair the accepted
// vector is read into are bound to registers by name: v_cmp / v_readlane take halves of those pairs, which an operand cannot express.
// Rounding differs from the three-operation form (one fused step); the decision is only taken outside the margin, which is 2e-3 rad
// against errors of ~1e-7 |S|.  Re-entry after the caller's exact test recomputes the masks.
__device__ __forceinline__ int lsdg_pops2(unsigned long long& cand, unsigned long long& live, int& reg_size, int& i, unsigned long long& S, int& seq,
                                          unsigned long long U, unsigned long long TH, unsigned long long interior) {
    int cx, c;
    unsigned long long tm, RA, RN, D, PT, SAB;
    asm volatile(
        "s_mov_b32 %[cx], -1\n\t"
        "s_cmp_lg_u64 %[cand], 0\n\t"
        "s_cbranch_scc1 .Lcomp%=\n\t"
        ".LpopS%=:\n\t"                                   // masks not valid
        "v_cmp_eq_u32 vcc, %[i], %[seq]\n\t"
        "s_and_b64 %[tm], vcc, %[inter]\n\t"
        "s_cbranch_scc0 .Lend%=\n\t"
        "s_ff1_i32_b64 %[c], %[tm]\n\t"
        "s_add_i32 %[c], %[c], -9\n\t"
        "s_lshl_b64 %[tm], 0x70707, %[c]\n\t"
        "s_add_i32 %[i], %[i], 1\n\t"
        "s_and_b64 %[cand], %[tm], %[live]\n\t"
        "s_cbranch_scc0 .LpopS%=\n\t"
        ".Lcomp%=:\n\t"
        "v_pk_mul_f32 %[PT], %[S], %[U] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %[D], %[S], %[U], %[PT] op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]\n\t"
        "v_pk_mul_f32 %[PT], %[TH], %[D] op_sel:[0,0] op_sel_hi:[1,0]\n\t"
        "v_cmp_lt_f32 %[RA], |v57|, v58\n\t"
        "v_cmp_ge_f32 %[RN], |v57|, v59\n\t"
        ".Lhave%=:\n\t"
        "s_andn2_b64 %[tm], %[cand], %[RN]\n\t"
        "s_cbranch_scc0 .LpopF%=\n\t"
        "s_ff1_i32_b64 %[c], %[tm]\n\t"
        "s_lshl_b64 %[tm], -2, %[c]\n\t"
        "s_and_b64 %[cand], %[cand], %[tm]\n\t"
        "s_bitcmp1_b64 %[RA], %[c]\n\t"
        "s_cbranch_scc0 .Lamb%=\n\t"
        "s_mov_b32 m0, %[c]\n\t"
        "v_readlane_b32 s68, v54, %[c]\n\t"
        "v_readlane_b32 s69, v55, %[c]\n\t"
        "v_writelane_b32 %[seq], %[rs], m0\n\t"
        "s_bitset0_b64 %[live], %[c]\n\t"
        "s_add_i32 %[rs], %[rs], 1\n\t"
        "s_cmp_lg_u64 %[cand], 0\n\t"
        "v_pk_add_f32 %[S], %[S], s[68:69]\n\t"
        "s_cbranch_scc1 .Lcomp%=\n\t"
        "s_branch .LpopS%=\n\t"
        ".LpopF%=:\n\t"                                   // masks valid: the last candidates were rejected, the sums have not changed
        "v_cmp_eq_u32 vcc, %[i], %[seq]\n\t"
        "s_and_b64 %[tm], vcc, %[inter]\n\t"
        "s_cbranch_scc0 .Lend%=\n\t"
        "s_ff1_i32_b64 %[c], %[tm]\n\t"
        "s_add_i32 %[c], %[c], -9\n\t"
        "s_lshl_b64 %[tm], 0x70707, %[c]\n\t"
        "s_add_i32 %[i], %[i], 1\n\t"
        "s_and_b64 %[cand], %[tm], %[live]\n\t"
        "s_cbranch_scc0 .LpopF%=\n\t"
        "s_branch .Lhave%=\n\t"
        ".Lamb%=:\n\t"
        "s_mov_b32 %[cx], %[c]\n\t"
        "s_branch .Ldone%=\n\t"
        ".Lend%=:\n\t"
        "s_mov_b64 %[cand], 0\n\t"
        ".Ldone%=:\n\t"
        : [cand] "+s"(cand), [live] "+s"(live), [rs] "+s"(reg_size), [i] "+s"(i), [S] "+v"(S), [seq] "+v"(seq), [cx] "=&s"(cx), [c] "=&s"(c), [tm] "=&s"(tm),
          [RA] "=&s"(RA), [RN] "=&s"(RN), [D] "=&{v[56:57]}"(D), [PT] "=&{v[58:59]}"(PT), [SAB] "=&{s[68:69]}"(SAB)
        : [U] "{v[54:55]}"(U), [TH] "v"(TH), [inter] "s"(interior)
        : "scc", "m0", "vcc");
    return cx;
}

// `regrow`: the refinement's second growth - its releases of the region's pixels must have completed, nothing is pending.
// `touched`: set when the region took a pixel whose raster index lies in (seed, trip_end): only then has the seed scan to look
// at the `used` words of its current 256-pixel trip again.
template <int LU>
__device__ int lsdg_region_grow4(const LsdW& F, int sx, int sy, double* reg_angle_out, double prec, const LsdgFast fc, LsdgPend& pd,
                                 bool regrow, int trip_end, bool& touched) {
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32* ring = (lds_u32*)F.ring;
    lds_u32* map = (lds_u32*)F.map;
    if (regrow) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pd.PA = 0ull;
    }
    const int lane = F.lane, lx = lane & 7, ly = lane >> 3;
    const int addr0 = sx + sy * F.W;
    int reg_size = 1;
    const uint32_t xy0 = (uint32_t)sx | ((uint32_t)sy << 16);
    if (lane == 0) {   // the seed: queue entry 0 and its mark (issued in front of the first window's load: it has landed when that load returns)
        ring[0] = xy0;
        lsdg_mark<LU>(F, addr0, 1);
    }
    float reg_deg = F.ang[addr0];       // these two loads and the first window's are in flight together
    const float2 t0 = F.seedt[addr0];
    float sumdx = t0.x, sumdy = t0.y;
    unsigned long long tch = 0ull;
#ifdef PSL_GROW_STATS
    unsigned gs[16] = {0};
#endif
    GS(0);
    int rs_angle = 1;  // the region size reg_deg belongs to (the seed's own angle at 1)
    unsigned long long PA = pd.PA;
    int pox = pd.pox, poy = pd.poy;
    int i = 0;
    while (i < reg_size) {
        const int rs0 = reg_size;
        GS(1);
        const int qi = i + lane;
        uint32_t q = 0xffffffffu;
        if (reg_size - i > PSL_LSD_HALF) {  // uniform: the frontier lags more than half a ring behind the queue's end: the entries in front of
            // the last half ring have been flushed to HBM and are mapped from there (the younger ones are reached in a later round)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the flush may be the previous round's
            if (qi < reg_size - PSL_LSD_HALF) {
                const uint32_t g = F.reg[qi];
                asm volatile("v_mov_b32 %0, %1" : "=v"(q) : "v"(g));
            }
        } else if (qi < reg_size) {
            q = ring[qi & (PSL_LSD_RING - 1)];
        }
        const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);   // entry i (the first round: the seed, from ring[0])
        const int ox = (int)(e & 0xffff) - 3, oy = (int)(e >> 16) - 1;
        const int x = ox + lx, y = oy + ly;
        const bool inside = (unsigned)x < (unsigned)F.W && (unsigned)y < (unsigned)F.H;
        const int cidx = inside ? lsdw_pix(F, x, y) : addr0;
#ifdef PSL_GROW_STATS
        const unsigned long long st_l0 = __builtin_amdgcn_s_memtime();
#endif
        const float2 t = F.trig[cidx];
        const bool ub = lsdg_used<LU>(F, cidx);
        // queue entry -> lane of the window (while the load is in flight)
        const int qx = (int)(q & 0xffff) - ox, qy = (int)(q >> 16) - oy;
        map[lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        if ((unsigned)qx < 8u && (unsigned)qy < 8u) map[qy * 8 + qx] = (uint32_t)(qi + 1);
        __builtin_amdgcn_wave_barrier();
        int seq = (int)map[lane] - 1;
        const int px = x - pox, py = y - poy;
        const bool pa = LU ? false : ((unsigned)px < 8u && (unsigned)py < 8u && ((PA >> (py * 8 + px)) & 1ull) != 0ull);   // (marks in LDS are never "on their way")
        const uint32_t xy = (uint32_t)x | ((uint32_t)y << 16);
#ifdef PSL_GROW_STATS
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // [8]: cycles from the window's loads to their data (with the LDS map work in their shadow)
        if (blockIdx.x == 0 && lane == 0) g_gstats[8] += __builtin_amdgcn_s_memtime() - st_l0;
#endif
        const float cs = t.x, sn = t.y;
        // a lane that holds a queue entry is a pixel of the region (the seed among them: its mark is not in memory yet)
        unsigned long long live = __ballot((int)inside & ((int)(cs != 0.f) | (int)(sn != 0.f)) & (int)!ub & (int)!pa & (int)(seq < 0));   // (no short circuit: one straight run of compares)
        unsigned long long RA = 0ull, RN = 0ull;  // valid while the sums are the ones they were computed from (`fresh`)
        int fresh = 0;
#if defined(PSL_GROW_STATS) || !PSL_GROW_ASM_POPS   // the same loop as the compiler writes it (diagnostic counters; A/B)
        for (;;) {
            const unsigned long long m = __ballot(seq == i) & F.interior;  // in the window, and in its 6 x 6 interior
            if (!m) break;  // (also when i == reg_size: no lane holds an index that does not exist yet)
            const int sh = __ffsll((long long)m) - 10;
            unsigned long long cand = (0x070707ull << sh) & live;  // the 3 x 3 neighbours, ascending bit = visiting order
            ++i;
            GS(2);
            if (cand) GS(3);
            while (cand) {
                GS(4);
                const int c = lsdg_decide(cand, live, RA, RN, fresh, reg_size, sumdx, sumdy, seq, cs, sn, fc.t_hi, fc.t_lo);
                if (c < 0) break;
                // lane c is within the margin of the threshold: the reference's arithmetic
                GS(6);
                if (rs_angle != reg_size) { reg_deg = psl_fast_atan2(sumdy, sumdx); rs_angle = reg_size; }
                const double ad = PSL_DMUL((double)F.ang[cidx], PSL_DEG2RAD), th = PSL_DMUL((double)reg_deg, PSL_DEG2RAD);
                if (!((__ballot(lsdg_aligned(ad, th, prec)) >> c) & 1ull)) continue;
                sumdx = PSL_FADD(sumdx, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs), c)));
                sumdy = PSL_FADD(sumdy, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sn), c)));
                if (lane == c) seq = reg_size;
                live &= ~(1ull << c);
                ++reg_size;
                fresh = 0;
            }
        }
#elif PSL_GROW_ASM_POPS == 2
        unsigned long long cand = 0ull;
        unsigned long long S = lsdg_pack2(sumdx, sumdy);
        const unsigned long long U = lsdg_pack2(cs, sn);
        for (;;) {
            const int c = lsdg_pops2(cand, live, reg_size, i, S, seq, U, fc.th2, F.interior);
            if (c < 0) break;   // no lane of the window's interior holds entry i: the round is over
            // lane c is within the margin of the threshold: the reference's arithmetic
            if (rs_angle != reg_size) { reg_deg = psl_fast_atan2(lsdg_hi(S), lsdg_lo(S)); rs_angle = reg_size; }
            const double ad = PSL_DMUL((double)F.ang[cidx], PSL_DEG2RAD), th = PSL_DMUL((double)reg_deg, PSL_DEG2RAD);
            if (!((__ballot(lsdg_aligned(ad, th, prec)) >> c) & 1ull)) continue;
            S = lsdg_pack2(PSL_FADD(lsdg_lo(S), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs), c))),
                           PSL_FADD(lsdg_hi(S), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sn), c))));
            if (lane == c) seq = reg_size;
            live &= ~(1ull << c);
            ++reg_size;
        }
        sumdx = lsdg_lo(S); sumdy = lsdg_hi(S);
        (void)RA; (void)RN; (void)fresh;
#else
        unsigned long long cand = 0ull;
        for (;;) {
            const int c = lsdg_pops(cand, live, RA, RN, fresh, reg_size, i, sumdx, sumdy, seq, cs, sn, fc.t_hi, fc.t_lo, F.interior);
            if (c < 0) break;   // no lane of the window's interior holds entry i: the round is over
            // lane c is within the margin of the threshold: the reference's arithmetic
            if (rs_angle != reg_size) { reg_deg = psl_fast_atan2(sumdy, sumdx); rs_angle = reg_size; }
            const double ad = PSL_DMUL((double)F.ang[cidx], PSL_DEG2RAD), th = PSL_DMUL((double)reg_deg, PSL_DEG2RAD);
            if (!((__ballot(lsdg_aligned(ad, th, prec)) >> c) & 1ull)) continue;
            sumdx = PSL_FADD(sumdx, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs), c)));
            sumdy = PSL_FADD(sumdy, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sn), c)));
            if (lane == c) seq = reg_size;
            live &= ~(1ull << c);
            ++reg_size;
            fresh = 0;
        }
#endif
        // the round's pixels: queue entries and marks, one store instruction each
        const bool mine = seq >= rs0;
        const unsigned long long acc_round = __ballot(mine);
        if (mine) {
            ring[seq & (PSL_LSD_RING - 1)] = xy;
            lsdg_mark<LU>(F, cidx, 1);
        }
        tch |= __ballot(mine && (unsigned)(cidx - addr0 - 1) < (unsigned)(trip_end - addr0 - 1));   // addr0 < cidx < trip_end (the seed lies in its trip)
        if ((unsigned)reg_size / PSL_LSD_HALF != (unsigned)rs0 / PSL_LSD_HALF) {  // half a ring of entries is complete: to HBM, long before the ring wraps over it
            __builtin_amdgcn_wave_barrier();
            const int b0 = (int)((unsigned)reg_size / PSL_LSD_HALF - 1u) * PSL_LSD_HALF;
#pragma unroll
            for (int k = 0; k < PSL_LSD_HALF / 64; ++k) F.reg[b0 + k * 64 + lane] = ring[(b0 + k * 64 + lane) & (PSL_LSD_RING - 1)];
        }
        PA = acc_round; pox = ox; poy = oy;
    }
    if (tch) touched = true;
    pd.PA = PA; pd.pox = pox; pd.poy = poy;
    if (rs_angle != reg_size) reg_deg = psl_fast_atan2(sumdy, sumdx);
#ifdef PSL_GROW_STATS
    if (blockIdx.x == 0 && lane == 0) for (int k = 0; k < 16; ++k) g_gstats[k] += gs[k];
#endif
    *reg_angle_out = PSL_DMUL((double)reg_deg, PSL_DEG2RAD);
    return reg_size;
}

__device__ __forceinline__ double lsdw_wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double u = __shfl_xor(v, o); v = u > v ? u : v; }
    return v;
}
__device__ __forceinline__ double lsdw_wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double u = __shfl_xor(v, o); v = u < v ? u : v; }
    return v;
}

// "Terms in parallel, sums in series", three sums at a time: the 64 terms of each of three sums are staged in LDS rows
// (F.term[r * 64 + t]); lane r (r = 0, 1, 2) then adds ITS row strictly in index order, so one dependent f64 add per element
// serves all three sums (every lane executes the adds anyway; lanes >= 3 repeat lane 0).  Rows are padded with +0.0 to the
// block: x + (+0.0) == x for every x a running sum that starts at +0.0 can hold (it is never -0.0), and a difference is
// staged negated (x - t == x + (-t) in IEEE arithmetic), so the padded sums are bit-identical to the reference's loops.
typedef __attribute__((address_space(3))) double lds_f64;
__device__ __forceinline__ double lsdw_sum_rows(const LsdW& F, double acc, int cnt) {
    typedef double __attribute__((ext_vector_type(2))) f64x2;
    typedef __attribute__((address_space(3))) f64x2 lds_f64x2;
    const lds_f64x2* my = (const lds_f64x2*)((const lds_f64*)F.term + (F.lane < 3 ? F.lane : 0) * 64);   // two terms per LDS instruction (s_term is 16-byte aligned)
    for (int t = 0; t < cnt; t += 8) {
        f64x2 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = my[(t >> 1) + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc = PSL_DADD(acc, v[k].x); acc = PSL_DADD(acc, v[k].y); }
    }
    return acc;
}
__device__ __forceinline__ double lsdw_lane_f64(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ void lsdw_region2rect(const LsdW& F, int reg_size, double reg_angle, double prec, LsdRect* rec) {
#if PSL_GROW_DIAG == 4 || PSL_GROW_DIAG == 5
  for (int diag_rep = 0; diag_rep < (PSL_GROW_DIAG == 4 ? 2 : 1); ++diag_rep) {
    asm volatile("" ::: "memory");
#endif
    double acc = 0;  // lane 0: sum x w, lane 1: sum y w, lane 2: sum w
    for (int base = 0; base < reg_size; base += 64) {
        const int j = base + F.lane, cnt = min(64, reg_size - base);
        double t0 = 0, t1 = 0, t2 = 0;
        if (j < reg_size) {
            const uint32_t rp = lsdw_reg(F, j, reg_size);
            const int px = (int)(rp & 0xffff), py = (int)(rp >> 16);
            const double w = F.mod[(uint32_t)lsdw_pix(F, px, py)];
            t0 = PSL_DMUL((double)px, w); t1 = PSL_DMUL((double)py, w); t2 = w;
        }
        F.term[F.lane] = t0; F.term[64 + F.lane] = t1; F.term[128 + F.lane] = t2;
        __builtin_amdgcn_wave_barrier();
        acc = lsdw_sum_rows(F, acc, cnt);
        __builtin_amdgcn_wave_barrier();
    }
    double x = lsdw_lane_f64(acc, 0), y = lsdw_lane_f64(acc, 1);
    const double sum = lsdw_lane_f64(acc, 2);
    x = x / sum; y = y / sum;
    acc = 0;  // lane 0: Ixx, lane 1: Iyy, lane 2: Ixy
    for (int base = 0; base < reg_size; base += 64) {
        const int j = base + F.lane, cnt = min(64, reg_size - base);
        double t0 = 0, t1 = 0, t2 = 0;
        if (j < reg_size) {
            const uint32_t rp = lsdw_reg(F, j, reg_size);
            const int px = (int)(rp & 0xffff), py = (int)(rp >> 16);
            const double w = F.mod[(uint32_t)lsdw_pix(F, px, py)];
            const double dx = PSL_DSUB((double)px, x), dy = PSL_DSUB((double)py, y);
            t0 = PSL_DMUL(PSL_DMUL(dy, dy), w); t1 = PSL_DMUL(PSL_DMUL(dx, dx), w);
            t2 = -PSL_DMUL(PSL_DMUL(dx, dy), w);  // Ixy -= dx dy w
        }
        F.term[F.lane] = t0; F.term[64 + F.lane] = t1; F.term[128 + F.lane] = t2;
        __builtin_amdgcn_wave_barrier();
        acc = lsdw_sum_rows(F, acc, cnt);
        __builtin_amdgcn_wave_barrier();
    }
    const double Ixx = lsdw_lane_f64(acc, 0), Iyy = lsdw_lane_f64(acc, 1), Ixy = lsdw_lane_f64(acc, 2);
    const double dI = PSL_DSUB(Ixx, Iyy);
    const double lambda = PSL_DMUL(0.5, PSL_DSUB(PSL_DADD(Ixx, Iyy), __dsqrt_rn(PSL_DADD(PSL_DMUL(dI, dI), PSL_DMUL(PSL_DMUL(4.0, Ixy), Ixy)))));
    double theta = (fabs(Ixx) > fabs(Iyy)) ? (double)psl_fast_atan2((float)PSL_DSUB(lambda, Ixx), (float)Ixy)
                                           : (double)psl_fast_atan2((float)Ixy, (float)PSL_DSUB(lambda, Iyy));
    theta = PSL_DMUL(theta, PSL_DEG2RAD);
    if (fabs(psl_angle_diff_signed(theta, reg_angle)) > prec) theta = PSL_DADD(theta, PSL_PI);
    // cos / sin of theta in [0, 3 pi) exactly as glibc's (psl_sincos_glibc.h): they feed the rectangle's f64 arithmetic, whose results
    // are truncated to int by the NFA's pixel scan
    const double dx = psl_glibc_cos(theta, F.sctab), dy = psl_glibc_sin(theta, F.sctab);
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;  // order-independent: max(0, max l), min(0, min l)
    for (int j = F.lane; j < reg_size; j += 64) {
        const uint32_t rp = lsdw_reg(F, j, reg_size);
        const double rdx = PSL_DSUB((double)(int)(rp & 0xffff), x), rdy = PSL_DSUB((double)(int)(rp >> 16), y);
        const double l = PSL_DADD(PSL_DMUL(rdx, dx), PSL_DMUL(rdy, dy));
        const double w = PSL_DADD(PSL_DMUL(-rdx, dy), PSL_DMUL(rdy, dx));
        l_max = l > l_max ? l : l_max; l_min = l < l_min ? l : l_min;
        w_max = w > w_max ? w : w_max; w_min = w < w_min ? w : w_min;
    }
    l_max = lsdw_wave_max(l_max); l_min = lsdw_wave_min(l_min);
    w_max = lsdw_wave_max(w_max); w_min = lsdw_wave_min(w_min);
    rec->x1 = PSL_DADD(x, PSL_DMUL(l_min, dx)); rec->y1 = PSL_DADD(y, PSL_DMUL(l_min, dy));
    rec->x2 = PSL_DADD(x, PSL_DMUL(l_max, dx)); rec->y2 = PSL_DADD(y, PSL_DMUL(l_max, dy));
    rec->width = PSL_DSUB(w_max, w_min);
    if (rec->width < 1.0) rec->width = 1.0;
    rec->theta = theta; rec->dx = dx; rec->dy = dy;  // read by the NFA validation (LSD_REFINE_ADV) only
#if PSL_GROW_DIAG == 4 || PSL_GROW_DIAG == 5
  }
#endif
}

template <int LU>
__device__ int lsdw_refine(const LsdW& F, int reg_size, double reg_angle, double prec, LsdRect* rec, double density_th, LsdgPend& pd, int trip_end,
                           bool& touched) {
    double density = psl_lsd_density(reg_size, *rec);
    if (density >= density_th) return reg_size;
    const uint32_t r0 = lsdw_reg(F, 0, reg_size);
    const int x0 = (int)(r0 & 0xffff), y0 = (int)(r0 >> 16);
    const double xc = (double)x0, yc = (double)y0;
    const double ang_c = PSL_DMUL((double)F.ang[x0 + y0 * F.W], PSL_DEG2RAD);
    double acc = 0;  // lane 0: sum of the angle differences, lane 1: sum of their squares (pixels outside the radius stage +0.0)
    int n = 0;
#if PSL_GROW_DIAG == 5
  for (int diag_rep = 0; diag_rep < 2; ++diag_rep) {
    asm volatile("" ::: "memory");
    acc = 0; n = 0;
#endif
    for (int base = 0; base < reg_size; base += 64) {
        const int j = base + F.lane, cnt = min(64, reg_size - base);
        double ang_d = 0;
        bool in = false;
        if (j < reg_size) {
            const uint32_t rp = lsdw_reg(F, j, reg_size);
            const int px = (int)(rp & 0xffff), py = (int)(rp >> 16), a = lsdw_pix(F, px, py);
            lsdg_mark<LU>(F, a, 0);
            if (__dsqrt_rn(psl_dist_sq(xc, yc, (double)px, (double)py)) < rec->width) {
                in = true;
                ang_d = psl_angle_diff_signed(PSL_DMUL((double)F.ang[a], PSL_DEG2RAD), ang_c);
            }
        }
        F.term[F.lane] = in ? ang_d : 0.0;
        F.term[64 + F.lane] = in ? PSL_DMUL(ang_d, ang_d) : 0.0;
        n += __popcll(__ballot(in));
        __builtin_amdgcn_wave_barrier();
        acc = lsdw_sum_rows(F, acc, cnt);
        __builtin_amdgcn_wave_barrier();
    }
#if PSL_GROW_DIAG == 5
  }
#endif
    const double sum = lsdw_lane_f64(acc, 0), s_sum = lsdw_lane_f64(acc, 1);
    const double mean_angle = sum / (double)n;
    const double tau = PSL_DMUL(2.0, __dsqrt_rn(PSL_DADD(PSL_DSUB(s_sum, PSL_DMUL(PSL_DMUL(2.0, mean_angle), sum)) / (double)n, PSL_DMUL(mean_angle, mean_angle))));
    reg_size = lsdg_region_grow4<LU>(F, x0, y0, &reg_angle, tau, lsdg_fast_setup(tau), pd, true, trip_end, touched);
    if (reg_size < 2) return 0;
    lsdw_region2rect(F, reg_size, reg_angle, prec, rec);
    density = psl_lsd_density(reg_size, *rec);
    if (density >= density_th) return reg_size;
    // reduce_region_radius.  The reference walks the region's list once per radius step and removes a pixel outside the radius by
    // swapping the list's LAST element into its place (and looks at that one next); the order it leaves defines the order of the later
    // sums, so it has to be reproduced - but not by walking: with m = the pixels that stay, every stayer in front of position m keeps
    // its place, and the k-th hole in front of m (ascending) receives the k-th stayer found behind m going DOWN from the end (the
    // elements the walk swaps to the front and removes right away are passed over by that descent).  Three coalesced passes per radius
    // step instead of one dependent memory round trip per listed pixel: count and release | stayers behind m, ranked from the end, to a
    // scratch list behind the queue | holes in front of m filled by rank.  Dense scenes spend most of the chain here if it is walked
    // (12288 frames of the 'sticks' scene: 80 k list entries walked per frame, k_lsd_grow4 211 ms; profiles/r03q_grow_parts.log).
    // A step that removes nothing leaves rectangle and density as they are: no region2rect for it.
    const double d1 = psl_dist_sq(xc, yc, rec->x1, rec->y1), d2 = psl_dist_sq(xc, yc, rec->x2, rec->y2);
    double radSq = d1 > d2 ? d1 : d2;
    for (int j = F.lane; j < reg_size; j += 64) F.reg[j] = lsdw_reg(F, j, reg_size);  // make HBM copy authoritative
    __builtin_amdgcn_wave_barrier();
    const unsigned long long lt = (1ull << F.lane) - 1ull;
    while (density < density_th) {
        radSq = PSL_DMUL(radSq, 0.75 * 0.75);
        const int n = reg_size;
#if !PSL_REDUCE_SERIAL
        int m = 0;
#if PSL_GROW_DIAG == 5
      for (int diag_rep = 0; diag_rep < 2; ++diag_rep) {
        asm volatile("" ::: "memory");
        m = 0;
#endif
        for (int base = 0; base < n; base += 64) {
            const int j = base + F.lane;
            bool in = false;
            if (j < n) {
                const uint32_t rp = F.reg[j];
                const int px = (int)(rp & 0xffff), py = (int)(rp >> 16);
                in = !(psl_dist_sq(xc, yc, (double)px, (double)py) > radSq);
                if (!in) lsdg_mark<LU>(F, lsdw_pix(F, px, py), 0);
            }
            m += __popcll(__ballot(in));
        }
#if PSL_GROW_DIAG == 5
      }
#endif
        if (m == n) continue;
        if (m >= 2 && (size_t)n + (size_t)(n - m) <= (size_t)F.W * F.H) {
            uint32_t* S = F.reg + n;   // scratch: the stayers behind m, last first (at most n - m of them)
            int fr = 0;
#if PSL_GROW_DIAG == 5
          for (int diag_rep = 0; diag_rep < 2; ++diag_rep) {
            asm volatile("" ::: "memory");
            fr = 0;
#endif
            for (int e = n; e > m; e -= 64) {
                const int q = e - 1 - F.lane;
                bool in = false;
                uint32_t rp = 0;
                if (q >= m) {
                    rp = F.reg[q];
                    in = !(psl_dist_sq(xc, yc, (double)(int)(rp & 0xffff), (double)(int)(rp >> 16)) > radSq);
                }
                const unsigned long long b = __ballot(in);
                if (in) S[fr + __popcll(b & lt)] = rp;
                fr += __popcll(b);
            }
#if PSL_GROW_DIAG == 5
          }
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int hr = 0;
            for (int base = 0; base < m; base += 64) {
                const int j = base + F.lane;
                bool out = false;
                if (j < m) {
                    const uint32_t rp = F.reg[j];
                    out = psl_dist_sq(xc, yc, (double)(int)(rp & 0xffff), (double)(int)(rp >> 16)) > radSq;
                }
                const unsigned long long b = __ballot(out);
                if (out) F.reg[j] = S[hr + __popcll(b & lt)];
                hr += __popcll(b);
            }
            reg_size = m;
        } else if (m < 2) {
            reg_size = m;   // (the marks are released; what the list holds no longer matters)
        } else
#endif
        {   // the walk as written (no room for the scratch list behind a queue of more than two thirds of the image; or -DPSL_REDUCE_SERIAL=1)
            for (int i = 0; i < reg_size; ++i) {
                const uint32_t rp = F.reg[i];
                const int px = (int)(rp & 0xffff), py = (int)(rp >> 16);
                if (psl_dist_sq(xc, yc, (double)px, (double)py) > radSq) {
                    const int a = px + py * F.W;
                    const uint32_t last = F.reg[reg_size - 1];
                    if (F.lane == 0) {
                        lsdg_mark<LU>(F, a, 0);
                        F.reg[i] = last; F.reg[reg_size - 1] = rp;
                    }
                    __builtin_amdgcn_wave_barrier();
                    --reg_size;
                    --i;
                }
            }
        }
        if (reg_size < 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the releases above have to land before the next region looks
            pd.PA = 0ull;
            return 0;
        }
        __builtin_amdgcn_wave_barrier();
        for (int j = F.lane; j < min(reg_size, PSL_LSD_RING); j += 64) {  // ring must mirror the last RING entries
            const int idx = reg_size - 1 - j;
            F.ring[idx & (PSL_LSD_RING - 1)] = F.reg[idx];
        }
        __builtin_amdgcn_wave_barrier();
        lsdw_region2rect(F, reg_size, reg_angle, prec, rec);
        density = psl_lsd_density(reg_size, *rec);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pd.PA = 0ull;
    return reg_size;
}

#define PSL_LSD_RECT_F64 12  // doubles per rectangle record handed to the NFA kernels: x1 y1 x2 y2 width theta dx dy (k_lsd_grow4) | prec p log_nfa pad

// flsd()'s output step (+0.5, / SCALE, to float) followed by the contrib wrapper's checkLineExtremes
// (Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:111-138)
__device__ __forceinline__ void psl_lsd_store_segment(const LineParams& P, double x1, double y1, double x2, double y2, float* out) {
    float e[4] = {(float)(PSL_DADD(x1, 0.5) / 0.8), (float)(PSL_DADD(y1, 0.5) / 0.8), (float)(PSL_DADD(x2, 0.5) / 0.8), (float)(PSL_DADD(y2, 0.5) / 0.8)};
    if (e[0] < 0) e[0] = 0;
    if (e[0] >= P.w) e[0] = (float)P.w - 1.0f;
    if (e[2] < 0) e[2] = 0;
    if (e[2] >= P.w) e[2] = (float)P.w - 1.0f;
    if (e[1] < 0) e[1] = 0;
    if (e[1] >= P.h) e[1] = (float)P.h - 1.0f;
    if (e[3] < 0) e[3] = 0;
    if (e[3] >= P.h) e[3] = (float)P.h - 1.0f;
    out[0] = e[0]; out[1] = e[1]; out[2] = e[2]; out[3] = e[3];
}

#ifndef PSL_GROW_DIAG
#define PSL_GROW_DIAG 0
#endif
#ifndef PSL_GROW_WAVES
#define PSL_GROW_WAVES 8   // waves per SIMD.  Round 2 (1024-entry ring, 12288 struct frames, tools/occ_sweep.sh): 5: 60.5 ms, 6: 52.1 ms, 7: 53.4 ms - but 5.9 KB of
                           // LDS per wave capped the CU at 27 waves, so 7 and 8 never ran.  Round 3, 512-entry ring (3.8 KB per wave, 32 waves per CU), 12288 frames of
                           // the dense scene, A/B in one session (tools/ab_round3.sh, profiles/r03c_ab_waves.log): 6: 261.3 ms, 7: 246.3 ms, 8: 242.8 ms (64 VGPRs, 36
                           // spilled to scratch, 133 SGPR spills: the kernel is bound by instruction issue, not by the waves in flight; -7 %).
                           // Without a bound the kernel takes 105 VGPRs (4 waves): 63.1 ms on the struct scene
#endif
// HELPERS: launches of a few frames (one workgroup per XCD at most) run the chain on wave 0 and let HELPERS more waves of the same
// workgroup - hence the same XCD's L2 - read one dword of every 64-byte piece of the frame's neighbour records, angles and `used` map
// (13 bytes per pixel: 2.6 MB of the 4 MB L2 at 640x480), top of the frame first: the chain's ~6 000 dependent round trips then
// end in L2 instead of HBM (one frame: 12.9 -> 11.9 ms).  The helpers stay at most 2.5 MB in front of the seed scan, whose position
// wave 0 publishes in LDS, so that a larger frame (1280x960: 10 MB) keeps a band in front of the scan warm instead of flushing the
// cache (all of it at once: 25.2 ms against 24.6 ms without helpers).  No effect on results.  Not used with thousands of frames in
// flight: no L2 to spare and no idle wave slot.  Seed vectors and magnitudes prefetched at defined pixels as well: no further gain.
template <int HELPERS, int LU>
__global__ __launch_bounds__(64 * (1 + HELPERS), HELPERS ? 1 : PSL_GROW_WAVES) void k_lsd_grow4(LineParams P, const float* __restrict__ angdeg, const double* __restrict__ modgrad,
                                                   const float2* __restrict__ trig, uint8_t* __restrict__ used, const float2* __restrict__ seedt, uint32_t* __restrict__ reg,
                                                   float* __restrict__ seg, int* __restrict__ nseg, double* __restrict__ rects, int nframes, const int* __restrict__ order) {
    __shared__ uint32_t s_ring[PSL_LSD_RING];
    __shared__ __attribute__((aligned(16))) double s_term[3 * 64];
    __shared__ uint32_t s_map[64];
    const int frame = order ? order[blockIdx.x] : (int)blockIdx.x, lane = threadIdx.x;   // many-frames launches: heaviest frames first (k_frame_order)
    const size_t npx = (size_t)P.W * P.H;
    const int words = (int)((npx + 31) >> 5);
    LsdW F;
    F.W = P.W; F.H = P.H; F.lane = lane;
    F.ang = angdeg + frame * npx; F.mod = modgrad + frame * npx; F.trig = trig + frame * npx; F.used = used + frame * npx; F.reg = reg + frame * npx;
    F.seedt = seedt + frame * npx; F.ring = s_ring; F.term = s_term; F.sctab = P.sctab; F.map = s_map;
    extern __shared__ uint32_t s_ubits[];   // LU: one "used" bit per scaled pixel (the launch passes 4 * words bytes)
    F.ubits = LU ? s_ubits : nullptr;
    __shared__ int s_scan_unit;  // HELPERS: the 64-pixel unit the seed scan has reached (written by wave 0, polled by the helpers)
    static_assert(!LU || HELPERS, "the LDS map needs a workgroup per frame that has the CU to itself");
    if (HELPERS) {
        if (threadIdx.x == 0) s_scan_unit = 0;
        if (LU) for (int k = threadIdx.x; k < words; k += 64 * (1 + HELPERS)) s_ubits[k] = 0u;
        __syncthreads();
    }
    if (HELPERS && threadIdx.x >= 64) {
        const char* pt = (const char*)F.trig;
        const char* pa = (const char*)F.ang;
        const char* pu = (const char*)F.used;
        const int units = (int)(npx >> 6);  // 64 pixels: 8 pieces of records, 4 of angles, 1 of the map (LU: the map in memory is read once per 256-pixel trip of the scan; its piece is still warmed)
        const int ahead = (int)(((size_t)5 << 19) / (64 * 13)) / ((nframes + 7) / 8);  // all helpers of an XCD together stay at most 2.5 MB (of its 4 MB L2) in front of their scans
        // The loads are never waited for individually, so their destination must stay reserved until the final wait: ONE register,
        // read-write operand of every load and consumed after the wait (an output-only operand is free for reuse - as the next
        // address, say - while the load is still in flight).
        uint32_t sink = 0;
        int j = (int)threadIdx.x - 64;
        while (j < units * 13) {
            const int lim = min(units, *(volatile int*)&s_scan_unit + ahead) * 13;
            if (j >= lim) { __builtin_amdgcn_s_sleep(64); continue; }  // the scan's last update is `units`: every helper gets to the end
            for (; j < lim; j += 64 * HELPERS) {
                const int u = j / 13, k = j - u * 13;
                const char* q = k < 8 ? pt + (size_t)u * 512 + k * 64 : (k < 12 ? pa + (size_t)u * 256 + (k - 8) * 64 : pu + (size_t)u * 64);
                asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(q) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) : : "memory");
        return;
    }
    F.interior = __ballot((lane & 7) >= 1 && (lane & 7) <= 6 && (lane >> 3) >= 1 && (lane >> 3) <= 6);
    const LsdgFast fcP = lsdg_fast_setup(P.prec);
    LsdgPend pd;
    pd.PA = 0ull; pd.pox = 0; pd.poy = 0;
#ifdef PSL_GROW_STATS
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // (the used flags start at 0: k_lsd_grad has just written the records)
    (void)words;
    float* out = seg + (size_t)frame * P.maxseg * 4;
    int count = 0;
    const int scan_end = (P.H - 1) * P.W;
    int trip_x = 0, trip_y = 0;  // (x, y) of pixel `base`, kept without divisions
    for (int base = 0; base < scan_end; base += 256, trip_x += 256) {
        while (trip_x >= P.W) { trip_x -= P.W; ++trip_y; }
        // four 64-pixel rows per trip, their loads in flight together; what is kept of them is wave-uniform: the masks of
        // pixels with a defined angle and of used pixels.  The seed loop below exists ONCE (not per row): the kernel's code is
        // dominated by the inlined region growing, and four copies of it did not fit the instruction cache.
        unsigned long long dm[4], um[4], sm[4];
        const int trip_end = min(base + 256, scan_end);
        if (HELPERS && lane == 0) *(volatile int*)&s_scan_unit = base >> 6;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // marks of the last region's last round
        {
            float a4[4];
            uint32_t u4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ad = base + q * 64 + lane;
                a4[q] = ad < scan_end ? F.ang[ad] : PSL_LSD_NOTDEF;
                u4[q] = ad < scan_end ? lsdg_state<LU>(F, ad) : 1u;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                dm[q] = __ballot(a4[q] != PSL_LSD_NOTDEF);  // column W-1 and row H-1 are NOTDEF by construction
                um[q] = __ballot(u4[q] == 1u);
                sm[q] = P.min_reg_size > 1 ? __ballot(u4[q] == 2u) : 0ull;  // static singletons (their flag never changes; taken ones drop out through um)
            }
        }
        // The bitmap changes only through the regions grown here: after each processed seed the remaining candidates
        // of the 64-pixel row are reconciled with ONE bitmap read per lane (not one read + round trip per candidate),
        // and the rows further down the chunk are re-read when their turn comes.  A pixel a region took and its
        // refinement released again is still a candidate, as in the reference's raster scan.
        bool stale = false;
#pragma nounroll
        for (int q = 0; q < 4; ++q) {
            const unsigned long long dmq = q == 0 ? dm[0] : q == 1 ? dm[1] : q == 2 ? dm[2] : dm[3];
            unsigned long long umq = q == 0 ? um[0] : q == 1 ? um[1] : q == 2 ? um[2] : um[3];
            const unsigned long long smq = q == 0 ? sm[0] : q == 1 ? sm[1] : q == 2 ? sm[2] : sm[3];
            if (!dmq) continue;
            const int ad = base + q * 64 + lane;
            if (stale) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                umq = __ballot(ad < scan_end ? lsdg_used<LU>(F, ad) : true);
            }
            unsigned long long mask = dmq & ~umq;
            bool dirty = false;
            while (mask) {
                if (dirty) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    mask &= __ballot(ad < scan_end && !lsdg_used<LU>(F, ad));
                    dirty = false;
                    if (!mask) break;
                }
                mask = lsdg_uniform64(mask);
                const int s = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                int x = trip_x + q * 64 + s, y = trip_y;
                while (x >= P.W) { x -= P.W; ++y; }
#if PSL_GROW_DIAG == 3   // diagnostic timing builds only (tools/ab_round3e.sh): 3 = no growth at all (every seed a singleton), 1 = no rectangle / refinement, 2 = no refinement;
                         // 4 = every region2rect twice, 5 = the refinement's statistics loop and the count / scratch passes of its radius steps twice (same results: their cost = the time added)
                if (true) {
#else
                if ((smq >> s) & 1ull) {
#endif
                    // a static singleton: the region it would grow is itself.  Its mark joins the pending ones if it lies in their window;
                    // otherwise those are waited for and it opens a window of its own.
                    if (!LU) {
                        const int dxp = x - pd.pox, dyp = y - pd.poy;
                        if (pd.PA != 0ull && (unsigned)dxp < 8u && (unsigned)dyp < 8u) {
                            pd.PA |= 1ull << (dyp * 8 + dxp);
                        } else {
                            if (pd.PA != 0ull) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            pd.PA = 1ull; pd.pox = x; pd.poy = y;
                        }
                    }
                    if (lane == 0) lsdg_mark<LU>(F, x + y * P.W, 1);
                    continue;
                }
                double reg_angle;
                bool touched = false;
#ifdef PSL_GROW_STATS
                const unsigned long long st_g0 = __builtin_amdgcn_s_memtime();
#endif
                int reg_size = lsdg_region_grow4<LU>(F, x, y, &reg_angle, P.prec, fcP, pd, false, trip_end, touched);
#ifdef PSL_GROW_STATS
                if (frame == 0 && lane == 0) g_gstats[11] += __builtin_amdgcn_s_memtime() - st_g0;
                const unsigned long long st_r1 = __builtin_amdgcn_s_memtime();
#endif
                if (touched) { stale = true; dirty = true; }
                if (reg_size < P.min_reg_size) continue;
#if PSL_GROW_DIAG == 1
                continue;
#endif
                LsdRect rec;
                lsdw_region2rect(F, reg_size, reg_angle, P.prec, &rec);
#ifdef PSL_GROW_STATS
                const unsigned long long st_r2 = __builtin_amdgcn_s_memtime();
                if (frame == 0 && lane == 0) g_gstats[9] += st_r2 - st_r1;
#endif
#if PSL_GROW_DIAG == 2
                const int kept = reg_size;
#else
                const int kept = lsdw_refine<LU>(F, reg_size, reg_angle, P.prec, &rec, 0.7, pd, trip_end, touched);
#endif
#ifdef PSL_GROW_STATS
                if (frame == 0 && lane == 0) g_gstats[10] += __builtin_amdgcn_s_memtime() - st_r2;
#endif
                if (touched) { stale = true; dirty = true; }
                if (!kept) continue;
                if (count < P.maxseg && lane == 0) {
                    if (P.refine >= 2) {  // LSD_REFINE_ADV: the NFA validation never touches `used`, so it runs afterwards, one wave per rectangle
                        double* r = rects + ((size_t)frame * P.maxseg + count) * PSL_LSD_RECT_F64;
                        r[0] = rec.x1; r[1] = rec.y1; r[2] = rec.x2; r[3] = rec.y2; r[4] = rec.width; r[5] = rec.theta; r[6] = rec.dx; r[7] = rec.dy;
                    } else {
                        psl_lsd_store_segment(P, rec.x1, rec.y1, rec.x2, rec.y2, out + 4 * count);
                    }
                }
                ++count;
            }
        }
    }
    if (HELPERS && lane == 0) *(volatile int*)&s_scan_unit = (int)(npx >> 6);  // the helpers' exit condition
    if (lane == 0) nseg[frame] = count < P.maxseg ? count : P.maxseg;
#ifdef PSL_GROW_STATS
    if (frame == 0 && lane == 0) {
        const unsigned long long dc = __builtin_amdgcn_s_memtime() - st_c0, dr = __builtin_amdgcn_s_memrealtime() - st_r0;
        printf("grow clock: %llu shader cycles in %llu ticks of 100 MHz = %.0f MHz\n", dc, dr, (double)dc / (double)dr * 100.0);
        printf("grow stats: regions %llu rounds %llu pops %llu pops_with_candidates %llu decision_blocks %llu exact_tests %llu\n", g_gstats[0], g_gstats[1],
               g_gstats[2], g_gstats[3], g_gstats[4], g_gstats[6]);
        printf("grow cycles: growth of the scan's regions %llu (of which waiting for the window loads of ALL growths incl. the refinement's %llu), region2rect %llu, refinement %llu\n",
               g_gstats[11], g_gstats[8], g_gstats[9], g_gstats[10]);
        for (int k = 0; k < 16; ++k) g_gstats[k] = 0;
    }
#endif
}

#endif
