// HIP kernels of the ORB extraction path (gfx950, wave64). Product code.
// Reference behaviour reproduced: src/ORBextractor.cc (ComputePyramid :1107-1132,
// ComputeKeyPointsOctTree :765-853, DistributeOctTree :539-763, IC_Angle :77-104,
// GaussianBlur :1085-1086, computeOrbDescriptor :108-147) with OpenCV-3.2 semantics for the
// library calls.  All of it is integer / index work except the angle and
// the rotated sampling coordinates, which use psl_device_math.h (no contraction).
#ifndef PSL_ORB_KERNELS_H
#define PSL_ORB_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pslfe.h"
#include "psl_device_math.h"

#define PSL_EDGE 16          // minBorderX = EDGE_THRESHOLD - 3 (src/ORBextractor.cc:773)
#define PSL_MAXCELL 64       // largest FAST cell interior handled by k_fast_cells

struct OrbLevelP {
    int w, h, pitch;            // level image (levels >= 1 live in the pyramid block)
    unsigned img_off;           // byte offset in the per-frame pyramid block (level 0: unused)
    unsigned blur_off;          // byte offset in the per-frame blurred block
    int nCols, nRows, wCell, hCell;
    int maxBX, maxBY;           // maxBorderX / maxBorderY
    int cell_off;               // first cell of this level in the per-frame cell arrays
    int cand_off, cand_cap;     // dense candidate segment of this level (per frame)
    int quota, kp_off, kp_cap;  // octree target N, output segment
    int nIni;
    float hX;
    float scale, kpsize;
    int tile_off, tiles_x, tiles_y;  // blur tiling (64 x 16 output tiles)
};

struct OrbParams {
    int nlevels, ncells, cellcap, cand_total, kp_total, out_cap, iniTh, minTh, ntiles;
    int blurK[7];
    int umax[16];
    OrbLevelP lv[PSLFE_MAX_LEVELS];
};

struct FrameSrc {          // where level 0 (the caller's image) and the derived levels live
    const uint8_t* img0;   // frame 0 of the input batch
    int stride0;
    size_t fstride0;
    uint8_t* pyr;          // per-frame pyramid block (levels >= 1)
    size_t pyr_fstride;
    int nframes;
    int xcd;               // launches use the XCD-aware grid (8, items, ceil(nframes / 8)), see psl_item_frame
};

// Workgroup -> (item, frame).  The hardware deals consecutive workgroups round-robin to the 8 XCDs, each with its own
// L2.  With the plain grid (items, frames) the tiles that share cache lines (neighbouring cells / tiles of a level) land
// on 8 different L2s and every line is fetched from HBM/MALL up to 8 times (measured: 4.6x the algorithmic bytes for
// FAST).  Many-frames launches therefore use grid (8, items, ceil(frames / 8)): blockIdx.x is the XCD, so all items of
// a frame run on one XCD, in item order.  Single frames keep the plain grid (all 8 XCDs work on the one frame).
__device__ __forceinline__ bool psl_item_frame(const FrameSrc& S, int* item, int* frame) {
    if (S.xcd) {
        *frame = (int)(blockIdx.z * 8 + blockIdx.x);
        *item = (int)blockIdx.y;
        return *frame < S.nframes;
    }
    *item = (int)blockIdx.x;
    *frame = (int)blockIdx.y;
    return true;
}

__device__ __forceinline__ const uint8_t* psl_level_ptr(const OrbParams& P, const FrameSrc& S, int level, int frame, int* pitch) {
    if (level == 0) { *pitch = S.stride0; return S.img0 + (size_t)frame * S.fstride0; }
    *pitch = P.lv[level].pitch;
    return S.pyr + (size_t)frame * S.pyr_fstride + P.lv[level].img_off;
}

// ---------------------------------------------------------------------------------------------
// Pyramid: level l from level l-1, cv::resize INTER_LINEAR 8UC1 fixed point (Appendix A.3).
// Tables (xofs, alpha, yofs, beta) are built on the host exactly as OpenCV builds them.
// The source is staged through LDS: a workgroup produces a 64 x 32 block of the level and first
// copies the source rectangle it needs (<= PSL_PYR_TR rows of <= PSL_PYR_TD dwords) with coalesced dword
// loads; the taps of a thread then come from LDS instead of scattered byte loads from HBM/L2.  A thread makes
// 4 adjacent pixels of two rows (16 apart), sharing the column tables; all table loads are issued before the
// tile loads so that the workgroup pays two memory round trips, not three.
// ---------------------------------------------------------------------------------------------
typedef unsigned short pyr_u16x2 __attribute__((ext_vector_type(2)));
#define PSL_PYR_TD 36   // tile pitch in dwords
#define PSL_PYR_TR 48   // tile rows
#define PSL_PYR_BH 32   // block height
__global__ __launch_bounds__(256) void k_pyr_resize_tiled(OrbParams P, FrameSrc S, int level,
                                                           const int* __restrict__ xofs, const short2* __restrict__ alpha,
                                                           const int* __restrict__ yofs, const short2* __restrict__ beta) {
    __shared__ uint32_t s_tile[PSL_PYR_TR * PSL_PYR_TD + 4];  // + 4: the three-dword reads below may run past the last row
    const OrbLevelP L = P.lv[level];
    const int tid = threadIdx.x;
    int item, frame;
    if (!psl_item_frame(S, &item, &frame)) return;
    const int nbx = (L.pitch + 63) / 64;
    const int by = item / nbx, bx = item - by * nbx;
    const int x0 = bx * 64, y0 = by * PSL_PYR_BH;
    int spitch;
    const uint8_t* src = psl_level_ptr(P, S, level - 1, frame, &spitch);
    const int sw = P.lv[level - 1].w, sh = P.lv[level - 1].h;
    // this thread's tables
    const int x4 = x0 + (tid & 15) * 4;
    int sx[4];
    short2 a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int dx = x4 + j;
        dx = dx < L.w ? dx : L.w - 1;  // padding columns repeat the last pixel
        sx[j] = xofs[dx];
        a[j] = alpha[dx];
        if (sx[j] + 1 >= P.lv[level - 1].w) a[j] = make_short2(2048, 0);  // clamped last column: h = p * 2048 (cv::resize)
    }
    int sy0[2], sy1[2];
    short2 b[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int dy = min(y0 + (tid >> 4) + 16 * rr, L.h - 1);
        const int t = yofs[dy];
        sy0[rr] = t < 0 ? 0 : (t >= sh ? sh - 1 : t);
        sy1[rr] = t + 1 < 0 ? 0 : (t + 1 >= sh ? sh - 1 : t + 1);
        b[rr] = beta[dy];
    }
    // source rectangle of this block (tables are non-decreasing)
    const int xe = min(x0 + 63, L.w - 1), ye = min(y0 + PSL_PYR_BH - 1, L.h - 1);
    const int cfirst = xofs[min(x0, L.w - 1)], clast = min(xofs[xe] + 1, sw - 1);
    const int rfirst = min(max(yofs[y0], 0), sh - 1), rlast = min(max(yofs[ye] + 1, 0), sh - 1);
    const int cbase = cfirst & ~3;
    const int ndw = ((clast - cbase) >> 2) + 1, nrows = rlast - rfirst + 1;
    // dword staging needs 4-byte aligned rows whose last dword stays inside the row pitch
    const bool staged = ndw <= PSL_PYR_TD && nrows <= PSL_PYR_TR && cfirst >= 0 &&
                        ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)spitch) & 3) == 0 && cbase + ndw * 4 <= spitch;
    if (staged) {
        const uint32_t mdw = (1048576u + (uint32_t)ndw - 1u) / (uint32_t)ndw;  // k / ndw == (k * mdw) >> 20 for k < 4096
        for (int k = tid; k < nrows * ndw; k += 256) {
            const int r = (int)(((uint32_t)k * mdw) >> 20), d = k - r * ndw;
            s_tile[r * PSL_PYR_TD + d] = *reinterpret_cast<const uint32_t*>(src + (size_t)(rfirst + r) * spitch + cbase + d * 4);
        }
    }
    __syncthreads();
    if (x4 >= L.pitch) return;
    // byte offsets of the four left taps inside the three dwords that start at this thread's first tap
    const int d0 = (sx[0] - cbase) >> 2;
    int ofs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ofs[j] = sx[j] - cbase - 4 * d0;
    const bool fits = ofs[0] >= 0 && ofs[1] >= ofs[0] && ofs[2] >= ofs[1] && ofs[3] >= ofs[2] && ofs[3] <= 7;  // scale factors up to ~1.3
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int dy = y0 + (tid >> 4) + 16 * rr;
        if (dy >= L.h) break;
        uint8_t* dst = S.pyr + (size_t)frame * S.pyr_fstride + L.img_off + (size_t)dy * L.pitch;
        uint32_t packed = 0;
        if (staged && fits) {
            // The eight taps of a source row (4 pixels x 2 columns) lie within 12 bytes: three aligned dwords per source row
            // instead of eight byte reads (the kernel was bound by LDS instructions), a tap pair is cut out with v_alignbyte,
            // spread to 16-bit lanes with v_perm and weighted with one v_dot2_u32_u16.
            const uint32_t* q0 = s_tile + (sy0[rr] - rfirst) * PSL_PYR_TD + d0;
            const uint32_t* q1 = s_tile + (sy1[rr] - rfirst) * PSL_PYR_TD + d0;
            const uint32_t u0 = q0[0], u1 = q0[1], u2 = q0[2], v0 = q1[0], v1 = q1[1], v2 = q1[2];
            const uint32_t bx = (uint32_t)(int)b[rr].x, by = (uint32_t)(int)b[rr].y;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool hi = ofs[j] >= 4;
                const uint32_t t0 = __builtin_amdgcn_alignbyte(hi ? u2 : u1, hi ? u1 : u0, (uint32_t)(ofs[j] & 3));
                const uint32_t t1 = __builtin_amdgcn_alignbyte(hi ? v2 : v1, hi ? v1 : v0, (uint32_t)(ofs[j] & 3));
                const pyr_u16x2 cf = __builtin_bit_cast(pyr_u16x2, a[j]);
                const uint32_t h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(pyr_u16x2, __builtin_amdgcn_perm(0u, t0, 0x0c010c00u)), cf, 0u, false);
                const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(pyr_u16x2, __builtin_amdgcn_perm(0u, t1, 0x0c010c00u)), cf, 0u, false);
                const uint32_t v = (((bx * (h0 >> 4)) >> 16) + ((by * (h1 >> 4)) >> 16) + 2u) >> 2;
                packed |= (v & 0xffu) << (8 * j);
            }
        } else if (staged) {
            const uint8_t* t0 = reinterpret_cast<const uint8_t*>(s_tile) + (sy0[rr] - rfirst) * (PSL_PYR_TD * 4) - cbase;
            const uint8_t* t1 = reinterpret_cast<const uint8_t*>(s_tile) + (sy1[rr] - rfirst) * (PSL_PYR_TD * 4) - cbase;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int h0, h1;
                if (sx[j] + 1 < sw) {
                    h0 = t0[sx[j]] * a[j].x + t0[sx[j] + 1] * a[j].y;
                    h1 = t1[sx[j]] * a[j].x + t1[sx[j] + 1] * a[j].y;
                } else {
                    h0 = t0[sx[j]] * 2048;
                    h1 = t1[sx[j]] * 2048;
                }
                const int v = ((((int)b[rr].x * (h0 >> 4)) >> 16) + (((int)b[rr].y * (h1 >> 4)) >> 16) + 2) >> 2;
                packed |= (uint32_t)(v & 0xff) << (8 * j);
            }
        } else {
            const uint8_t* r0 = src + (size_t)sy0[rr] * spitch;
            const uint8_t* r1 = src + (size_t)sy1[rr] * spitch;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int h0, h1;
                if (sx[j] + 1 < sw) {
                    h0 = r0[sx[j]] * a[j].x + r0[sx[j] + 1] * a[j].y;
                    h1 = r1[sx[j]] * a[j].x + r1[sx[j] + 1] * a[j].y;
                } else {
                    h0 = r0[sx[j]] * 2048;
                    h1 = r1[sx[j]] * 2048;
                }
                const int v = ((((int)b[rr].x * (h0 >> 4)) >> 16) + (((int)b[rr].y * (h1 >> 4)) >> 16) + 2) >> 2;
                packed |= (uint32_t)(v & 0xff) << (8 * j);
            }
        }
        *reinterpret_cast<uint32_t*>(dst + x4) = packed;
    }
}

// ---------------------------------------------------------------------------------------------
// FAST-9/16 per cell with score NMS and the per-cell threshold fallback (src/ORBextractor.cc:789-829
// around cv::FAST, Appendix A.6).  One workgroup = one cell of one level of one frame.
//   S(p) = max over the 16 arcs of 9 ring pixels of min(v - ring), and of min(ring - v), minus 1
//        = cornerScore<16>; p is a corner at threshold t  <=>  S(p) >= t.
//   cv::FAST's NMS (score strictly greater than the 8 neighbours, scores outside the scanned
//   interior = 0) at threshold t keeps exactly { p : S(p) >= t and S(p) > S(q) for all 8 q } because
//   a neighbour below t has S(q) < t <= S(p) anyway.
// Survivors are written in raster order: x | y << 12 | S << 24, (x, y) relative to minBorder.
// ---------------------------------------------------------------------------------------------
// k_fast_cells4: dense work only where it is needed.
//   * the tile is stored with interior column 0 at a dword boundary (tile byte = 1 + tile x), shifting the
//     global dwords with v_alignbyte on the way in, so the 9 ring samples of the quick test for 4 pixels come
//     from 11 aligned LDS dword reads (+ 6 alignbyte) instead of 36 byte reads; four horizontally adjacent
//     pixels per thread, so index arithmetic, ballots and compaction are paid once per 4 pixels and a typical
//     31 x 31 cell is one pass;
//   * the quick test records which polarity can reach minTh (ring brighter / ring darker); the score is then
//     computed for that polarity only - exact after thresholding, because a polarity that fails the quick
//     test scores below minTh - and both only for the rare pixel where both pass;
//   * NMS and the raster-ordered output are driven by the compacted list (a few % of the pixels): survivors
//     set a bit in a per-row mask, row prefix sums give the raster positions.
#ifndef PSL_FAST_DIAG
#define PSL_FAST_DIAG 0   // timing builds only (tools/ab_round3x.sh): 1 = the quick test twice, 2 = the scores twice, 3 = NMS + output twice, 4 = tile load + clears twice (same results)
#endif
#define PSL_FAST4_TP 76   // tile pitch (bytes): 1 + (64 + 6) + slack, multiple of 4
#define PSL_FAST4_SP 72   // score pitch (bytes): interior x at byte 4 + x
__device__ __forceinline__ uint32_t psl_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }

// max over the 16 arcs of 9 contiguous ring pixels of min(sgn * (v - ring)), minus 1: the cornerScore of one polarity
__device__ __forceinline__ int psl_fast_score_pol(const uint8_t* c, const int tp, const int sgn) {
    const int sv = sgn * (int)c[0];
    int e[16];
    e[0] = sv - sgn * c[3 * tp];       e[1] = sv - sgn * c[3 * tp + 1];   e[2] = sv - sgn * c[2 * tp + 2];   e[3] = sv - sgn * c[tp + 3];
    e[4] = sv - sgn * c[3];            e[5] = sv - sgn * c[-tp + 3];      e[6] = sv - sgn * c[-2 * tp + 2];  e[7] = sv - sgn * c[-3 * tp + 1];
    e[8] = sv - sgn * c[-3 * tp];      e[9] = sv - sgn * c[-3 * tp - 1];  e[10] = sv - sgn * c[-2 * tp - 2]; e[11] = sv - sgn * c[-tp - 3];
    e[12] = sv - sgn * c[-3];          e[13] = sv - sgn * c[tp - 3];      e[14] = sv - sgn * c[2 * tp - 2];  e[15] = sv - sgn * c[3 * tp - 1];
    int lo2[16], lo4[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) lo2[k] = min(e[k], e[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) lo4[k] = min(lo2[k], lo2[(k + 2) & 15]);
    int A = -256;
#pragma unroll
    for (int k = 0; k < 16; ++k) A = max(A, min(min(lo4[k], lo4[(k + 4) & 15]), e[(k + 8) & 15]));
    return A - 1;
}

#ifndef PSL_FAST_SCORE_PK
#define PSL_FAST_SCORE_PK 1   // 0: psl_fast_score_pol, one ring difference per instruction (A/B, tools/ab_build.sh)
#endif
// The same score with TWO ring positions per instruction (packed signed 16-bit: the differences lie in [-255, 255]).  The scores are 36 % of the kernel
// (12.9 of 35.4 ms per 12288 dense frames, profiles/r03g_fast_parts_twice.log) and the kernel is bound by vector issue.  E[j] = (e[2j], e[2j+1]),
// O[j] = (e[2j+1], e[2j+2]) (one v_alignbit of two E's); then min over a pair of neighbours, over four, over the nine of an arc, and the maximum
// over the arcs, all on pairs: 41 packed instructions instead of 80, and e = sgn v - sgn ring is one packed multiply-add per pair.
// Measured: 35.15 -> 34.63 ms (profiles/r03g_ab_fast_score_pk.log) - far less than the instruction count promised: the 16 byte reads of the ring from LDS, not the
// min / max tree, are what a score costs.
typedef short psl_i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int psl_fast_score_pol_pk(const uint8_t* c, const int tp, const int sgn) {
    const uint32_t r[16] = {c[3 * tp],  c[3 * tp + 1],  c[2 * tp + 2],  c[tp + 3],  c[3],  c[-tp + 3], c[-2 * tp + 2], c[-3 * tp + 1],
                            c[-3 * tp], c[-3 * tp - 1], c[-2 * tp - 2], c[-tp - 3], c[-3], c[tp - 3],  c[2 * tp - 2],  c[3 * tp - 1]};
    const short sv = (short)(sgn * (int)c[0]), ns = (short)-sgn;
    const psl_i16x2 SV = {sv, sv}, NS = {ns, ns};
    psl_i16x2 E[8], O[8], L2[8], L4[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) E[j] = __builtin_bit_cast(psl_i16x2, r[2 * j] | (r[2 * j + 1] << 16)) * NS + SV;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        O[j] = __builtin_bit_cast(psl_i16x2, __builtin_amdgcn_alignbit(__builtin_bit_cast(uint32_t, E[(j + 1) & 7]), __builtin_bit_cast(uint32_t, E[j]), 16u));
#pragma unroll
    for (int j = 0; j < 8; ++j) L2[j] = __builtin_elementwise_min(E[j], O[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) L4[j] = __builtin_elementwise_min(L2[j], L2[(j + 1) & 7]);
    psl_i16x2 T[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) T[j] = __builtin_elementwise_min(__builtin_elementwise_min(L4[j], L4[(j + 2) & 7]), E[(j + 4) & 7]);
    const psl_i16x2 a = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_elementwise_max(T[0], T[1]), __builtin_elementwise_max(T[2], T[3])),
                                                  __builtin_elementwise_max(__builtin_elementwise_max(T[4], T[5]), __builtin_elementwise_max(T[6], T[7])));
    const int A = a.x > a.y ? (int)a.x : (int)a.y;
    return A - 1;
}

__global__ __launch_bounds__(256, 8) void k_fast_cells4(OrbParams P, FrameSrc S, const uint32_t* __restrict__ celltab,
                                                      int* __restrict__ cellcnt, uint32_t* __restrict__ cellcand, int cell_begin) {
    __shared__ __attribute__((aligned(16))) uint32_t s_tile32[(PSL_MAXCELL + 6) * (PSL_FAST4_TP / 4) + 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_score32[(PSL_MAXCELL + 2) * (PSL_FAST4_SP / 4)];
    __shared__ uint32_t s_rowmask[2][PSL_MAXCELL][2];  // [iniTh | minTh][row][x >> 5]: NMS survivors
    __shared__ int s_rowoff[PSL_MAXCELL];
    __shared__ uint16_t s_list[PSL_MAXCELL * PSL_MAXCELL];  // y << 6 | x | polarity << 12 of pixels that pass the quick test
    __shared__ int s_nlist, s_use, s_total;
    uint8_t* s_tile = reinterpret_cast<uint8_t*>(s_tile32);
    uint8_t* s_score = reinterpret_cast<uint8_t*>(s_score32);

    int cell, frame;
    if (!psl_item_frame(S, &cell, &frame)) return;
    cell += cell_begin;  // the launch covers the cells [cell_begin, cell_begin + items)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t ct = celltab[cell];  // level | row << 8 | column << 20 of the cell (host table; one scalar load)
    const int level = (int)(ct & 0xff), i = (int)((ct >> 8) & 0xfff), j = (int)(ct >> 20);
    const OrbLevelP L = P.lv[level];
    int* out_cnt = cellcnt + (size_t)frame * P.ncells + cell;
    uint32_t* out = cellcand + ((size_t)frame * P.ncells + cell) * P.cellcap;

    const int iniY = PSL_EDGE + i * L.hCell, iniX = PSL_EDGE + j * L.wCell;  // src/ORBextractor.cc:789-806
    int maxY = iniY + L.hCell + 6, maxX = iniX + L.wCell + 6;
    if (maxY > L.maxBY) maxY = L.maxBY;
    if (maxX > L.maxBX) maxX = L.maxBX;
    const int tw = maxX - iniX, th = maxY - iniY;
    if (iniY >= L.maxBY - 3 || iniX >= L.maxBX - 6 || tw < 7 || th < 7) {
        if (tid == 0) *out_cnt = 0;
        return;
    }
    const int iw = tw - 6, ih = th - 6;

    int pitch;
    const uint8_t* img = psl_level_ptr(P, S, level, frame, &pitch);
    // LDS tile byte (1 + x) of row y = image pixel (iniX + x, iniY + y); dword d of a row = image bytes
    // iniX - 1 + 4d .. + 3, assembled from the two aligned global dwords that hold them
    const bool aligned4 = ((reinterpret_cast<uintptr_t>(img) | (uintptr_t)pitch) & 3) == 0 && iniX >= 4 && maxX + 8 <= pitch;
#if PSL_FAST_DIAG == 4
  for (int diag_rep = 0; diag_rep < 2; ++diag_rep) {
    asm volatile("" ::: "memory");
#endif
    const int ndw = (tw + 4) >> 2;  // <= 18
    if (aligned4) {
        const int gx0 = (iniX - 1) & ~3;
        const uint32_t sh = (uint32_t)((iniX - 1) & 3);
        const uint32_t mdw = (1048576u + (uint32_t)ndw - 1u) / (uint32_t)ndw;  // k / ndw == (k * mdw) >> 20 for k < 4096
        for (int k = tid; k < th * ndw; k += 256) {
            const int y = (int)(((uint32_t)k * mdw) >> 20), d = k - y * ndw;
            const uint32_t* g = reinterpret_cast<const uint32_t*>(img + (size_t)(iniY + y) * pitch + gx0) + d;
            s_tile32[y * (PSL_FAST4_TP / 4) + d] = psl_alignbyte(g[1], g[0], sh);
        }
    } else {
        for (int y = wave; y < th; y += 4) {
            const uint8_t* row = img + (size_t)(iniY + y) * pitch + iniX;
            for (int x = lane; x < tw; x += 64) s_tile[y * PSL_FAST4_TP + 1 + x] = row[x];
        }
    }
    for (int k = tid; k < (ih + 2) * (PSL_FAST4_SP / 4); k += 256) s_score32[k] = 0;
    (&s_rowmask[0][0][0])[tid] = 0;  // 2 * 64 * 2 words
    if (tid == 0) s_nlist = 0;
    __syncthreads();
#if PSL_FAST_DIAG == 4
  }
#endif

    const int ng = (iw + 3) >> 2;            // groups of 4 pixels per row, <= 16
    const int nitems = ng * ih;              // <= 1024
    const int npass = (nitems + 255) >> 8;   // <= 4
    const uint32_t magic = (1048576u + (uint32_t)ng - 1u) / (uint32_t)ng;  // idx / ng == (idx * magic) >> 20 for idx < 4096
    const int minTh = P.minTh, iniTh = P.iniTh;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // Quick reject (exact necessary condition): an arc of 9 contains one pixel of every opposite pair, so a pixel
    // with S >= minTh has, for the 4 pairs (0,8) (2,10) (4,12) (6,14), one member > v+minTh in each pair (polarity
    // bit 0) or one member < v-minTh in each pair (polarity bit 1).
    for (int p = 0; p < npass; ++p) {
        const int idx = p * 256 + tid;
        uint32_t m4 = 0, pol = 0;  // pol: 2 bits per pixel
        int y = 0, g = 0;
#if PSL_FAST_DIAG == 1
      for (int diag_rep = 0; diag_rep < 2; ++diag_rep) {
        asm volatile("" ::: "memory");
#endif
        if (idx < nitems) {
            y = (int)(((uint32_t)idx * magic) >> 20);
            g = idx - y * ng;
            // centre of pixel 4g + jj: tile byte 4(g+1) + jj of tile row y + 3
            const uint32_t* r = &s_tile32[(y + 3) * (PSL_FAST4_TP / 4) + g];
            const uint32_t m0 = r[0], m1 = r[1], m2 = r[2];
            const uint32_t R0 = r[3 * (PSL_FAST4_TP / 4) + 1], R8 = r[-3 * (PSL_FAST4_TP / 4) + 1];
            const uint32_t* rp = r + 2 * (PSL_FAST4_TP / 4);
            const uint32_t* rm = r - 2 * (PSL_FAST4_TP / 4);
            const uint32_t p0 = rp[0], p1 = rp[1], p2 = rp[2], q0 = rm[0], q1 = rm[1], q2 = rm[2];
            // Two pixels per instruction: v_perm lifts the bytes of a pixel pair (byte offsets o, o + 1 of the 8-byte
            // pool hi:lo) to two u16, packed min/max/add/sub do the test, the sign bits of hi - bmin and dmax - lo are
            // the two polarity flags.
            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
#define PSL_PX2(hi, lo, o) __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (uint32_t)(o) | ((uint32_t)((o) + 1) << 16)))
            const u16x2 t2 = __builtin_bit_cast(u16x2, (uint32_t)minTh * 0x10001u);
            uint32_t sb[2], sd[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // pixels 4g + 2h, 4g + 2h + 1: centre bytes 4 + 2h, 5 + 2h of the pool m1:m0
                const int o = 2 * h;
                const u16x2 c = PSL_PX2(m1, m0, 4 + o);
                const u16x2 r0 = PSL_PX2(0u, R0, o), r8 = PSL_PX2(0u, R8, o);
                const u16x2 r12 = PSL_PX2(m1, m0, 1 + o), r4 = PSL_PX2(m2, m1, 3 + o);      // centre row, columns -3 / +3
                const u16x2 r14 = PSL_PX2(p1, p0, 2 + o), r2 = PSL_PX2(p2, p1, 2 + o);      // row +2, columns -2 / +2
                const u16x2 r10 = PSL_PX2(q1, q0, 2 + o), r6 = PSL_PX2(q2, q1, 2 + o);      // row -2, columns -2 / +2
                const u16x2 bmin = __builtin_elementwise_min(__builtin_elementwise_min(__builtin_elementwise_max(r0, r8), __builtin_elementwise_max(r4, r12)),
                                                             __builtin_elementwise_min(__builtin_elementwise_max(r2, r10), __builtin_elementwise_max(r6, r14)));
                const u16x2 dmax = __builtin_elementwise_max(__builtin_elementwise_max(__builtin_elementwise_min(r0, r8), __builtin_elementwise_min(r4, r12)),
                                                             __builtin_elementwise_max(__builtin_elementwise_min(r2, r10), __builtin_elementwise_min(r6, r14)));
                const u16x2 hi = c + t2, lo = c - t2;  // lo wraps below 0: the differences below are read as signed 16 bit
                sb[h] = __builtin_bit_cast(uint32_t, (u16x2)(hi - bmin)) & 0x80008000u;  // bmin > hi
                sd[h] = __builtin_bit_cast(uint32_t, (u16x2)(dmax - lo)) & 0x80008000u;  // dmax < lo
            }
#undef PSL_PX2
            // polarity bits (2 per pixel): pixel 2h from bit 15, pixel 2h + 1 from bit 31
            pol = ((sb[0] >> 15) & 0x1u) | ((sd[0] >> 14) & 0x2u) | ((sb[0] >> 29) & 0x4u) | ((sd[0] >> 28) & 0x8u) |
                  ((sb[1] >> 11) & 0x10u) | ((sd[1] >> 10) & 0x20u) | ((sb[1] >> 25) & 0x40u) | ((sd[1] >> 24) & 0x80u);
            const int nin = iw - 4 * g;  // pixels of this group inside the cell
            if (nin < 4) pol &= (1u << (2 * nin)) - 1u;
            m4 = ((pol | (pol >> 1)) & 0x1u) | (((pol | (pol >> 1)) >> 1) & 0x2u) | (((pol | (pol >> 1)) >> 2) & 0x4u) | (((pol | (pol >> 1)) >> 3) & 0x8u);
        }
#if PSL_FAST_DIAG == 1
        asm volatile("" : "+v"(m4), "+v"(pol));
      }
#endif
        // order inside s_list is irrelevant: one slot range per (wave, jj)
        const unsigned long long b0 = __ballot(m4 & 1), b1 = __ballot(m4 & 2), b2 = __ballot(m4 & 4), b3 = __ballot(m4 & 8);
        const int n0 = __popcll(b0), n1 = __popcll(b1), n2 = __popcll(b2), n3 = __popcll(b3);
        const int tot = n0 + n1 + n2 + n3;
        if (tot) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_nlist, tot);
            base = __shfl(base, 0);
            const uint32_t e = (uint32_t)(y << 6) | (uint32_t)(4 * g);
            if (m4 & 1) s_list[base + __popcll(b0 & lt)] = (uint16_t)(e | ((pol & 3) << 12));
            if (m4 & 2) s_list[base + n0 + __popcll(b1 & lt)] = (uint16_t)((e + 1) | (((pol >> 2) & 3) << 12));
            if (m4 & 4) s_list[base + n0 + n1 + __popcll(b2 & lt)] = (uint16_t)((e + 2) | (((pol >> 4) & 3) << 12));
            if (m4 & 8) s_list[base + n0 + n1 + n2 + __popcll(b3 & lt)] = (uint16_t)((e + 3) | (((pol >> 6) & 3) << 12));
        }
    }
    __syncthreads();
    const int nlist = s_nlist;
#if PSL_FAST_DIAG == 2
  for (int diag_rep = 0; diag_rep < 2; ++diag_rep) {
    asm volatile("" ::: "memory");
#endif
    for (int k = tid; k < nlist; k += 256) {
        const int e = s_list[k];
        const int y = (e >> 6) & 63, x = e & 63, pl = e >> 12;
        const uint8_t* c = &s_tile[(y + 3) * PSL_FAST4_TP + x + 4];
        // polarity bit 0: ring brighter than the centre (ring - v), bit 1: ring darker (v - ring)
#if PSL_FAST_SCORE_PK
        int s = psl_fast_score_pol_pk(c, PSL_FAST4_TP, (pl & 1) ? -1 : 1);
        if (pl == 3) s = max(s, psl_fast_score_pol_pk(c, PSL_FAST4_TP, 1));
#else
        int s = psl_fast_score_pol(c, PSL_FAST4_TP, (pl & 1) ? -1 : 1);
        if (pl == 3) s = max(s, psl_fast_score_pol(c, PSL_FAST4_TP, 1));
#endif
        s = s < minTh ? 0 : (s > 255 ? 255 : s);
        s_score[(y + 1) * PSL_FAST4_SP + x + 4] = (uint8_t)s;
    }
#if PSL_FAST_DIAG == 2
  }
#endif
    __syncthreads();
#if PSL_FAST_DIAG == 3
  for (int diag_rep = 0; diag_rep < 2; ++diag_rep) {
    asm volatile("" ::: "memory");
    __syncthreads();
#endif
    // cv::FAST's NMS: strictly greater than the 8 neighbours (scores outside the interior are 0)
    for (int k = tid; k < nlist; k += 256) {
        const int e = s_list[k];
        const int y = (e >> 6) & 63, x = e & 63;
        const uint8_t* c = &s_score[(y + 1) * PSL_FAST4_SP + x + 4];
        const int s = c[0];
        if (s > 0) {
            const int m = max(max(max((int)c[-1], (int)c[1]), max((int)c[-PSL_FAST4_SP - 1], (int)c[-PSL_FAST4_SP])),
                              max(max((int)c[-PSL_FAST4_SP + 1], (int)c[PSL_FAST4_SP - 1]), max((int)c[PSL_FAST4_SP], (int)c[PSL_FAST4_SP + 1])));
            if (s > m) {
                atomicOr(&s_rowmask[1][y][x >> 5], 1u << (x & 31));
                if (s >= iniTh) atomicOr(&s_rowmask[0][y][x >> 5], 1u << (x & 31));
            }
        }
    }
    __syncthreads();
    if (wave == 0) {  // row counts -> raster offsets; retry at minTh only if iniTh found nothing (:812-816)
        const int c_ini = __popc(s_rowmask[0][lane][0]) + __popc(s_rowmask[0][lane][1]);
        const int c_min = __popc(s_rowmask[1][lane][0]) + __popc(s_rowmask[1][lane][1]);
        int i_ini = c_ini, i_min = c_min;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(i_ini, o), v = __shfl_up(i_min, o);
            if (lane >= o) { i_ini += u; i_min += v; }
        }
        const int t_ini = __shfl(i_ini, 63), t_min = __shfl(i_min, 63);
        const int use = t_ini > 0 ? 0 : 1;
        s_rowoff[lane] = use == 0 ? i_ini - c_ini : i_min - c_min;
        if (lane == 0) { s_use = use; s_total = use == 0 ? t_ini : t_min; }
    }
    __syncthreads();
    const int use = s_use, total = s_total;
    for (int k = tid; k < nlist; k += 256) {
        const int e = s_list[k];
        const int y = (e >> 6) & 63, x = e & 63;
        const uint32_t w0 = s_rowmask[use][y][0], w1 = s_rowmask[use][y][1];
        const uint32_t mine = x < 32 ? (w0 >> x) & 1 : (w1 >> (x - 32)) & 1;
        if (mine) {
            const int before = x < 32 ? __popc(w0 & ((1u << x) - 1u)) : __popc(w0) + __popc(w1 & ((1u << (x - 32)) - 1u));
            const int pos = s_rowoff[y] + before;
            const uint32_t sc = s_score[(y + 1) * PSL_FAST4_SP + x + 4];
            if (pos < P.cellcap)
                out[pos] = (uint32_t)(x + 3 + j * L.wCell) | ((uint32_t)(y + 3 + i * L.hCell) << 12) | (sc << 24);
        }
    }
#if PSL_FAST_DIAG == 3
    if (diag_rep == 1)
#endif
    if (tid == 0) *out_cnt = total < P.cellcap ? total : P.cellcap;
#if PSL_FAST_DIAG == 3
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// DistributeOctTree as scans over arrays (prototype + fuzz against the list version:
// tools/octree_proto.cpp).  One workgroup per (level, frame); one thread per list node, node id ==
// list position; keys keep their original order and carry the id of the node that owns them.
// ---------------------------------------------------------------------------------------------
template <int BS>
__device__ __forceinline__ int psl_block_excl_scan(int v, int* s_w, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int k = 0; k < BS / 64; ++k) { const int t = s_w[k]; s_w[k] = acc; acc += t; }
        s_w[BS / 64] = acc;
    }
    __syncthreads();
    const int r = inc - v + s_w[w];
    *total = s_w[BS / 64];
    __syncthreads();
    return r;
}

template <int BS>
__global__ __launch_bounds__(BS) void k_octree(OrbParams P, const int* __restrict__ cellcnt,
                                                const uint32_t* __restrict__ cellcand, int* __restrict__ celloff,
                                                uint32_t* __restrict__ cand, uint16_t* __restrict__ knode,
                                                uint32_t* __restrict__ lvlkp, int* __restrict__ lvlcnt) {
    __shared__ short4 s_rect[2][BS];  // x0, y0, x1, y1
    __shared__ int s_cntn[2][BS];
    __shared__ int s_seq[2][BS];
    __shared__ short2 s_mid[BS];
    __shared__ int s_ccnt[BS][4];
    __shared__ int s_c[BS];    // children per processing rank
    __shared__ int s_cex[BS];  // exclusive scan of s_c over ranks
    __shared__ short s_newpos[BS];
    __shared__ short s_childpos[BS][4];
    __shared__ unsigned long long s_best[BS];
    __shared__ int s_w[BS / 64 + 1];
    __shared__ int s_misc[4];  // 0: ndiv, 1: nToExpand, 2: carry

    const int level = blockIdx.x, frame = blockIdx.y, tid = threadIdx.x;
    const OrbLevelP L = P.lv[level];
    const int ncell = L.nCols * L.nRows;
    const int* ccnt_g = cellcnt + (size_t)frame * P.ncells + L.cell_off;
    const uint32_t* ccand_g = cellcand + ((size_t)frame * P.ncells + L.cell_off) * P.cellcap;
    int* coff_g = celloff + (size_t)frame * P.ncells + L.cell_off;
    uint32_t* keys = cand + (size_t)frame * P.cand_total + L.cand_off;
    uint16_t* kn = knode + (size_t)frame * P.cand_total + L.cand_off;

    // ---- step 0: concatenate the cell lists in cell row-major order (== vToDistributeKeys order).  Offsets by a block
    // scan over the cells, kept in LDS (aliasing s_rect, which is not live yet); then thread = candidate: its cell by
    // binary search in the offsets, so the reads of all candidates are independent and several are in flight per
    // thread (walking the cells one after another put two dependent round trips per cell on the critical path: 40 %
    // of the kernel, measured with s_memtime).
    int* s_off = reinterpret_cast<int*>(&s_rect[0][0]);  // 2 * BS short4 = 4 * BS ints
    const bool offlds = ncell <= 4 * BS;
    if (tid == 0) s_misc[2] = 0;
    __syncthreads();
    for (int base = 0; base < ncell; base += BS) {
        const int c = base + tid;
        const int v = c < ncell ? ccnt_g[c] : 0;
        int tot;
        const int ex = psl_block_excl_scan<BS>(v, s_w, &tot);
        if (c < ncell) {
            const int o = s_misc[2] + ex;
            coff_g[c] = o;
            if (offlds) s_off[c] = o;
        }
        __syncthreads();
        if (tid == 0) s_misc[2] += tot;
        __syncthreads();
    }
    int K = s_misc[2];
    if (K > L.cand_cap) K = L.cand_cap;  // cannot happen: cand_cap = sum of the cell capacities
    for (int k0 = tid; k0 < K; k0 += 4 * BS) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * BS;
            v[u] = 0;
            if (k < K) {
                int lo = 0, hi = ncell;  // first cell whose offset exceeds k; the cell before it holds candidate k
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if ((offlds ? s_off[mid] : coff_g[mid]) > k) hi = mid; else lo = mid + 1;
                }
                const int cell = lo - 1;
                v[u] = ccand_g[(size_t)cell * P.cellcap + (k - (offlds ? s_off[cell] : coff_g[cell]))];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u * BS < K) keys[k0 + u * BS] = v[u];
    }
    __syncthreads();

    // ---- initial nodes (:541-583)
    const int nIni = L.nIni;
    const float hX = L.hX;
    const int H = L.maxBY - PSL_EDGE;
    if (tid < nIni) s_ccnt[tid][0] = 0;
    __syncthreads();
    for (int k = tid; k < K; k += BS) {
        int idx = (int)PSL_FDIV((float)(keys[k] & 0xfff), hX);
        if (idx >= nIni) idx = nIni - 1;
        kn[k] = (uint16_t)idx;
        atomicAdd(&s_ccnt[idx][0], 1);
    }
    __syncthreads();
    int n;
    {
        const int cnt0 = tid < nIni ? s_ccnt[tid][0] : 0;
        const int pos = psl_block_excl_scan<BS>(cnt0 > 0 ? 1 : 0, s_w, &n);
        if (tid < nIni) {
            s_newpos[tid] = (short)pos;
            if (cnt0 > 0) {
                s_rect[0][pos] = make_short4((short)(int)PSL_FMUL(hX, (float)tid), 0, (short)(int)PSL_FMUL(hX, (float)(tid + 1)), (short)H);
                s_cntn[0][pos] = cnt0;
                s_seq[0][pos] = tid;
            }
        }
        __syncthreads();
        for (int k = tid; k < K; k += BS) kn[k] = (uint16_t)s_newpos[kn[k]];
        __syncthreads();
    }
    int seq = nIni, cur = 0;
    bool phase2 = false;
    const int N = L.quota;

    while (true) {  // all loop-control values are workgroup-uniform
        const int prevSize = n;
        const int nxt = cur ^ 1;
        int mycnt = 0;
        short4 r4 = make_short4(0, 0, 0, 0);
        int mx = 0, my = 0;
        if (tid < n) {
            mycnt = s_cntn[cur][tid];
            r4 = s_rect[cur][tid];
            mx = r4.x + ((r4.z - r4.x + 1) >> 1);  // UL.x + ceil((UR.x-UL.x)/2)  (:483)
            my = r4.y + ((r4.w - r4.y + 1) >> 1);
            s_mid[tid] = make_short2((short)mx, (short)my);
            s_ccnt[tid][0] = 0; s_ccnt[tid][1] = 0; s_ccnt[tid][2] = 0; s_ccnt[tid][3] = 0;
        }
        if (tid == 0) { s_misc[0] = 0x7fffffff; s_misc[1] = 0; }
        __syncthreads();
        for (int k = tid; k < K; k += BS) {
            const int nd = kn[k];
            if (s_cntn[cur][nd] > 1) {
                const uint32_t key = keys[k];
                const short2 m = s_mid[nd];
                const int q = ((int)(key & 0xfff) >= m.x ? 1 : 0) + ((int)((key >> 12) & 0xfff) >= m.y ? 2 : 0);
                atomicAdd(&s_ccnt[nd][q], 1);
            }
        }
        __syncthreads();
        const bool isc = mycnt > 1;
        int cc[4] = {0, 0, 0, 0}, c = 0, e = 0;
        if (isc) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { cc[q] = s_ccnt[tid][q]; c += cc[q] > 0; e += cc[q] > 1; }
        }
        // processing rank among the nodes to divide
        int rank, ncand;
        if (!phase2) {
            rank = psl_block_excl_scan<BS>(isc ? 1 : 0, s_w, &ncand);  // list order (:597-660)
        } else {
            // sort by (size, creation sequence) descending (:684-685, convention H1)
            const int myseq = tid < n ? s_seq[cur][tid] : 0;
            int r = 0;
            if (isc)
                for (int jn = 0; jn < n; ++jn) {
                    const int cj = s_cntn[cur][jn], sj = s_seq[cur][jn];
                    r += (cj > 1) && (cj > mycnt || (cj == mycnt && sj > myseq));
                }
            rank = r;
            int dummy = psl_block_excl_scan<BS>(isc ? 1 : 0, s_w, &ncand);
            (void)dummy;
        }
        if (isc) s_c[rank] = c;
        __syncthreads();
        int C_all;
        {
            const int v = tid < ncand ? s_c[tid] : 0;
            const int ex = psl_block_excl_scan<BS>(v, s_w, &C_all);
            if (tid < ncand) {
                s_cex[tid] = ex;
                // list size after processing rank tid: n + (children so far) - (nodes erased so far)
                if (phase2 && n + (ex + v) - (tid + 1) >= N) atomicMin(&s_misc[0], tid + 1);  // break (:737-738)
            }
        }
        __syncthreads();
        const int ndiv = (phase2 && s_misc[0] < ncand) ? s_misc[0] : ncand;
        const int C = ndiv == ncand ? C_all : s_cex[ndiv];
        const bool divided = isc && rank < ndiv;
        int dummyTotal;
        const int keep = psl_block_excl_scan<BS>((tid < n && !divided) ? 1 : 0, s_w, &dummyTotal);
        if (tid < n) {
            if (!divided) {
                const int pos = C + keep;
                s_newpos[tid] = (short)pos;
                s_rect[nxt][pos] = r4;
                s_cntn[nxt][pos] = mycnt;
                s_seq[nxt][pos] = s_seq[cur][tid];
            } else {
                s_newpos[tid] = -1;
                const int cex = s_cex[rank];
                const int front = C - (cex + c);  // children of later-processed nodes sit in front (push_front)
                int seen = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (cc[q] == 0) continue;
                    const int pos = front + (c - 1 - seen);
                    s_rect[nxt][pos] = make_short4((q & 1) ? (short)mx : r4.x, (q & 2) ? (short)my : r4.y,
                                                   (q & 1) ? r4.z : (short)mx, (q & 2) ? r4.w : (short)my);
                    s_cntn[nxt][pos] = cc[q];
                    s_seq[nxt][pos] = seq + cex + seen;
                    s_childpos[tid][q] = (short)pos;
                    ++seen;
                }
                if (e) atomicAdd(&s_misc[1], e);
            }
        }
        __syncthreads();
        for (int k = tid; k < K; k += BS) {
            const int nd = kn[k];
            const int np = s_newpos[nd];
            if (np >= 0) kn[k] = (uint16_t)np;
            else {
                const uint32_t key = keys[k];
                const short2 m = s_mid[nd];
                const int q = ((int)(key & 0xfff) >= m.x ? 1 : 0) + ((int)((key >> 12) & 0xfff) >= m.y ? 2 : 0);
                kn[k] = (uint16_t)s_childpos[nd][q];
            }
        }
        const int nToExpand = s_misc[1];
        __syncthreads();
        n = C + n - ndiv;
        seq += C;
        cur = nxt;
        if (n >= N || n == prevSize) break;                       // (:665-669, :741-742)
        if (!phase2 && n + nToExpand * 3 > N) phase2 = true;      // (:670)
    }

    // ---- best key of each node: max response, first in key order (:746-760)
    if (tid < n) s_best[tid] = 0ull;
    __syncthreads();
    for (int k = tid; k < K; k += BS) {
        const unsigned long long v = ((unsigned long long)(keys[k] >> 24) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)k);
        atomicMax(&s_best[kn[k]], v);
    }
    __syncthreads();
    uint32_t* outk = lvlkp + (size_t)frame * P.kp_total + L.kp_off;
    if (tid < n && tid < L.kp_cap) {
        const uint32_t k = 0xffffffffu - (uint32_t)(s_best[tid] & 0xffffffffull);
        outk[tid] = keys[k];
    }
    if (tid == 0) lvlcnt[(size_t)frame * P.nlevels + level] = n < L.kp_cap ? n : L.kp_cap;
}

// ---------------------------------------------------------------------------------------------
// GaussianBlur 7x7 sigma 2, BORDER_REFLECT_101, on every (un-padded) level: OpenCV 3.2 8-bit
// path = integer kernel round(k*256), 32-bit row sums, (v + 2^15) >> 16 (Appendix A.4).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int psl_reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// 64 x 64 output tile per workgroup: 70 x 72-byte input rows staged in LDS (dword loads when the
// tile lies inside the image and the rows are 4-byte aligned, per-byte reflect-101 at the borders), row
// sums as u16 (255*257 fits), then the column pass; every thread produces 4 adjacent pixels per step.
#define PSL_BLUR_TH 64
__global__ __launch_bounds__(256) void k_blur7(OrbParams P, FrameSrc S, uint8_t* __restrict__ blur, size_t blur_fstride) {
    __shared__ __attribute__((aligned(16))) uint8_t s_in[(PSL_BLUR_TH + 6) * 72];
    __shared__ __attribute__((aligned(16))) uint16_t s_row[(PSL_BLUR_TH + 6) * 64];
    const int tid = threadIdx.x;
    int tile, frame;
    if (!psl_item_frame(S, &tile, &frame)) return;
    int level = 0;
    while (level + 1 < P.nlevels && tile >= P.lv[level + 1].tile_off) ++level;
    const OrbLevelP L = P.lv[level];
    const int t = tile - L.tile_off;
    const int ty = t / L.tiles_x, tx = t - ty * L.tiles_x;
    const int x0 = tx * 64, y0 = ty * PSL_BLUR_TH;
    const int rows = min(PSL_BLUR_TH, L.h - y0) + 6;  // input rows y0-3 .. y0+rows-4
    int pitch;
    const uint8_t* img = psl_level_ptr(P, S, level, frame, &pitch);
    // s_in[r][c] holds pixel (x0 - 4 + c, y0 - 3 + r), c in [0, 72)
    const bool fast = x0 >= 4 && x0 + 68 <= L.w && ((reinterpret_cast<uintptr_t>(img) | (uintptr_t)pitch) & 3) == 0;
    if (fast) {
        for (int k = tid; k < rows * 18; k += 256) {
            const int r = k / 18, c4 = k - r * 18;
            const int gy = psl_reflect101(y0 + r - 3, L.h);
            reinterpret_cast<uint32_t*>(s_in)[r * 18 + c4] = *reinterpret_cast<const uint32_t*>(img + (size_t)gy * pitch + x0 - 4 + c4 * 4);
        }
    } else {
        for (int k = tid; k < rows * 72; k += 256) {
            const int r = k / 72, c = k - r * 72;
            const int gy = psl_reflect101(y0 + r - 3, L.h), gx = psl_reflect101(x0 - 4 + c, L.w);
            s_in[r * 72 + c] = img[(size_t)gy * pitch + gx];
        }
    }
    __syncthreads();
    // Row pass with packed 16-bit math (row sums fit 16 bits: 255 * 257): P[t] = (v[t], v[t+1]) as two u16 built by
    // v_perm from the three dwords that hold the 10 input bytes; outputs (o0,o1) = sum K[t] P[t], (o2,o3) = sum K[t] P[t+2].
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const uint32_t K0 = (uint32_t)P.blurK[0], K1 = (uint32_t)P.blurK[1], K2 = (uint32_t)P.blurK[2], K3 = (uint32_t)P.blurK[3];
    const u16x2 k0 = __builtin_bit_cast(u16x2, K0 | (K0 << 16)), k1 = __builtin_bit_cast(u16x2, K1 | (K1 << 16));
    const u16x2 k2 = __builtin_bit_cast(u16x2, K2 | (K2 << 16)), k3 = __builtin_bit_cast(u16x2, K3 | (K3 << 16));
    for (int k = tid; k < rows * 16; k += 256) {
        const int r = k >> 4, g = k & 15;
        // output x = x0 + 4g + j reads stream bytes (4g + j + 1) .. (4g + j + 7) of the row: three aligned dwords
        const uint32_t* in32 = reinterpret_cast<const uint32_t*>(&s_in[r * 72 + g * 4]);
        const uint32_t w0 = in32[0], w1 = in32[1], w2 = in32[2];
        // v[n] = stream byte n + 1; perm(hi, lo, sel): selector bytes 0-3 pick from lo, 4-7 from hi, 0x0c = zero
#define PSL_PAIR(hi, lo, b0, b1) __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (uint32_t)(b0) | ((uint32_t)(b1) << 16)))
        const u16x2 p0 = PSL_PAIR(w1, w0, 1, 2), p1 = PSL_PAIR(w1, w0, 2, 3), p2 = PSL_PAIR(w1, w0, 3, 4), p3 = PSL_PAIR(w1, w0, 4, 5);
        const u16x2 p4 = PSL_PAIR(w1, w0, 5, 6), p5 = PSL_PAIR(w1, w0, 6, 7), p6 = PSL_PAIR(w2, w1, 3, 4), p7 = PSL_PAIR(w2, w1, 4, 5);
        const u16x2 p8 = PSL_PAIR(w2, w1, 5, 6);
#undef PSL_PAIR
        const u16x2 o01 = k0 * (p0 + p6) + k1 * (p1 + p5) + k2 * (p2 + p4) + k3 * p3;
        const u16x2 o23 = k0 * (p2 + p8) + k1 * (p3 + p7) + k2 * (p4 + p6) + k3 * p5;
        *reinterpret_cast<uint2*>(&s_row[r * 64 + g * 4]) = make_uint2(__builtin_bit_cast(uint32_t, o01), __builtin_bit_cast(uint32_t, o23));
    }
    __syncthreads();
    // Column pass: a thread makes 4 adjacent pixels of 4 consecutive rows from 10 row-sum rows read once.  Two
    // vertically adjacent row sums of a pixel are paired into one register (v_perm) and meet their two
    // coefficients in v_dot2_u32_u16; the 7th tap is a plain multiply-add.  Accumulators start at the rounding
    // constant 2^15; min(acc, 2^24 - 1) >> 16 is saturate_cast<uchar>((acc) >> 16).
    const int orows = rows - 6;
    uint8_t* dst = blur + (size_t)frame * blur_fstride + L.blur_off;
    const u16x2 c01 = __builtin_bit_cast(u16x2, K0 | (K1 << 16)), c23 = __builtin_bit_cast(u16x2, K2 | (K3 << 16));
    const u16x2 c45 = __builtin_bit_cast(u16x2, K2 | (K1 << 16));
    {
        const int g = tid & 15, oy0 = (tid >> 4) * 4;
        if (x0 + g * 4 < L.pitch && oy0 < orows) {
            uint2 q[10];
#pragma unroll
            for (int rr = 0; rr < 10; ++rr) q[rr] = *reinterpret_cast<const uint2*>(&s_row[min(oy0 + rr, rows - 1) * 64 + g * 4]);
            uint32_t acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int x = 0; x < 4; ++x) acc[i][x] = 1u << 15;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {  // pair of rows (rr, rr + 1)
                const u16x2 a0 = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(q[rr + 1].x, q[rr].x, 0x05040100u));
                const u16x2 a1 = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(q[rr + 1].x, q[rr].x, 0x07060302u));
                const u16x2 a2 = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(q[rr + 1].y, q[rr].y, 0x05040100u));
                const u16x2 a3 = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(q[rr + 1].y, q[rr].y, 0x07060302u));
#pragma unroll
                for (int i = 0; i < 4; ++i) {  // output row oy0 + i uses the pairs starting at i, i + 2, i + 4
                    const int m = rr - i;
                    if (m == 0 || m == 2 || m == 4) {
                        const u16x2 c = m == 0 ? c01 : (m == 2 ? c23 : c45);
                        acc[i][0] = __builtin_amdgcn_udot2(a0, c, acc[i][0], false);
                        acc[i][1] = __builtin_amdgcn_udot2(a1, c, acc[i][1], false);
                        acc[i][2] = __builtin_amdgcn_udot2(a2, c, acc[i][2], false);
                        acc[i][3] = __builtin_amdgcn_udot2(a3, c, acc[i][3], false);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (oy0 + i >= orows) break;
                const uint2 t = q[i + 6];  // 7th tap, coefficient K0
                acc[i][0] += K0 * (t.x & 0xffff); acc[i][1] += K0 * (t.x >> 16);
                acc[i][2] += K0 * (t.y & 0xffff); acc[i][3] += K0 * (t.y >> 16);
                const uint32_t m0 = min(acc[i][0], 0xffffffu), m1 = min(acc[i][1], 0xffffffu);
                const uint32_t m2 = min(acc[i][2], 0xffffffu), m3 = min(acc[i][3], 0xffffffu);
                // byte 2 of each -> bytes 0..3
                const uint32_t lo = __builtin_amdgcn_perm(m1, m0, 0x0c0c0602u), hi = __builtin_amdgcn_perm(m3, m2, 0x06020c0cu);
                *reinterpret_cast<uint32_t*>(dst + (size_t)(y0 + oy0 + i) * L.pitch + x0 + g * 4) = lo | hi;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Orientation (IC_Angle) + rBRIEF + final KeyPoint record: one wave per keypoint.
// Output order: levels 0..n-1 concatenated, inside a level the octree's list order (:1076-1103).
// ---------------------------------------------------------------------------------------------
__constant__ int8_t c_orb_pattern[1024] = {
#include "orb_pattern.inc"
};

#define PSL_DESC_PD 11  // LDS pitch of the descriptor patch in dwords: 37 + 3 alignment bytes -> 10, odd pitch
#define PSL_ORI_PD 9    // LDS pitch of the orientation patch in dwords: 31 + 3 alignment bytes -> 9
__global__ __launch_bounds__(256) void k_orient_describe(OrbParams P, FrameSrc S, const uint8_t* __restrict__ blur,
                                                          size_t blur_fstride, const uint32_t* __restrict__ lvlkp,
                                                          const int* __restrict__ lvlcnt, PslKeyPoint* __restrict__ kps,
                                                          uint8_t* __restrict__ desc, int* __restrict__ counts) {
    __shared__ uint32_t s_patch[4][37 * PSL_DESC_PD];
    __shared__ uint32_t s_ori[4][31 * PSL_ORI_PD];
    int grp, frame;
    if (!psl_item_frame(S, &grp, &frame)) return;
    const int lane = threadIdx.x & 63;
    const int slot = grp * 4 + (threadIdx.x >> 6);
    const int* lc = lvlcnt + (size_t)frame * P.nlevels;
    int level = -1, idx = slot, total = 0;
    for (int l = 0; l < P.nlevels; ++l) {
        const int c = lc[l];
        if (level < 0 && idx < c) level = l;
        if (level < 0) idx -= c;
        total += c;
    }
    if (slot == 0 && lane == 0) counts[frame] = total < P.out_cap ? total : P.out_cap;
    if (level < 0 || slot >= P.out_cap) return;
    const OrbLevelP L = P.lv[level];
    const uint32_t key = lvlkp[(size_t)frame * P.kp_total + L.kp_off + idx];
    const int cx = (int)(key & 0xfff) + PSL_EDGE, cy = (int)((key >> 12) & 0xfff) + PSL_EDGE;
    int pitch;
    const uint8_t* img = psl_level_ptr(P, S, level, frame, &pitch);

    // One memory round trip for everything the keypoint needs: the 31 x 31 patch of the level image (orientation)
    // and the 37 x 37 patch of the blurred level (the 512 rBRIEF samples lie within radius 18.4 of the keypoint, so
    // the rotated, rounded offsets stay in [-18, 18]) are fetched with row-coalesced dword loads that are all in
    // flight together, then parked in LDS; the scattered per-lane samples come from there.
    const uint8_t* bl = blur + (size_t)frame * blur_fstride + L.blur_off;
    uint32_t* patch = s_patch[threadIdx.x >> 6];
    uint32_t* ori = s_ori[threadIdx.x >> 6];
    const int pc0 = (cx - 18) & ~3, pr0 = cy - 18;  // blurred patch origin (dword-aligned column)
    const int oc0 = (cx - 15) & ~3, or0 = cy - 15;  // orientation patch origin
    const bool ori_dw = ((reinterpret_cast<uintptr_t>(img) | (uintptr_t)pitch) & 3) == 0 && oc0 + 4 * PSL_ORI_PD <= pitch;
    uint32_t pv[7], ov[5];
#pragma unroll
    for (int t = 0; t < 7; ++t) {
        const int k = lane + 64 * t;
        const int r = k / PSL_DESC_PD, d = k - r * PSL_DESC_PD;
        int gy = pr0 + r;
        gy = gy < 0 ? 0 : (gy >= L.h ? L.h - 1 : gy);  // never sampled; keeps the loads inside the level
        const int gx = min(pc0 + d * 4, L.pitch - 4);
        pv[t] = k < 37 * PSL_DESC_PD ? *reinterpret_cast<const uint32_t*>(bl + (size_t)gy * L.pitch + gx) : 0u;
    }
    if (ori_dw) {
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int k = lane + 64 * t;
            const int r = k / PSL_ORI_PD, d = k - r * PSL_ORI_PD;
            ov[t] = k < 31 * PSL_ORI_PD ? *reinterpret_cast<const uint32_t*>(img + (size_t)(or0 + r) * pitch + oc0 + d * 4) : 0u;
        }
    }
#pragma unroll
    for (int t = 0; t < 7; ++t) { const int k = lane + 64 * t; if (k < 37 * PSL_DESC_PD) patch[k] = pv[t]; }
    if (ori_dw) {
#pragma unroll
        for (int t = 0; t < 5; ++t) { const int k = lane + 64 * t; if (k < 31 * PSL_ORI_PD) ori[k] = ov[t]; }
    } else {  // caller's level-0 image with unaligned rows: bytes
        for (int k = lane; k < 31 * 31; k += 64) {
            const int r = k / 31, c = k - r * 31;
            reinterpret_cast<uint8_t*>(ori)[r * (PSL_ORI_PD * 4) + (cx - 15 - oc0) + c] = img[(size_t)(or0 + r) * pitch + cx - 15 + c];
        }
    }
    __builtin_amdgcn_wave_barrier();

    // intensity centroid over the radius-15 disc: lanes 0..30 take column u = lane-15 for v >= 0,
    // lanes 32..62 the same column for v < 0
    int m10 = 0, m01 = 0;
    {
        const int col = lane & 31;
        if (col < 31) {
            const int u = col - 15, au = u < 0 ? -u : u;
            const uint8_t* c = reinterpret_cast<const uint8_t*>(ori) + 15 * (PSL_ORI_PD * 4) + (cx - oc0) + u;
            int colsum = 0, vsum = 0;
            if (lane < 32) {
#pragma unroll
                for (int v = 0; v <= 15; ++v)
                    if (au <= P.umax[v]) { const int val = c[v * (PSL_ORI_PD * 4)]; colsum += val; vsum += v * val; }
            } else {
#pragma unroll
                for (int v = 1; v <= 15; ++v)
                    if (au <= P.umax[v]) { const int val = c[-v * (PSL_ORI_PD * 4)]; colsum += val; vsum -= v * val; }
            }
            m10 = u * colsum;
            m01 = vsum;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m10 += __shfl_xor(m10, o); m01 += __shfl_xor(m01, o); }
    }
    const float angle = psl_fast_atan2((float)m01, (float)m10);

    // rBRIEF: lane i evaluates comparisons 4i..4i+3
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float a, b;
    psl_sincosf(PSL_FMUL(angle, factorPI), &b, &a);  // a = cos, b = sin
    const uint8_t* pb = reinterpret_cast<const uint8_t*>(patch) + 18 * (PSL_DESC_PD * 4) + (cx - pc0);
    uint32_t nib = 0;
#pragma unroll
    for (int cmp = 0; cmp < 4; ++cmp) {
        const int8_t* pt = &c_orb_pattern[(lane * 4 + cmp) * 4];
        const float x0 = (float)pt[0], y0 = (float)pt[1], x1 = (float)pt[2], y1 = (float)pt[3];
        const int r0 = psl_cvround_f(PSL_FADD(PSL_FMUL(x0, b), PSL_FMUL(y0, a)));
        const int c0 = psl_cvround_f(PSL_FSUB(PSL_FMUL(x0, a), PSL_FMUL(y0, b)));
        const int r1 = psl_cvround_f(PSL_FADD(PSL_FMUL(x1, b), PSL_FMUL(y1, a)));
        const int c1 = psl_cvround_f(PSL_FSUB(PSL_FMUL(x1, a), PSL_FMUL(y1, b)));
        const int t0 = pb[r0 * (PSL_DESC_PD * 4) + c0], t1 = pb[r1 * (PSL_DESC_PD * 4) + c1];
        nib |= (uint32_t)(t0 < t1) << cmp;
    }
    uint32_t v = nib << ((lane & 1) * 4);           // byte lane/2
    v |= __shfl_xor(v, 1);
    v <<= ((lane >> 1) & 3) * 8;                    // dword lane/8
    v |= __shfl_xor(v, 2);
    v |= __shfl_xor(v, 4);
    const size_t o = (size_t)frame * P.out_cap + slot;
    if ((lane & 7) == 0) reinterpret_cast<uint32_t*>(desc + o * 32)[lane >> 3] = v;
    if (lane < 7) {
        float f;
        const float px = (float)cx, py = (float)cy;
        switch (lane) {
            case 0: f = level ? PSL_FMUL(px, L.scale) : px; break;
            case 1: f = level ? PSL_FMUL(py, L.scale) : py; break;
            case 2: f = L.kpsize; break;
            case 3: f = angle; break;
            case 4: f = (float)(key >> 24); break;
            case 5: f = __int_as_float(level); break;
            default: f = __int_as_float(-1); break;
        }
        reinterpret_cast<float*>(kps + o)[lane] = f;
    }
}

#endif
