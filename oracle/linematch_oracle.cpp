// ORACLE — test infrastructure only (see orb_oracle.cpp header). CPU restatement of the grid-guided
// line matchers:
//   LineIterator (Bresenham on the 64x48 grid)   add_src/lineIterator.cpp:34-77
//   Frame::AssignFeaturesToGridForLine           src/Frame.cc:286-309
//   Frame::GetFeaturesInAreaForLine              src/Frame.cc:752-826
//   LSDmatcher::SearchByProjection(cur,last,th)  add_src/LSDmatcher.cpp:112-215   (mode 0)
//   LSDmatcher::SearchByProjection(F,MLs,..,th)  add_src/LSDmatcher.cpp:260-352   (mode 1)
// from the projected map lines on (isInFrustum / pose algebra stay host logic of Tracking).
// PARITY UNPINNED (no fixtures upstream).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <set>
#include <vector>

#include "psl_oracle.h"

namespace {
const int GC = 64, GR = 48;

struct LineIt {
    bool steep; double x1, y1, x2, y2, dx, dy, error; int maxX, ystep, y, x;
    LineIt(double x1_, double y1_, double x2_, double y2_) : steep(std::abs(y2_ - y1_) > std::abs(x2_ - x1_)), x1(x1_), y1(y1_), x2(x2_), y2(y2_) {
        if (steep) { std::swap(x1, y1); std::swap(x2, y2); }
        if (x1 > x2) { std::swap(x1, x2); std::swap(y1, y2); }
        dx = x2 - x1; dy = std::abs(y2 - y1);
        error = dx / 2.0; ystep = (y1 < y2) ? 1 : -1;
        x = static_cast<int>(x1); y = static_cast<int>(y1); maxX = static_cast<int>(x2);
    }
    bool next(int& px, int& py) {
        if (x > maxX) return false;
        if (steep) { px = y; py = x; } else { px = x; py = y; }
        error -= dy;
        if (error < 0) { y += ystep; error += dx; }
        x++;
        return true;
    }
};

struct LGrid {
    std::vector<int> cell[GC][GR];
    float minX, minY, invW, invH;
    void build(const PsoKeyLine* k, int n, float mnx, float mny, float mxx, float mxy) {
        minX = mnx; minY = mny;
        invW = static_cast<float>(GC) / static_cast<float>(mxx - mnx);
        invH = static_cast<float>(GR) / static_cast<float>(mxy - mny);
        for (int i = 0; i < n; ++i) {
            LineIt it(k[i].startPointX * invW, k[i].startPointY * invH, k[i].endPointX * invW, k[i].endPointY * invH);
            int px, py;
            while (it.next(px, py))
                if (px >= 0 && px < GC && py >= 0 && py < GR) cell[px][py].push_back(i);
        }
    }
    std::vector<int> area(const PsoKeyLine* k, const double* eq, float x1, float y1, float x2, float y2, float r, float TH) const {
        std::vector<int> out;
        std::set<int> seen;
        float x[3] = {x1, (float)((x1 + x2) / 2.0), x2};
        float y[3] = {y1, (float)((y1 + y2) / 2.0), y2};
        float d1x = x1 - x2, d1y = y1 - y2;
        float n1 = std::sqrt(d1x * d1x + d1y * d1y);
        d1x /= n1; d1y /= n1;
        for (int i = 0; i < 3; i++) {
            const int nMinCellX = std::max(0, (int)std::floor((x[i] - minX - r) * invW));
            if (nMinCellX >= GC) continue;
            const int nMaxCellX = std::min(GC - 1, (int)std::ceil((x[i] - minX + r) * invW));
            if (nMaxCellX < 0) continue;
            const int nMinCellY = std::max(0, (int)std::floor((y[i] - minY - r) * invH));
            if (nMinCellY >= GR) continue;
            const int nMaxCellY = std::min(GR - 1, (int)std::ceil((y[i] - minY + r) * invH));
            if (nMaxCellY < 0) continue;
            for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
                for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
                    for (int j : cell[ix][iy]) {
                        if (seen.count(j)) continue;
                        float d2x = k[j].startPointX - k[j].endPointX, d2y = k[j].startPointY - k[j].endPointY;
                        float n2 = std::sqrt(d2x * d2x + d2y * d2y);
                        d2x /= n2; d2y /= n2;
                        float CosSita = std::abs(d1x * d2x + d1y * d2y);
                        if (CosSita < TH) continue;
                        const float dist = (float)(eq[3 * j] * x[i] + eq[3 * j + 1] * y[i] + eq[3 * j + 2]);
                        if (std::fabs(dist) < r) { out.push_back(j); seen.insert(j); }
                    }
        }
        return out;
    }
};

int hamming(const uint8_t* a, const uint8_t* b) { return pso_hamming256(a, b); }
}  // namespace

extern "C" {

// the Bresenham walk alone (tap for tests/test_oracle_ref_cpu.py, which compares it with the reference's own lineIterator.cpp)
int pso_line_iterator_walk(double x1, double y1, double x2, double y2, int* xy, int cap) {
    LineIt it(x1, y1, x2, y2);
    int px, py, n = 0;
    while (it.next(px, py)) {
        if (n < cap) { xy[2 * n] = px; xy[2 * n + 1] = py; }
        ++n;
    }
    return n;
}

// CSR of mGridForLine, cell = ix*48+iy; idx capacity >= sum over lines of visited cells
int pso_line_grid_build(const PsoKeyLine* k, int n, float minX, float minY, float maxX, float maxY, int* start, int* idx, int cap) {
    LGrid* g = new LGrid();
    g->build(k, n, minX, minY, maxX, maxY);
    int p = 0;
    for (int ix = 0; ix < GC; ++ix)
        for (int iy = 0; iy < GR; ++iy) {
            start[ix * GR + iy] = p;
            for (int j : g->cell[ix][iy]) { if (p < cap) idx[p] = j; ++p; }
        }
    start[GC * GR] = p;
    delete g;
    return p;
}

int pso_line_search_by_projection(const PsoKeyLine* k, const uint8_t* desc, const double* eq, const double* dir3d, int n, float minX,
                                  float minY, float maxX, float maxY, const PsoLineQuery* q, const uint8_t* qdesc, int nq,
                                  const uint8_t* taken, int mode, float nnratio, int* match, int* assigned) {
    LGrid* g = new LGrid();
    g->build(k, n, minX, minY, maxX, maxY);
    std::vector<int> owner(n, -1);
    std::vector<char> blocked(n, 0);
    for (int i = 0; i < n; ++i) blocked[i] = taken ? taken[i] != 0 : 0;
    const double cos10 = std::cos(10.0 / 180.0 * M_PI), cos15 = std::cos(15.0 / 180.0 * M_PI);
    int nmatches = 0;
    for (int i = 0; i < nq; ++i) {
        match[i] = -1;
        const std::vector<int> cand = g->area(k, eq, q[i].x1, q[i].y1, q[i].x2, q[i].y2, q[i].radius, q[i].th_cos);
        if (cand.empty()) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int i2 : cand) {
            if (blocked[i2]) continue;
            if (mode == 0) {
                const double vc0 = k[i2].ePointInOctaveX - k[i2].sPointInOctaveX, vc1 = k[i2].ePointInOctaveY - k[i2].sPointInOctaveY;
                const double vl0 = q[i].vx, vl1 = q[i].vy;
                const double dot = vc0 * vl0 + vc1 * vl1;
                const double angle = std::abs(dot / (std::sqrt(vc0 * vc0 + vc1 * vc1) * std::sqrt(vl0 * vl0 + vl1 * vl1)));
                if (angle < cos10) continue;
                const int dist = hamming(qdesc + (size_t)i * 32, desc + (size_t)i2 * 32);
                float max_ = std::max(q[i].length, k[i2].lineLength), min_ = std::min(q[i].length, k[i2].lineLength);
                if (min_ / max_ < 0.75) continue;
                if (dist < bestDist) { bestDist = dist; bestIdx = i2; }
            } else {
                const double* f = dir3d + 3 * (size_t)i2;
                const double* w = q[i].wdir;
                float dot = (float)(f[0] * w[0] + f[1] * w[1] + f[2] * w[2]);
                float mag_f = (float)std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
                float mag_ml = (float)std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
                float angle = std::abs(dot / (mag_f * mag_ml));
                if (angle < cos15) continue;
                const int dist = hamming(qdesc + (size_t)i * 32, desc + (size_t)i2 * 32);
                if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = k[i2].octave; bestIdx = i2; }
                else if (dist < bestDist2) { bestLevel2 = k[i2].octave; bestDist2 = dist; }
            }
        }
        if (bestDist <= 95) {
            if (mode == 1 && bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            owner[bestIdx] = i;
            blocked[bestIdx] = q[i].blocks != 0;
            match[i] = bestIdx;
            nmatches++;
        }
    }
    if (assigned) for (int i = 0; i < n; ++i) assigned[i] = owner[i];
    delete g;
    return nmatches;
}

}  // extern "C"
