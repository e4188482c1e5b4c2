// ORACLE — test infrastructure only (see orb_oracle.cpp header). CPU restatement of the
// descriptor-matching part of the path, written sequentially exactly as the reference runs it:
//   ORBmatcher::DescriptorDistance            src/ORBmatcher.cc:1647-1663 (SWAR popcount)
//   Frame::AssignFeaturesToGrid / PosInGrid   src/Frame.cc:269-284, 1040-1050
//   Frame::GetFeaturesInArea                  src/Frame.cc:985-1038
//   ORBmatcher::SearchByProjection(cur,last)  src/ORBmatcher.cc:1328-1470 (from the projected
//        queries on; the 3-D projection itself is host logic of Tracking)
//   ORBmatcher::SearchByProjection(F,MPs)     src/ORBmatcher.cc:45-129
//   ORBmatcher::ComputeThreeMaxima            src/ORBmatcher.cc:1601-1645
//   cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) Appendix A.7; LSDmatcher::matchNNR add_src/LSDmatcher.cpp:354-376
// PARITY UNPINNED: the reference holds no fixtures for these functions.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "psl_oracle.h"

namespace {

const int GRID_COLS = 64, GRID_ROWS = 48;  // include/Frame.h:45-46
const int TH_HIGH = 100, HISTO_LENGTH = 30;

int descriptor_distance(const uint8_t* a, const uint8_t* b) {
    const int32_t* pa = (const int32_t*)a;
    const int32_t* pb = (const int32_t*)b;
    int dist = 0;
    for (int i = 0; i < 8; i++, pa++, pb++) {
        unsigned int v = *pa ^ *pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

struct Grid {
    float minX, minY, maxX, maxY, invW, invH;
    std::vector<int> cell[GRID_COLS][GRID_ROWS];
    const PsoKeyPoint* kps;
    int n;

    void build(const PsoKeyPoint* k, int nn, float mnx, float mny, float mxx, float mxy) {
        kps = k; n = nn; minX = mnx; minY = mny; maxX = mxx; maxY = mxy;
        invW = static_cast<float>(GRID_COLS) / static_cast<float>(maxX - minX);
        invH = static_cast<float>(GRID_ROWS) / static_cast<float>(maxY - minY);
        for (int i = 0; i < n; ++i) {
            int posX = (int)std::round((k[i].x - minX) * invW);
            int posY = (int)std::round((k[i].y - minY) * invH);
            if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) continue;
            cell[posX][posY].push_back(i);
        }
    }

    std::vector<int> area(float x, float y, float r, int minLevel, int maxLevel) const {
        std::vector<int> out;
        const int nMinCellX = std::max(0, (int)std::floor((x - minX - r) * invW));
        if (nMinCellX >= GRID_COLS) return out;
        const int nMaxCellX = std::min(GRID_COLS - 1, (int)std::ceil((x - minX + r) * invW));
        if (nMaxCellX < 0) return out;
        const int nMinCellY = std::max(0, (int)std::floor((y - minY - r) * invH));
        if (nMinCellY >= GRID_ROWS) return out;
        const int nMaxCellY = std::min(GRID_ROWS - 1, (int)std::ceil((y - minY + r) * invH));
        if (nMaxCellY < 0) return out;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
                for (int j : cell[ix][iy]) {
                    const PsoKeyPoint& kp = kps[j];
                    if (bCheckLevels) {
                        if (kp.octave < minLevel) continue;
                        if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                    }
                    const float distx = kp.x - x, disty = kp.y - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) out.push_back(j);
                }
        return out;
    }
};

void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

}  // namespace

extern "C" {

int pso_hamming256(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }

// CSR of the grid in GetFeaturesInArea visiting order (cell = ix*48+iy). start: 3073 ints.
int pso_grid_build(const PsoKeyPoint* kps, int n, float minX, float minY, float maxX, float maxY, int* start, int* idx) {
    Grid* g = new Grid();
    g->build(kps, n, minX, minY, maxX, maxY);
    int p = 0;
    for (int ix = 0; ix < GRID_COLS; ++ix)
        for (int iy = 0; iy < GRID_ROWS; ++iy) {
            start[ix * GRID_ROWS + iy] = p;
            for (int j : g->cell[ix][iy]) idx[p++] = j;
        }
    start[GRID_COLS * GRID_ROWS] = p;
    delete g;
    return p;
}

int pso_search_by_projection_last(const PsoKeyPoint* kps, const uint8_t* desc, const float* uright, int n, float minX,
                                  float minY, float maxX, float maxY, const PsoProjQuery* q, const uint8_t* qdesc, int nq,
                                  const uint8_t* taken, int checkOri, int* match, int* assigned) {
    Grid* g = new Grid();
    g->build(kps, n, minX, minY, maxX, maxY);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> owner(n, -1);      // CurrentFrame.mvpMapPoints[i2] as a query index
    std::vector<char> blocked(n, 0);    // mvpMapPoints[i2] && Observations()>0
    for (int i = 0; i < n; ++i) blocked[i] = taken ? (taken[i] != 0) : 0;
    std::vector<std::pair<int, int>> histEntries[HISTO_LENGTH];  // (query, keypoint)
    for (int i = 0; i < nq; ++i) {
        match[i] = -1;
        const std::vector<int> cand = g->area(q[i].u, q[i].v, q[i].radius, q[i].min_level, q[i].max_level);
        if (cand.empty()) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : cand) {
            if (blocked[i2]) continue;
            const float ur_i2 = uright ? uright[i2] : -1.f;
            if (ur_i2 > 0) {
                const float er = std::fabs(q[i].ur - ur_i2);
                if (er > q[i].radius) continue;
            }
            const int dist = descriptor_distance(qdesc + (size_t)i * 32, desc + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            owner[bestIdx2] = i;
            blocked[bestIdx2] = q[i].blocks != 0;
            match[i] = bestIdx2;
            nmatches++;
            if (checkOri) {
                float rot = q[i].angle - kps[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
                histEntries[bin].push_back(std::make_pair(i, bestIdx2));
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int b = 0; b < HISTO_LENGTH; b++)
            if (b != ind1 && b != ind2 && b != ind3)
                for (auto& e : histEntries[b]) {
                    owner[e.second] = -1;
                    match[e.first] = -1;
                    nmatches--;
                }
    }
    if (assigned) for (int i = 0; i < n; ++i) assigned[i] = owner[i];
    delete g;
    return nmatches;
}

int pso_search_by_projection_map(const PsoKeyPoint* kps, const uint8_t* desc, const float* uright, int n, float minX,
                                 float minY, float maxX, float maxY, const PsoProjQuery* q, const uint8_t* qdesc, int nq,
                                 const uint8_t* taken, float nnratio, int* match, int* assigned) {
    Grid* g = new Grid();
    g->build(kps, n, minX, minY, maxX, maxY);
    int nmatches = 0;
    std::vector<int> owner(n, -1);
    std::vector<char> blocked(n, 0);
    for (int i = 0; i < n; ++i) blocked[i] = taken ? (taken[i] != 0) : 0;
    for (int i = 0; i < nq; ++i) {
        match[i] = -1;
        const std::vector<int> cand = g->area(q[i].u, q[i].v, q[i].radius, q[i].min_level, q[i].max_level);
        if (cand.empty()) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : cand) {
            if (blocked[idx]) continue;
            const float ur = uright ? uright[idx] : -1.f;
            if (ur > 0) {
                const float er = std::fabs(q[i].ur - ur);
                if (er > q[i].radius) continue;
            }
            const int dist = descriptor_distance(qdesc + (size_t)i * 32, desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kps[idx].octave; bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = kps[idx].octave; bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
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

// BFMatcher(NORM_HAMMING, crossCheck=false).knnMatch(k=2): ascending distance, lower train index
// first on ties; missing neighbours = (-1, 0x7fffffff).
void pso_hamming_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int* idx, int* dist) {
    for (int i = 0; i < nq; ++i) {
        int b0 = 0x7fffffff, b1 = 0x7fffffff, i0 = -1, i1 = -1;
        for (int j = 0; j < nt; ++j) {
            const int d = descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
            if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = j; }
            else if (d < b1) { b1 = d; i1 = j; }
        }
        idx[2 * i] = i0; idx[2 * i + 1] = i1; dist[2 * i] = b0; dist[2 * i + 1] = b1;
    }
}

int pso_line_match_nnr(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnr, int* matches12) {
    std::vector<int> idx(2 * (size_t)std::max(n1, 1)), dist(2 * (size_t)std::max(n1, 1));
    pso_hamming_knn2(d1, n1, d2, n2, idx.data(), dist.data());
    int matches = 0;
    for (int i = 0; i < n1; ++i) {
        matches12[i] = -1;
        if (n2 < 2) continue;  // reference indexes matches_[idx][1] out of bounds (:369): defined as no match
        if ((float)dist[2 * i] < (float)dist[2 * i + 1] * nnr) { matches12[i] = idx[2 * i]; matches++; }
    }
    return matches;
}

}  // extern "C"

// ---- Frame post-processing between extraction and matching ----------------
//   cvtColor RGB/BGR -> GRAY (src/Tracking.cc:219-232): OpenCV 8-bit fixed point, (R*4899 + G*9617 + B*1868 + 8192) >> 14
//   depth.convertTo(CV_32F, factor) (src/Tracking.cc:234-235): (float)d * (float)factor
//   Frame::UndistortKeyPoints (src/Frame.cc:1062-1092) -> cv::undistortPoints(mat, mat, K, dist, Mat(), K): 5 fixed
//        iterations in double, restated from the OpenCV 3.2 algorithm (Appendix A; not in the reference tree)
//   Frame::ComputeImageBounds (src/Frame.cc:1135-1168), Frame::ComputeStereoFromRGBD (src/Frame.cc:1342-1363)
namespace {
void undistort_point(double u, double v, const double* K /*fx,fy,cx,cy*/, const double* k /*k1,k2,p1,p2,k3*/, float* ox, float* oy) {
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3], ifx = 1. / fx, ify = 1. / fy;
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + 0 * r2 + 0 * r2 * r2;
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + 0 * r2 + 0 * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0 * y + cx, yy = 0 * x + fy * y + cy, ww = 1. / (0 * x + 0 * y + 1);
    *ox = (float)(xx * ww);
    *oy = (float)(yy * ww);
}
}  // namespace

extern "C" {

void pso_rgb_to_gray(const uint8_t* rgb, int w, int h, int stride, int is_rgb, uint8_t* gray) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const uint8_t* p = rgb + (size_t)y * stride + 3 * x;
            const int r = is_rgb ? p[0] : p[2], g = p[1], b = is_rgb ? p[2] : p[0];
            gray[(size_t)y * w + x] = (uint8_t)((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14);
        }
}

void pso_depth_to_float(const uint16_t* d, int n, float factor, float* out) {
    for (int i = 0; i < n; ++i) out[i] = (float)d[i] * factor;
}

// bounds: mnMinX, mnMinY, mnMaxX, mnMaxY
void pso_image_bounds(int cols, int rows, const float* K, const float* dist, float* bounds) {
    if (dist[0] != 0.0f) {
        const double Kd[4] = {K[0], K[1], K[2], K[3]}, kd[5] = {dist[0], dist[1], dist[2], dist[3], dist[4]};
        float cx[4], cy[4];
        const float src[4][2] = {{0.f, 0.f}, {(float)cols, 0.f}, {0.f, (float)rows}, {(float)cols, (float)rows}};
        for (int i = 0; i < 4; ++i) undistort_point(src[i][0], src[i][1], Kd, kd, &cx[i], &cy[i]);
        bounds[0] = std::min(cx[0], cx[2]); bounds[2] = std::max(cx[1], cx[3]);
        bounds[1] = std::min(cy[0], cy[1]); bounds[3] = std::max(cy[2], cy[3]);
    } else { bounds[0] = 0.0f; bounds[2] = (float)cols; bounds[1] = 0.0f; bounds[3] = (float)rows; }
}

// mvKeysUn, mvDepth, mvuRight from mvKeys and the float depth image
void pso_frame_post_rgbd(const PsoKeyPoint* kps, int n, const float* depth, int w, int h, int dstride /*floats*/, const float* K, const float* dist,
                         float mbf, PsoKeyPoint* kpsUn, float* mvDepth, float* mvuRight) {
    const double Kd[4] = {K[0], K[1], K[2], K[3]}, kd[5] = {dist[0], dist[1], dist[2], dist[3], dist[4]};
    for (int i = 0; i < n; ++i) {
        kpsUn[i] = kps[i];
        if (dist[0] != 0.0f) undistort_point(kps[i].x, kps[i].y, Kd, kd, &kpsUn[i].x, &kpsUn[i].y);
        mvDepth[i] = -1; mvuRight[i] = -1;
        const int v = (int)kps[i].y, u = (int)kps[i].x;
        const float d = depth[(size_t)v * dstride + u];
        if (d > 0) { mvDepth[i] = d; mvuRight[i] = kpsUn[i].x - mbf / d; }
    }
    (void)w; (void)h;
}

}  // extern "C"

// ---- SURVEY.md §8a row a18 ---------------------------------------------------------------------------------------
//   ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist)   src/ORBmatcher.cc:1472-1599
//        (relocalisation): window search like (cur,last) but every occupied keypoint is skipped, every match occupies,
//        no stereo gate, threshold = ORBdist; projection / sAlreadyFound / scale prediction stay with the caller.
//   ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches)                              src/ORBmatcher.cc:159-288
//        from the point where the two DBoW2 FeatureVectors are walked: the caller (DBoW2 stays on the host) passes, in the
//        reference's iteration order (nodes ascending, vIndicesKF order, bad/NULL map points dropped), one query per
//        keyframe feature: its descriptor, its angle and the run [start, start+len) of the frame's node list
//        (`fidx` = the frame's FeatureVector flattened in node order).
extern "C" {

int pso_search_by_projection_kf(const PsoKeyPoint* kps, const uint8_t* desc, int n, float minX, float minY, float maxX, float maxY,
                                const PsoProjQuery* q, const uint8_t* qdesc, int nq, const uint8_t* taken, int orbDist, int checkOri,
                                int* match, int* assigned) {
    Grid* g = new Grid();
    g->build(kps, n, minX, minY, maxX, maxY);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> owner(n, -1);
    std::vector<char> occupied(n, 0);  // CurrentFrame.mvpMapPoints[i2] != NULL
    for (int i = 0; i < n; ++i) occupied[i] = taken ? (taken[i] != 0) : 0;
    std::vector<std::pair<int, int>> histEntries[HISTO_LENGTH];
    for (int i = 0; i < nq; ++i) {
        match[i] = -1;
        const std::vector<int> cand = g->area(q[i].u, q[i].v, q[i].radius, q[i].min_level, q[i].max_level);
        if (cand.empty()) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : cand) {
            if (occupied[i2]) continue;
            const int dist = descriptor_distance(qdesc + (size_t)i * 32, desc + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= orbDist) {
            owner[bestIdx2] = i;
            occupied[bestIdx2] = 1;
            match[i] = bestIdx2;
            nmatches++;
            if (checkOri) {
                float rot = q[i].angle - kps[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
                histEntries[bin].push_back(std::make_pair(i, bestIdx2));
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int b = 0; b < HISTO_LENGTH; b++)
            if (b != ind1 && b != ind2 && b != ind3)
                for (auto& e : histEntries[b]) { owner[e.second] = -1; match[e.first] = -1; nmatches--; }
    }
    if (assigned) for (int i = 0; i < n; ++i) assigned[i] = owner[i];
    delete g;
    return nmatches;
}

// run[2*i], run[2*i+1]: start and length of query i's candidate run in fidx; qangle: pKF->mvKeysUn[realIdxKF].angle;
// fangle: F.mvKeys[.].angle.  match[i] = frame feature given to query i or -1; assigned[f] = query owning frame feature f.
int pso_search_by_bow(const uint8_t* fdesc, const float* fangle, int nf, const int32_t* fidx, const int32_t* run, const uint8_t* qdesc,
                      const float* qangle, int nq, float nnratio, int checkOri, int* match, int* assigned) {
    const int TH_LOW = 50;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    std::vector<std::pair<int, int>> histEntries[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> owner(nf, -1);  // vpMapPointMatches[realIdxF] as a query index
    for (int i = 0; i < nq; ++i) {
        match[i] = -1;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        for (int p = run[2 * i]; p < run[2 * i] + run[2 * i + 1]; ++p) {
            const int realIdxF = fidx[p];
            if (owner[realIdxF] >= 0) continue;
            const int dist = descriptor_distance(qdesc + (size_t)i * 32, fdesc + (size_t)realIdxF * 32);
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
            else if (dist < bestDist2) { bestDist2 = dist; }
        }
        if (bestDist1 <= TH_LOW) {
            if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                owner[bestIdxF] = i;
                match[i] = bestIdxF;
                if (checkOri) {
                    float rot = qangle[i] - fangle[bestIdxF];
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(bestIdxF);
                    histEntries[bin].push_back(std::make_pair(i, bestIdxF));
                }
                nmatches++;
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int b = 0; b < HISTO_LENGTH; b++)
            if (b != ind1 && b != ind2 && b != ind3)
                for (auto& e : histEntries[b]) { owner[e.second] = -1; match[e.first] = -1; nmatches--; }
    }
    if (assigned) for (int f = 0; f < nf; ++f) assigned[f] = owner[f];
    return nmatches;
}

}  // extern "C"
