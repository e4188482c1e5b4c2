// ORACLE (test infrastructure only; never linked into or called by the product): CPU restatement of the
// KeyFrame-rate matchers of SURVEY.md §8(f) rank 3, from the point where the reference has projected its map points.
//   ORBmatcher::Fuse(pKF, vpMapPoints, th)                      src/ORBmatcher.cc:825-966   (inner loop :893-948)
//   ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)    src/ORBmatcher.cc:968-1100  (inner loop :1058-1078)
//   ORBmatcher::SearchBySim3                                    src/ORBmatcher.cc:1102-1326
//   ORBmatcher::SearchForTriangulation / CheckDistEpipolarLine  src/ORBmatcher.cc:657-823, 140-157
//   KeyFrame::GetFeaturesInArea / GetLinesInArea                src/KeyFrame.cc:685-724, 857-891
//   LSDmatcher::Fuse                                            add_src/LSDmatcher.cpp:847-984 (inner loop :933-958)
//   MapPoint::ComputeDistinctiveDescriptors (and MapLine's)     src/MapPoint.cc:242-304, add_src/MapLine.cpp:250-310
// PARITY UNPINNED: the reference holds no fixtures for these functions and cannot be built here.
#include <math.h>
#include <stdint.h>
#include <limits.h>

#include <algorithm>
#include <cmath>
#include <utility>
#include <vector>

#include "psl_oracle.h"

namespace {

const int GRID_COLS = 64, GRID_ROWS = 48;  // include/Frame.h:45-46 (KeyFrame copies the frame's grid, src/KeyFrame.cc:30-60)
const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;

int descriptor_distance(const uint8_t* a, const uint8_t* b) {  // src/ORBmatcher.cc:1647-1663
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

struct KfGrid {  // mGrid of a KeyFrame + GetFeaturesInArea (src/KeyFrame.cc:685-724)
    float minX, minY, invW, invH;
    std::vector<int> cell[GRID_COLS][GRID_ROWS];
    const PsoKeyPoint* kps;
    KfGrid(const PsoKeyPoint* k, int n, const float* b) : kps(k) {
        minX = b[0]; minY = b[1];
        invW = static_cast<float>(GRID_COLS) / static_cast<float>(b[2] - b[0]);
        invH = static_cast<float>(GRID_ROWS) / static_cast<float>(b[3] - b[1]);
        for (int i = 0; i < n; ++i) {  // Frame::PosInGrid src/Frame.cc:1040-1050
            const int posX = (int)std::round((k[i].x - minX) * invW), posY = (int)std::round((k[i].y - minY) * invH);
            if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) continue;
            cell[posX][posY].push_back(i);
        }
    }
    std::vector<int> area(float x, float y, float r) const {
        std::vector<int> out;
        const int nMinCellX = std::max(0, (int)std::floor((x - minX - r) * invW));
        if (nMinCellX >= GRID_COLS) return out;
        const int nMaxCellX = std::min(GRID_COLS - 1, (int)std::ceil((x - minX + r) * invW));
        if (nMaxCellX < 0) return out;
        const int nMinCellY = std::max(0, (int)std::floor((y - minY - r) * invH));
        if (nMinCellY >= GRID_ROWS) return out;
        const int nMaxCellY = std::min(GRID_ROWS - 1, (int)std::ceil((y - minY + r) * invH));
        if (nMaxCellY < 0) return out;
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
                for (int j : cell[ix][iy]) {
                    const float distx = kps[j].x - x, disty = kps[j].y - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) out.push_back(j);
                }
        return out;
    }
};

// the candidate loop shared by both Fuse variants and both directions of SearchBySim3; q.max_level = nPredictedLevel
void window_best(const KfGrid& G, const uint8_t* desc, const float* uright, const PsoProjQuery* q, const uint8_t* qdesc, int nq, int chi2,
                 const float* invSigma2, int* best_idx, int* best_dist) {
    for (int i = 0; i < nq; ++i) {
        best_idx[i] = -1;
        best_dist[i] = INT_MAX;
        if (!(q[i].radius >= 0)) continue;  // the host dropped this map point before the window search
        const float u = q[i].u, v = q[i].v, ur = q[i].ur;
        const int nPredictedLevel = q[i].max_level;
        int bestDist = INT_MAX, bestIdx = -1;
        for (int idx : G.area(u, v, q[i].radius)) {
            const PsoKeyPoint& kp = G.kps[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (chi2) {  // src/ORBmatcher.cc:907-934
                if (uright[idx] >= 0) {
                    const float ex = u - kp.x, ey = v - kp.y, er = ur - uright[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * invSigma2[kpLevel] > 7.8) continue;
                } else {
                    const float ex = u - kp.x, ey = v - kp.y;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * invSigma2[kpLevel] > 5.99) continue;
                }
            }
            const int dist = descriptor_distance(qdesc + (size_t)i * 32, desc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        best_idx[i] = bestIdx;
        best_dist[i] = bestDist;
    }
}

void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {  // src/ORBmatcher.cc:1601-1645
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

// bounds = mnMinX, mnMinY, mnMaxX, mnMaxY.  best_dist = INT_MAX where no candidate passed (the first Fuse starts from 256,
// which changes nothing: its decision is bestDist <= TH_LOW).
void pso_window_best(const PsoKeyPoint* kps, const uint8_t* desc, const float* uright, int n, const float* bounds, const PsoProjQuery* q,
                     const uint8_t* qdesc, int nq, int chi2, const float* invSigma2, int* best_idx, int* best_dist) {
    KfGrid G(kps, n, bounds);
    window_best(G, desc, uright, q, qdesc, nq, chi2, invSigma2, best_idx, best_dist);
}

// q12[i1]: map point i1 of KF1 projected into KF2 (radius < 0: no map point / already matched / bad / a gate failed), qdesc1 its
// descriptor (pMP->GetDescriptor()); q21 likewise.  match12[i1] = idx2 where both directions agree, else -1 (:1307-1323).
int pso_search_by_sim3(const PsoKeyPoint* kps1, const uint8_t* desc1, int n1, const float* bounds1, const PsoKeyPoint* kps2,
                       const uint8_t* desc2, int n2, const float* bounds2, const PsoProjQuery* q12, const uint8_t* qdesc1,
                       const PsoProjQuery* q21, const uint8_t* qdesc2, int* match12) {
    KfGrid G1(kps1, n1, bounds1), G2(kps2, n2, bounds2);
    std::vector<int> b1(n1), d1(n1), b2(n2), d2(n2), vnMatch1(n1, -1), vnMatch2(n2, -1);
    window_best(G2, desc2, nullptr, q12, qdesc1, n1, 0, nullptr, b1.data(), d1.data());
    window_best(G1, desc1, nullptr, q21, qdesc2, n2, 0, nullptr, b2.data(), d2.data());
    for (int i = 0; i < n1; ++i) if (d1[i] <= TH_HIGH) vnMatch1[i] = b1[i];
    for (int i = 0; i < n2; ++i) if (d2[i] <= TH_HIGH) vnMatch2[i] = b2[i];
    int nFound = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        match12[i1] = -1;
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            const int idx1 = vnMatch2[idx2];
            if (idx1 == i1) { match12[i1] = idx2; nFound++; }
        }
    }
    return nFound;
}

// One query per feature of KF1 in the reference's iteration order (common nodes ascending, f1it->second order; features with a
// map point and, under bOnlyStereo, non-stereo ones dropped): PsoTriQuery.  KF2: mvKeysUn, mvuRight, descriptors, taken2[idx2] =
// (pKF2->GetMapPoint(idx2) != NULL), fidx2 = its FeatureVector flattened in node order.  F12 row-major 3x3.
int pso_search_for_triangulation(const PsoKeyPoint* kps2, const uint8_t* desc2, const float* uright2, const uint8_t* taken2,
                                 const int32_t* fidx2, const PsoTriQuery* q, const uint8_t* qdesc, int nq, const float* F12, float ex,
                                 float ey, int bOnlyStereo, int checkOri, const float* scaleFactors, const float* levelSigma2, int* match) {
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < nq; ++i) {
        match[i] = -1;
        const bool bStereo1 = q[i].stereo != 0;
        const float kp1x = q[i].x, kp1y = q[i].y;
        int bestDist = TH_LOW, bestIdx2 = -1;
        for (int p = q[i].start; p < q[i].start + q[i].len; ++p) {
            const int idx2 = fidx2[p];
            if (taken2[idx2]) continue;  // vbMatched2 is never set in the reference (:686, :715)
            const bool bStereo2 = uright2[idx2] >= 0;
            if (bOnlyStereo && !bStereo2) continue;
            const int dist = descriptor_distance(qdesc + (size_t)i * 32, desc2 + (size_t)idx2 * 32);
            if (dist > TH_LOW || dist > bestDist) continue;
            const PsoKeyPoint& kp2 = kps2[idx2];
            if (!bStereo1 && !bStereo2) {
                const float distex = ex - kp2.x, distey = ey - kp2.y;
                if (distex * distex + distey * distey < 100 * scaleFactors[kp2.octave]) continue;
            }
            // CheckDistEpipolarLine :140-157
            const float a = kp1x * F12[0] + kp1y * F12[3] + F12[6];
            const float b = kp1x * F12[1] + kp1y * F12[4] + F12[7];
            const float c = kp1x * F12[2] + kp1y * F12[5] + F12[8];
            const float num = a * kp2.x + b * kp2.y + c;
            const float den = a * a + b * b;
            if (den == 0) continue;
            const float dsqr = num * num / den;
            if (dsqr < 3.84 * levelSigma2[kp2.octave]) { bestIdx2 = idx2; bestDist = dist; }
        }
        if (bestIdx2 >= 0) {
            match[i] = bestIdx2;
            nmatches++;
            if (checkOri) {
                float rot = q[i].angle - kps2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(i);
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int b = 0; b < HISTO_LENGTH; b++)
            if (b != ind1 && b != ind2 && b != ind3)
                for (int i : rotHist[b]) { match[i] = -1; nmatches--; }
    }
    return nmatches;
}

// LSDmatcher::Fuse after the projection: GetLinesInArea(u1, v1, u2, v2, radius, TH = 0.998) then the best descriptor among
// octaves nPredictedLevel-1 .. nPredictedLevel.  `desc` (ndesc rows) is whatever the caller hands over as pKF->mDescriptors
// (:945 reads the ORB descriptor matrix with a LINE index); a line index without a row is skipped (the reference reads out of
// bounds there).
void pso_line_fuse_best(const PsoKeyLine* kls, int n, const uint8_t* desc, int ndesc, const PsoLineFuseQuery* q, const uint8_t* qdesc, int nq,
                        int* best_idx, int* best_dist) {
    const float TH = 0.998;
    for (int i = 0; i < nq; ++i) {
        best_idx[i] = -1;
        best_dist[i] = 256;
        if (!(q[i].radius >= 0)) continue;
        const float x1 = q[i].x1, y1 = q[i].y1, x2 = q[i].x2, y2 = q[i].y2, r = q[i].radius;
        float delta1x = x1 - x2, delta1y = y1 - y2;
        const float norm_delta1 = std::sqrt(delta1x * delta1x + delta1y * delta1y);
        delta1x /= norm_delta1;
        delta1y /= norm_delta1;
        int bestDist = 256, bestIdx = -1;
        for (int k = 0; k < n; ++k) {
            const PsoKeyLine& kl = kls[k];
            const float distance = (0.5 * (x1 + x2) - kl.pt_x) * (0.5 * (x1 + x2) - kl.pt_x) + (0.5 * (y1 + y2) - kl.pt_y) * (0.5 * (y1 + y2) - kl.pt_y);
            if (distance > r * r) continue;
            float delta2x = kl.startPointX - kl.endPointX, delta2y = kl.startPointY - kl.endPointY;
            const float norm_delta2 = std::sqrt(delta2x * delta2x + delta2y * delta2y);
            delta2x /= norm_delta2;
            delta2y /= norm_delta2;
            const float CosSita = std::abs(delta1x * delta2x + delta1y * delta2y);
            if (CosSita < TH) continue;
            if (kl.octave < q[i].level - 1 || kl.octave > q[i].level) continue;
            if (k >= ndesc) continue;
            const int dist = descriptor_distance(qdesc + (size_t)i * 32, desc + (size_t)k * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = k; }
        }
        best_idx[i] = bestIdx;
        best_dist[i] = bestDist;
    }
}

// offsets[p] .. offsets[p+1]: the observed descriptors of map point / map line p (those of non-bad keyframes, in the order of the
// std::map iteration).  best[p] = index (within the run) of the descriptor with the least median distance, -1 for an empty run.
void pso_distinctive_descriptors(const uint8_t* desc, const int32_t* offsets, int npts, int* best) {
    for (int p = 0; p < npts; ++p) {
        const uint8_t* D = desc + (size_t)offsets[p] * 32;
        const size_t N = (size_t)(offsets[p + 1] - offsets[p]);
        best[p] = -1;
        if (N == 0) continue;
        std::vector<std::vector<float>> Distances(N, std::vector<float>(N, 0.f));
        for (size_t i = 0; i < N; i++) {
            Distances[i][i] = 0;
            for (size_t j = i + 1; j < N; j++) {
                const int distij = descriptor_distance(D + i * 32, D + j * 32);
                Distances[i][j] = distij;
                Distances[j][i] = distij;
            }
        }
        int BestMedian = INT_MAX, BestIdx = 0;
        for (size_t i = 0; i < N; i++) {
            std::vector<int> vDists(Distances[i].begin(), Distances[i].end());
            std::sort(vDists.begin(), vDists.end());
            const int median = vDists[0.5 * (N - 1)];
            if (median < BestMedian) { BestMedian = median; BestIdx = (int)i; }
        }
        best[p] = BestIdx;
    }
}

}  // extern "C"
