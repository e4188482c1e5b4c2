// ORACLE — test infrastructure only (see orb_oracle.cpp header). CPU restatement of the small
// per-frame association routines that sit on top of the descriptor matchers:
//   LSDmatcher::SearchByGeomNApearance    add_src/LSDmatcher.cpp:36-110 (+ computeAngle2D :20-34)
//   LSDmatcher::FrameBFMatch              add_src/LSDmatcher.cpp:492-516 (+ lineDescriptorMAD :660-685)
//   Map::AssociatePlanesByBoundary (live) src/Map.cc:204-272
//   InsectLineMatch::SearchMapInsectline  add_src/InsectlineMatch.cpp:9-59 (dead code upstream, H14)
// PARITY UNPINNED (no fixtures upstream).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "psl_oracle.h"

extern "C" {

int pso_search_by_geom_appearance(const PsoKeyLine* kl_last, const uint8_t* d_last, int n1, const PsoKeyLine* kl_cur, const uint8_t* d_cur,
                                  int n2, const uint8_t* has_mapline, float desc_th, float minX, float maxX, float minY, float maxY,
                                  int* matches12, int* assigned) {
    for (int i = 0; i < n2; ++i) assigned[i] = -1;
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    if (n2 == 0) return 0;  // mLdesc.empty() -> return 0 (:40-43)
    pso_line_match_nnr(d_last, n1, d_cur, n2, desc_th, matches12);
    int lmatches = 0;
    const double deltaWidth = (maxX - minX) * 0.1, deltaHeight = (maxY - minY) * 0.1;
    const double th_rad = 20.0 / 180.0 * M_PI, cos_th_angle = std::cos(th_rad);
    for (int i1 = 0; i1 < n1; ++i1) {
        if (!has_mapline[i1]) continue;
        const int i2 = matches12[i1];
        if (i2 < 0) continue;
        if (kl_cur[i2].startPointX == 0) continue;
        const double vc0 = kl_cur[i2].ePointInOctaveX - kl_cur[i2].sPointInOctaveX, vc1 = kl_cur[i2].ePointInOctaveY - kl_cur[i2].sPointInOctaveY;
        const double vl0 = kl_last[i1].ePointInOctaveX - kl_last[i1].sPointInOctaveX, vl1 = kl_last[i1].ePointInOctaveY - kl_last[i1].sPointInOctaveY;
        const double dot = vc0 * vl0 + vc1 * vl1;
        const double mA = std::sqrt(vc0 * vc0 + vc1 * vc1), mB = std::sqrt(vl0 * vl0 + vl1 * vl1);
        const double angle = std::abs(dot / (mA * mB));
        if (angle < cos_th_angle) { matches12[i1] = -1; continue; }
        const float sXc = kl_cur[i2].sPointInOctaveX, sXl = kl_last[i1].sPointInOctaveX, sYc = kl_cur[i2].sPointInOctaveY, sYl = kl_last[i1].sPointInOctaveY;
        const float eXc = kl_cur[i2].ePointInOctaveX, eXl = kl_last[i1].ePointInOctaveX, eYc = kl_cur[i2].ePointInOctaveY, eYl = kl_last[i1].ePointInOctaveY;
        if ((std::fabs(sXc - sXl) > deltaWidth || std::fabs(sYc - sYl) > deltaHeight) && (std::fabs(eXc - eXl) > deltaWidth || std::fabs(eYc - eYl) > deltaHeight)) {
            matches12[i1] = -1;
            continue;
        }
        assigned[i2] = i1;
        ++lmatches;
    }
    return lmatches;
}

void pso_frame_bf_match(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnratio, float TH, int* lineMatches) {
    for (int i = 0; i < n1; ++i) lineMatches[i] = -1;
    if (n1 == 0 || n2 < 2) return;  // knnMatch(k=2) with < 2 train rows: reference indexes out of bounds (H12)
    std::vector<int> idx(2 * (size_t)n1), dist(2 * (size_t)n1);
    pso_hamming_knn2(d1, n1, d2, n2, idx.data(), dist.data());
    // lineDescriptorMAD: only nn12_mad reaches the decision
    std::vector<float> d12(n1);
    for (int i = 0; i < n1; ++i) d12[i] = (float)dist[2 * i + 1] - (float)dist[2 * i];
    std::vector<float> s = d12;
    std::sort(s.begin(), s.end());
    const double nn12_median = s[n1 / 2];
    std::vector<float> dev(n1);
    for (int i = 0; i < n1; ++i) dev[i] = fabsf((float)((double)(float)dist[2 * i + 1] - (double)(float)dist[2 * i] - nn12_median));
    std::sort(dev.begin(), dev.end());
    double nn12_th = 1.4826 * dev[n1 / 2];
    nn12_th = nn12_th * 0.5;
    for (int i = 0; i < n1; ++i) {
        const float a = (float)dist[2 * i], b = (float)dist[2 * i + 1];
        const double dist_12 = b - a;
        if (dist_12 > nn12_th && a < TH && a < nnratio * b) lineMatches[i] = idx[2 * i];
    }
}

// planes: N x 4 floats (world plane of frame LIL i); pts: N x 5 x 3 doubles (is, ie, js, je, intersection);
// map: M x 4 floats; bad: M flags (dead variant only).  live != 0: Map::AssociatePlanesByBoundary semantics.
int pso_associate_planes(const float* planes, const double* pts, int N, const float* map, const uint8_t* bad, int M, float dTh, float aTh,
                         int live, int* assoc) {
    int nmatches = 0;
    for (int i = 0; i < N; ++i) {
        assoc[i] = -1;
        const float* pM = planes + 4 * i;
        const double* P = pts + 15 * i;
        float ldTh = dTh;
        bool found = false;
        for (int j = 0; j < M; ++j) {
            if (!live && bad && bad[j]) continue;
            float pW[4] = {map[4 * j], map[4 * j + 1], map[4 * j + 2], map[4 * j + 3]};
            if (live && pW[3] < 0) { pW[0] = -pW[0]; pW[1] = -pW[1]; pW[2] = -pW[2]; pW[3] = -pW[3]; }
            const float angle = pM[0] * pW[0] + pM[1] * pW[1] + pM[2] * pW[2];
            if (angle > aTh || angle < -aTh) {
                float d5[5];
                for (int k = 0; k < 5; ++k) d5[k] = (float)(pW[0] * P[3 * k] + pW[1] * P[3 * k + 1] + pW[2] * P[3 * k + 2] + pW[3]);
                const float dis = (d5[0] + d5[1] + d5[2] + d5[3] + d5[4]) / 5;
                if (live) {
                    if (std::abs(dis) < dTh) { dTh = dis; assoc[i] = j; nmatches++; }   // dTh is the (mutated) argument, shared by all i (:249-251)
                } else {
                    if (std::abs(dis) < ldTh) { ldTh = dis; assoc[i] = j; found = true; }
                }
            }
        }
        if (!live && found) nmatches++;
    }
    return nmatches;
}

}  // extern "C"
