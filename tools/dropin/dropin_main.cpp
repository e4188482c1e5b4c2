// dropin_main — a compiled C++ consumer of psl-slam_amd/host/pslfe.hpp (and through it of the C ABI in
// include/pslfe.h): runs, one frame at a time, the call sequence the reference's tracking thread issues for
// an RGB-D frame and times it with the host clock, copies included.  Harness code, not part of libpslfe.
//
//   Frame::Frame (RGB-D)            src/Frame.cc:133-208   ExtractORB; ExtractLSD (LINEextractor, CPartiallyRecover-
//                                                          Connectivity, isLineGood / fans / planes); UndistortKeyPoints;
//                                                          ComputeStereoFromRGBD; AssignFeaturesToGrid
//   Tracking::TrackWithMotionModel  src/Tracking.cc:1164-1214  SearchByGeomNApearance(cur,last,0.95);
//                                                          SearchByProjection(cur,last,th=15); retry with 2*th and the line
//                                                          projection search when nmatches+lmatches < 20;
//                                                          AssociatePlanesByBoundary(cur, 0.05, 0.999)
// The pose is a stand-in (the solver is out of scope): a map point of the last frame is predicted at the pixel it
// was seen at (the synthetic stream drifts by <= 2 px per frame), the "map" of planes is the last frame's planes.
// Per-frame wall time is what the reference itself reports (Examples/RGB-D/rgbd_tum.cc:103-119).
//
// With a look-ahead K > 1 the Frame members come from pslfe::FramePrefetcher instead: the "reader" pushes up to K frames ahead
// (Examples/RGB-D/rgbd_tum.cc:88-93 reads them from disk), their extraction runs as ONE batched launch, and every frame is then tracked
// as before.  Results are those of K = 1 bit for bit.
//
// usage: dropin_main <frames.bin> <nfeatures> <nlines> <warmup> [results.bin | -] [lookahead]
//   frames.bin : int32 magic 0x50534C46, w, h, n; n gray frames (w*h u8); n depth frames (w*h f32, metres)
//   results.bin: per frame the outputs a parity test compares with the oracle (see dump()).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../psl-slam_amd/host/pslfe.hpp"

using Clock = std::chrono::steady_clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

typedef pslfe::FramePrefetcher::Frame FrameData;  // the members of ORB_SLAM2::Frame this path fills

template <class T> static void put(FILE* f, const std::vector<T>& v) {
    const int64_t n = (int64_t)v.size();
    fwrite(&n, sizeof(n), 1, f);
    if (n) fwrite(v.data(), sizeof(T), v.size(), f);
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s frames.bin nfeatures nlines warmup [results.bin]\n", argv[0]); return 2; }
    FILE* fi = fopen(argv[1], "rb");
    if (!fi) { perror(argv[1]); return 2; }
    int32_t hdr[4];
    if (fread(hdr, sizeof(int32_t), 4, fi) != 4 || hdr[0] != 0x50534C46) { fprintf(stderr, "bad frames file\n"); return 2; }
    const int w = hdr[1], h = hdr[2], n = hdr[3];
    const int nfeatures = atoi(argv[2]), nlines = atoi(argv[3]), warmup = atoi(argv[4]);
    std::vector<uint8_t> gray((size_t)w * h * n);
    std::vector<float> depth((size_t)w * h * n);
    if (fread(gray.data(), 1, gray.size(), fi) != gray.size() || fread(depth.data(), sizeof(float), depth.size(), fi) != depth.size()) {
        fprintf(stderr, "short frames file\n"); return 2;
    }
    fclose(fi);
    FILE* fo = (argc > 5 && strcmp(argv[5], "-") != 0) ? fopen(argv[5], "wb") : nullptr;
    const int lookahead = argc > 6 ? std::max(1, atoi(argv[6])) : 1;

    try {
        pslfe::Context ctx(0);
        // Examples/RGB-D/TUM3.yaml scaled with the image width (zero distortion, as every RGB-D YAML of the reference)
        const float s = (float)w / 640.0f;
        PslCamera cam = {535.4f * s, 539.2f * s, 320.1f * s, 247.6f * s, 0, 0, 0, 0, 0, 40.0f * s};
        pslfe::ORBextractor orb(ctx, nfeatures, 1.2f, 8, 20, 7);               // src/Tracking.cc:120
        pslfe::LINEextractor lsd(ctx, 1, 1.2f, (unsigned)nlines, 0.0);         // src/Tracking.cc:127
        const int cap = pslfe_orb_max_keypoints(orb.get(), w, h);
        pslfe::FrameGrid grid1(ctx, cap, 2);
        pslfe::FrameGlue glue(ctx, 2048, 4096);
        pslfe::FramePrefetcher* pf = lookahead > 1 ? new pslfe::FramePrefetcher(ctx, w, h, lookahead, nfeatures, 1.2f, 8, 20, 7, nlines, cam) : nullptr;
        pslfe::FrameGrid* gridp = &grid1;   // with a look-ahead: the grid of the lane the current frame was extracted on (cur.grid)
        pslfe::ORBmatcher matcher(0.9f, true);
        pslfe::LSDmatcher lmatcher(ctx);
        const std::vector<float> scale = orb.GetScaleFactors();
        float bounds[4];
        grid1.imageBounds(cam, w, h, bounds);  // ComputeImageBounds, first frame only (src/Frame.cc:158-174)
        int next_push = 0;

        std::map<std::string, std::vector<double>> T;  // per-call times of the timed frames
        std::vector<double> frame_ms, track_ms;
        FrameData last, cur;
        std::vector<PslProjQuery> q;
        std::vector<int32_t> match, assigned, lm12, lassigned, plane_assoc;
        long sum_kp = 0, sum_kl = 0, sum_fans = 0, sum_planes = 0, sum_nm = 0, sum_lm = 0, retries = 0;
        // PSLFE_DROPIN_STAGES=1: the library's per-kernel-stage HIP-event timers are on as well (they add event records
        // between kernels, so the wall times of such a run are a little longer)
        const bool stage_prof = getenv("PSLFE_DROPIN_STAGES") != nullptr;
        for (int t = 0; t < n; ++t) {
            if (stage_prof && t == warmup) { pslfe_ctx_profile(ctx.get(), 1); pslfe_ctx_profile_reset(ctx.get()); }
            const uint8_t* img = gray.data() + (size_t)t * w * h;
            const float* dep = depth.data() + (size_t)t * w * h;
            const bool timed = t >= warmup;
            auto lap = [&](const char* name, Clock::time_point& t0) {
                const double ms = ms_since(t0);
                if (timed) T[name].push_back(ms);
                t0 = Clock::now();
            };
            // ---------------- Frame::Frame ----------------
            const auto tf = Clock::now();
            auto t0 = tf;
            int slot = t & 1;
            if (pf) {   // look-ahead: the reader is up to K frames ahead of the tracker; the Frame members come out of the prefetcher
                // the reader runs ahead for as long as a lane is free (two batches of K frames at most)
                while (next_push < n && pf->push(gray.data() + (size_t)next_push * w * h, w, depth.data() + (size_t)next_push * w * h, w)) ++next_push;
                lap("FramePrefetcher push (H2D)", t0);
                if (!pf->pop(cur)) { fprintf(stderr, "dropin_main: the prefetcher ran dry at frame %d\n", t); return 1; }
                slot = cur.slot;
                gridp = cur.grid;
                lap("FramePrefetcher pop", t0);
            } else {
            orb(img, w, h, w, cur.mvKeys, cur.mDescriptors);                                   // ExtractORB
            lap("ORBextractor()", t0);
            lsd(img, w, h, w, cur.mvKeylinesUn, cur.mLdesc, cur.mvKeyLineFunctions);           // ExtractLSD: extractor
            lap("LINEextractor()", t0);
            std::vector<float> mLines(cur.mvKeylinesUn.size() * 4);                            // keyLinesToMat (src/Frame.cc:355-373)
            for (size_t i = 0; i < cur.mvKeylinesUn.size(); ++i) {
                const PslKeyLine& k = cur.mvKeylinesUn[i];
                mLines[4 * i] = k.startPointX; mLines[4 * i + 1] = k.startPointY; mLines[4 * i + 2] = k.endPointX; mLines[4 * i + 3] = k.endPointY;
            }
            lsd.PartiallyRecoverConnectivity(mLines, 20.0f, cur.fans, w, h, (float)(M_PI / 4));  // src/Frame.cc:505
            lap("CPartiallyRecoverConnectivity", t0);
            cur.glue = glue.run(cur.mvKeylinesUn, cur.fans, dep, w, h, w, cam, 1u + (uint32_t)t);  // isLineGood, fans, planes
            lap("isLineGood+fans+planes", t0);
            if (!cur.mvKeys.empty()) {                                                         // UndistortKeyPoints .. AssignFeaturesToGrid
                grid1.setRGBD(t & 1, cur.mvKeys, cur.mDescriptors, dep, w, h, w, cam);
                grid1.fetch(t & 1, cur.mvKeysUn, cur.mvDepth, cur.mvuRight, cap);
            }
            lap("Undistort+StereoFromRGBD+Grid", t0);
            }
            const double fms = ms_since(tf);
            // ---------------- Tracking::TrackWithMotionModel ----------------
            const auto tt = Clock::now();
            t0 = tt;
            int nmatches = 0, lmatches = 0, ljl = 0;
            if (t > 0 && !cur.mvKeys.empty()) {
                std::vector<uint8_t> hasMapLine(last.mvKeylinesUn.size(), 1);
                lmatches = lmatcher.SearchByGeomNApearance(last.mvKeylinesUn, last.mLdesc, cur.mvKeylinesUn, cur.mLdesc, hasMapLine, 0.95f,
                                                           bounds[0], bounds[2], bounds[1], bounds[3], lm12, lassigned);
                lap("SearchByGeomNApearance", t0);
                auto project = [&](int th) {
                    q.resize(last.mvKeysUn.size());
                    for (size_t i = 0; i < q.size(); ++i) {
                        const PslKeyPoint& k = last.mvKeysUn[i];
                        q[i].u = k.x; q[i].v = k.y; q[i].radius = (float)th * scale[k.octave];
                        q[i].ur = last.mvuRight[i];
                        q[i].min_level = k.octave - 1; q[i].max_level = k.octave + 1;          // neither forward nor backward (:1387-1390)
                        q[i].angle = k.angle; q[i].blocks = 1;
                    }
                    assigned.assign(cur.mvKeys.size(), -1);
                    return matcher.SearchByProjection(*gridp, slot, q, last.mDescriptors, nullptr, match, &assigned);
                };
                nmatches = project(15);
                lap("SearchByProjection(cur,last)", t0);
                if (nmatches + lmatches < 20) { nmatches = project(30); ++retries; lap("SearchByProjection retry", t0); }
                // AssociatePlanesByBoundary against the last frame's planes (stand-in for the map's)
                const size_t np = cur.glue.planes.size() / 4, nmap = last.glue.planes.size() / 4;
                if (np && nmap) {
                    std::vector<double> pts(np * 15);
                    for (size_t p = 0; p < np; ++p) {
                        const int li = cur.glue.lineNo[2 * p], lj = cur.glue.lineNo[2 * p + 1];
                        memcpy(&pts[p * 15], &cur.glue.lines3d[(size_t)li * 6], 6 * sizeof(double));
                        memcpy(&pts[p * 15 + 6], &cur.glue.lines3d[(size_t)lj * 6], 6 * sizeof(double));
                        memcpy(&pts[p * 15 + 12], &cur.glue.cross3d[p * 3], 3 * sizeof(double));
                    }
                    plane_assoc.assign(np, -1);
                    pslfe::check(pslfe_associate_planes(ctx.get(), cur.glue.planes.data(), pts.data(), (int)np, last.glue.planes.data(), nullptr,
                                                        (int)nmap, 0.05f, 0.999f, 1, plane_assoc.data(), &ljl), "pslfe_associate_planes");
                } else plane_assoc.clear();
                lap("AssociatePlanesByBoundary", t0);
            }
            const double tms = ms_since(tt);
            if (timed) {
                frame_ms.push_back(fms); track_ms.push_back(tms);
                sum_kp += (long)cur.mvKeys.size(); sum_kl += (long)cur.mvKeylinesUn.size(); sum_fans += (long)cur.fans.size() / 4;
                sum_planes += (long)cur.glue.planes.size() / 4; sum_nm += nmatches; sum_lm += lmatches;
            }
            if (fo) {  // what the parity test compares with the oracle
                put(fo, cur.mvKeys); put(fo, cur.mDescriptors); put(fo, cur.mvKeysUn); put(fo, cur.mvDepth); put(fo, cur.mvuRight);
                put(fo, cur.mvKeylinesUn); put(fo, cur.mLdesc); put(fo, cur.mvKeyLineFunctions); put(fo, cur.fans);
                put(fo, cur.glue.lines3d); put(fo, cur.glue.planes); put(fo, cur.glue.lineNo);
                put(fo, t > 0 ? match : std::vector<int32_t>()); put(fo, t > 0 ? lm12 : std::vector<int32_t>());
                put(fo, t > 0 ? lassigned : std::vector<int32_t>()); put(fo, t > 0 ? plane_assoc : std::vector<int32_t>());
            }
            std::swap(last, cur);
        }
        if (fo) fclose(fo);
        auto stats = [](std::vector<double> v, double* med, double* mean, double* p95) {
            std::sort(v.begin(), v.end());
            double s = 0; for (double x : v) s += x;
            *mean = v.empty() ? 0 : s / v.size();
            *med = v.empty() ? 0 : v[v.size() / 2];
            *p95 = v.empty() ? 0 : v[std::min(v.size() - 1, (size_t)(0.95 * v.size()))];
        };
        const size_t m = frame_ms.size();
        std::vector<double> total(m);
        for (size_t i = 0; i < m; ++i) total[i] = frame_ms[i] + track_ms[i];
        double med, mean, p95, fmed, fmean, fp95, tmed, tmean, tp95;
        stats(total, &med, &mean, &p95); stats(frame_ms, &fmed, &fmean, &fp95); stats(track_ms, &tmed, &tmean, &tp95);
        printf("{\"frames_timed\": %zu, \"lookahead\": %d, \"w\": %d, \"h\": %d, \"nfeatures\": %d, \"nlines\": %d, "
               "\"ms_per_frame\": {\"median\": %.4f, \"mean\": %.4f, \"p95\": %.4f}, "
               "\"frame_ctor_ms\": {\"median\": %.4f, \"mean\": %.4f, \"p95\": %.4f}, "
               "\"track_ms\": {\"median\": %.4f, \"mean\": %.4f, \"p95\": %.4f}, "
               "\"mean_keypoints\": %.1f, \"mean_keylines\": %.1f, \"mean_fans\": %.1f, \"mean_planes\": %.1f, \"mean_matches\": %.1f, "
               "\"mean_line_matches\": %.1f, \"retries\": %ld, \"calls_ms_mean\": {",
               m, lookahead, w, h, nfeatures, nlines, med, mean, p95, fmed, fmean, fp95, tmed, tmean, tp95, (double)sum_kp / m, (double)sum_kl / m,
               (double)sum_fans / m, (double)sum_planes / m, (double)sum_nm / m, (double)sum_lm / m, retries);
        bool first = true;
        for (auto& kv : T) {
            double a, b, c; stats(kv.second, &a, &b, &c);
            printf("%s\"%s\": %.4f", first ? "" : ", ", kv.first.c_str(), b);
            first = false;
        }
        printf("}");
        if (stage_prof) {
            static const char* names[] = {"orb.pyramid", "orb.fast", "orb.octree", "orb.blur", "orb.describe", "match.grid", "match.window",
                                          "line.lsd_scale", "line.lsd_grad", "line.lsd_grow", "line.nfa_count", "line.nfa_eval", "line.merge", "line.lbd_pre", "line.lbd", "line.pair",
                                          "line.match", "line.good", "line.planes"};
            printf(", \"gpu_stage_ms_per_frame\": {");
            first = true;
            for (const char* nm : names) {
                double ms = 0; int cnt = 0;
                if (pslfe_ctx_stage_time(ctx.get(), nm, &ms, &cnt) == PSLFE_OK && cnt > 0) {
                    printf("%s\"%s\": %.4f", first ? "" : ", ", nm, ms / (double)m);
                    first = false;
                }
            }
            printf("}");
        }
        printf("}\n");
        delete pf;
    } catch (const std::exception& e) {
        fprintf(stderr, "dropin_main: %s\n", e.what());
        return 1;
    }
    return 0;
}
