// C++ host-side mirror of the reference's extractor / matcher classes over the C ABI
// (include/pslfe.h).  Header-only, no OpenCV: keypoints are PslKeyPoint (layout == cv::KeyPoint),
// descriptors std::vector<uint8_t> (N x 32 row-major == cv::Mat N x 32 CV_8U).  The OpenCV-typed
// adapter a PSL-SLAM maintainer adds on top is shown in INTEGRATION.md.
//
// Names, argument meaning and error behaviour follow the reference:
//   ORBextractor  include/ORBextractor.h:45-114   (operator(), getters)
//   ORBmatcher    include/ORBmatcher.h:36-104     (SearchByProjection, DescriptorDistance, TH_*)
//   LSDmatcher    add_inc/LSDmatcher.h:18-75      (matchNNR / match)
#ifndef PSLFE_HPP
#define PSLFE_HPP

#include <cstdint>
#include <cstring>
#include <deque>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pslfe.h"

namespace pslfe {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what + ": " + pslfe_last_error()), code(c) {}
};
inline void check(int rc, const char* what) { if (rc != PSLFE_OK) throw Error(rc, what); }

class Context {
public:
    explicit Context(int device = 0) : device_(device) { check(pslfe_ctx_create(device, &h_), "pslfe_ctx_create"); }
    int device() const { return device_; }
    ~Context() { pslfe_ctx_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    pslfe_ctx* get() const { return h_; }
    void setStream(void* hipStream) { check(pslfe_ctx_set_stream(h_, hipStream), "pslfe_ctx_set_stream"); }
    void synchronize() { check(pslfe_ctx_synchronize(h_), "pslfe_ctx_synchronize"); }
private:
    pslfe_ctx* h_ = nullptr;
    int device_ = 0;
};

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(Context& ctx, int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int maxBatch = 1)
        : nlevels_(nlevels) {
        check(pslfe_orb_create(ctx.get(), nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, maxBatch, &h_), "pslfe_orb_create");
    }
    ~ORBextractor() { pslfe_orb_destroy(h_); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // operator()(image, mask, keypoints, descriptors): mask is ignored, as in the reference;
    // an empty image leaves the outputs untouched (src/ORBextractor.cc:1046).
    void operator()(const uint8_t* image, int cols, int rows, int step, std::vector<PslKeyPoint>& keypoints,
                    std::vector<uint8_t>& descriptors) {
        if (!image || cols <= 0 || rows <= 0) return;
        const int cap = pslfe_orb_max_keypoints(h_, cols, rows);
        if (cap < 0) throw Error(cap, "pslfe_orb_max_keypoints");
        keypoints.resize(cap);
        descriptors.resize((size_t)cap * 32);
        int n = 0;
        check(pslfe_orb_extract(h_, image, cols, rows, step, keypoints.data(), descriptors.data(), cap, &n), "pslfe_orb_extract");
        keypoints.resize(n);
        descriptors.resize((size_t)n * 32);  // n == 0 <=> descriptors.release() (:1064-1065)
    }

    int GetLevels() const { return pslfe_orb_levels(h_); }
    float GetScaleFactor() const { return pslfe_orb_scale_factor(h_); }
    std::vector<float> GetScaleFactors() const { return factors(0); }
    std::vector<float> GetInverseScaleFactors() const { return factors(1); }
    std::vector<float> GetScaleSigmaSquares() const { return factors(2); }
    std::vector<float> GetInverseScaleSigmaSquares() const { return factors(3); }
    pslfe_orb* get() const { return h_; }

private:
    std::vector<float> factors(int which) const {
        std::vector<float> v[4];
        for (auto& x : v) x.resize(nlevels_);
        check(pslfe_orb_scale_factors(h_, v[0].data(), v[1].data(), v[2].data(), v[3].data()), "pslfe_orb_scale_factors");
        return v[which];
    }
    pslfe_orb* h_ = nullptr;
    int nlevels_;
};

// The part of ORB_SLAM2::Frame the matchers read (mvKeysUn, mDescriptors, mvuRight, mGrid).
class FrameGrid {
public:
    FrameGrid(Context& ctx, int maxKeypoints, int maxFrames = 1) {
        check(pslfe_frame_create(ctx.get(), maxKeypoints, maxFrames, &h_), "pslfe_frame_create");
    }
    ~FrameGrid() { pslfe_frame_destroy(h_); }
    FrameGrid(const FrameGrid&) = delete;
    FrameGrid& operator=(const FrameGrid&) = delete;
    void set(int slot, const std::vector<PslKeyPoint>& keysUn, const std::vector<uint8_t>& descriptors, const float* uRight,
             float mnMinX, float mnMinY, float mnMaxX, float mnMaxY) {
        check(pslfe_frame_set(h_, slot, keysUn.data(), descriptors.data(), uRight, (int)keysUn.size(), mnMinX, mnMinY, mnMaxX, mnMaxY),
              "pslfe_frame_set");
    }
    // The RGB-D part of the Frame constructor (src/Frame.cc:105-171): UndistortKeyPoints, ComputeStereoFromRGBD,
    // ComputeImageBounds and AssignFeaturesToGrid from the raw keypoints and the CV_32F depth image.
    void setRGBD(int slot, const std::vector<PslKeyPoint>& keys, const std::vector<uint8_t>& descriptors, const float* depth, int cols,
                 int rows, int strideFloats, const PslCamera& cam) {
        check(pslfe_frame_set_rgbd(h_, slot, keys.data(), descriptors.data(), (int)keys.size(), depth, cols, rows, strideFloats, &cam),
              "pslfe_frame_set_rgbd");
    }
    // mvKeysUn, mvDepth, mvuRight of a slot
    void fetch(int slot, std::vector<PslKeyPoint>& keysUn, std::vector<float>& depth, std::vector<float>& uRight, int capacity) {
        keysUn.resize(capacity); depth.resize(capacity); uRight.resize(capacity);
        int n = 0;
        check(pslfe_frame_fetch(h_, slot, keysUn.data(), depth.data(), uRight.data(), capacity, &n), "pslfe_frame_fetch");
        keysUn.resize(n); depth.resize(n); uRight.resize(n);
    }
    // mnMinX, mnMinY, mnMaxX, mnMaxY (src/Frame.cc:1135-1168)
    void imageBounds(const PslCamera& cam, int cols, int rows, float bounds[4]) {
        check(pslfe_image_bounds(h_, &cam, cols, rows, bounds), "pslfe_image_bounds");
    }
    pslfe_frame* get() const { return h_; }
private:
    pslfe_frame* h_ = nullptr;
};

class ORBmatcher {
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;  // src/ORBmatcher.cc:37-39
    ORBmatcher(float nnratio = 0.6f, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    // SearchByProjection(CurrentFrame, LastFrame, th, bMono), src/ORBmatcher.cc:1328: the caller has
    // projected LastFrame's map points (queries / qdesc); returns nmatches.
    int SearchByProjection(FrameGrid& cur, int slot, const std::vector<PslProjQuery>& queries, const std::vector<uint8_t>& qdesc,
                           const uint8_t* taken, std::vector<int32_t>& match, std::vector<int32_t>* assigned = nullptr) {
        match.assign(queries.size(), -1);
        int nm = 0;
        check(pslfe_orb_search_by_projection_last(cur.get(), slot, queries.data(), qdesc.data(), (int)queries.size(), taken,
                                                  mbCheckOrientation ? 1 : 0, match.data(), assigned ? assigned->data() : nullptr, &nm),
              "pslfe_orb_search_by_projection_last");
        return nm;
    }
    // SearchByProjection(F, vpMapPoints, th), src/ORBmatcher.cc:45.
    int SearchByProjectionMap(FrameGrid& cur, int slot, const std::vector<PslProjQuery>& queries, const std::vector<uint8_t>& qdesc,
                              const uint8_t* taken, std::vector<int32_t>& match, std::vector<int32_t>* assigned = nullptr) {
        match.assign(queries.size(), -1);
        int nm = 0;
        check(pslfe_orb_search_by_projection_map(cur.get(), slot, queries.data(), qdesc.data(), (int)queries.size(), taken, mfNNratio,
                                                 match.data(), assigned ? assigned->data() : nullptr, &nm),
              "pslfe_orb_search_by_projection_map");
        return nm;
    }
    // SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist), src/ORBmatcher.cc:1472 (relocalisation): the caller has
    // projected the keyframe's map points; taken[c] != 0 <=> CurrentFrame.mvpMapPoints[c] != NULL.
    int SearchByProjectionKF(FrameGrid& cur, int slot, const std::vector<PslProjQuery>& queries, const std::vector<uint8_t>& qdesc,
                             const uint8_t* taken, int ORBdist, std::vector<int32_t>& match, std::vector<int32_t>* assigned = nullptr) {
        match.assign(queries.size(), -1);
        int nm = 0;
        check(pslfe_orb_search_by_projection_kf(cur.get(), slot, queries.data(), qdesc.data(), (int)queries.size(), taken, ORBdist,
                                                mbCheckOrientation ? 1 : 0, match.data(), assigned ? assigned->data() : nullptr, &nm),
              "pslfe_orb_search_by_projection_kf");
        return nm;
    }
    // SearchByBoW(pKF, F, vpMapPointMatches), src/ORBmatcher.cc:159, on host-side DBoW2 FeatureVectors: fidx = F.mFeatVec flattened
    // in node order; one query (node run of fidx, angle) + descriptor per keyframe feature, in the reference's iteration order.
    int SearchByBoW(FrameGrid& frame, int slot, const std::vector<int32_t>& fidx, const std::vector<PslBowQuery>& queries,
                    const std::vector<uint8_t>& qdesc, std::vector<int32_t>& match, std::vector<int32_t>* assigned = nullptr) {
        match.assign(queries.size(), -1);
        int nm = 0;
        check(pslfe_orb_search_by_bow(frame.get(), slot, fidx.data(), (int)fidx.size(), queries.data(), qdesc.data(), (int)queries.size(),
                                      mfNNratio, mbCheckOrientation ? 1 : 0, match.data(), assigned ? assigned->data() : nullptr, &nm),
              "pslfe_orb_search_by_bow");
        return nm;
    }
    // DescriptorDistance, src/ORBmatcher.cc:1647-1663 (host helper, same SWAR popcount)
    static int DescriptorDistance(const uint8_t* a, const uint8_t* b) {
        const uint32_t* pa = reinterpret_cast<const uint32_t*>(a);
        const uint32_t* pb = reinterpret_cast<const uint32_t*>(b);
        int dist = 0;
        for (int i = 0; i < 8; ++i) dist += __builtin_popcount(pa[i] ^ pb[i]);
        return dist;
    }
    float mfNNratio;
    bool mbCheckOrientation;
};

// == ORB_SLAM2::LINEextractor (add_inc/LineExtractor.h:160-255)
class LINEextractor {
public:
    LINEextractor(Context& ctx, int numOctaves, float scale, unsigned int nLSDFeature, double min_line_length, int maxBatch = 1)
        : numOctaves_(numOctaves) {
        check(pslfe_line_create(ctx.get(), numOctaves, scale, (int)nLSDFeature, min_line_length, maxBatch, &h_), "pslfe_line_create");
    }
    ~LINEextractor() { pslfe_line_destroy(h_); }
    LINEextractor(const LINEextractor&) = delete;
    LINEextractor& operator=(const LINEextractor&) = delete;

    // operator()(image, mask, keylines, descriptors, lineVec2d): empty image leaves the outputs untouched
    // (add_src/LineExtractor.cpp:327); lineVec2d holds 3 doubles per line (Eigen::Vector3d layout).
    void operator()(const uint8_t* image, int cols, int rows, int step, std::vector<PslKeyLine>& keylines, std::vector<uint8_t>& descriptors,
                    std::vector<double>& lineVec2d) {
        if (!image || cols <= 0 || rows <= 0) return;
        const int cap = 2048;
        keylines.resize(cap); descriptors.resize((size_t)cap * 32); lineVec2d.resize((size_t)cap * 3);
        int n = 0;
        check(pslfe_line_extract(h_, image, cols, rows, step, keylines.data(), descriptors.data(), lineVec2d.data(), cap, &n), "pslfe_line_extract");
        keylines.resize(n); descriptors.resize((size_t)n * 32); lineVec2d.resize((size_t)n * 3);
    }
    // CPartiallyRecoverConnectivity(mLines, radius, fans, img, fanThr): mLines n x 4 floats, fans k x 4
    void PartiallyRecoverConnectivity(const std::vector<float>& mLines, float radius, std::vector<float>& fans, int imgCols, int imgRows, float fanThr) {
        const int n = (int)mLines.size() / 4, cap = 4096;
        fans.resize((size_t)cap * 4);
        int k = 0;
        check(pslfe_lil_pair(h_, mLines.data(), n, radius, fanThr, imgCols, imgRows, fans.data(), cap, &k), "pslfe_lil_pair");
        fans.resize((size_t)k * 4);
    }
    // PSLFE_LSD_REFINE_ADV (default, what the stock contrib LSDDetector constructs) or PSLFE_LSD_REFINE_STD
    void SetRefine(int mode) { check(pslfe_line_set_refine(h_, mode), "pslfe_line_set_refine"); }
    int GetLevels() const { return pslfe_line_levels(h_); }
    float GetScaleFactor() const { return pslfe_line_scale_factor(h_); }
    std::vector<float> GetScaleFactors() const { return factors(0); }
    std::vector<float> GetInverseScaleFactors() const { return factors(1); }
    std::vector<float> GetScaleSigmaSquares() const { return factors(2); }
    std::vector<float> GetInverseScaleSigmaSquares() const { return factors(3); }
    pslfe_line* get() const { return h_; }

private:
    std::vector<float> factors(int which) const {
        std::vector<float> v[4];
        for (auto& x : v) x.resize(numOctaves_);
        check(pslfe_line_scale_factors(h_, v[0].data(), v[1].data(), v[2].data(), v[3].data()), "pslfe_line_scale_factors");
        return v[which];
    }
    pslfe_line* h_ = nullptr;
    int numOctaves_;
};

class LSDmatcher {
public:
    static const int TH_HIGH = 80, TH_LOW = 50;  // add_src/LSDmatcher.cpp:12-14
    LSDmatcher(Context& ctx, float nnratio = 0.95f, bool checkOri = true) : ctx_(ctx), mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
    // matchNNR(desc1, desc2, nnr, matches_12), add_src/LSDmatcher.cpp:354-376
    int matchNNR(const std::vector<uint8_t>& desc1, const std::vector<uint8_t>& desc2, float nnr, std::vector<int>& matches_12) {
        const int n1 = (int)desc1.size() / 32, n2 = (int)desc2.size() / 32;
        matches_12.assign(n1, -1);
        int nm = 0;
        check(pslfe_line_match_nnr(ctx_.get(), desc1.data(), n1, desc2.data(), n2, nnr, matches_12.data(), &nm), "pslfe_line_match_nnr");
        return nm;
    }
    int match(const std::vector<uint8_t>& d1, const std::vector<uint8_t>& d2, float nnr, std::vector<int>& m12) { return matchNNR(d1, d2, nnr, m12); }
    // SearchByGeomNApearance(CurrentFrame, LastFrame, desc_th), add_src/LSDmatcher.cpp:36-110: assigned[i2] = last-frame line
    // whose map line CurrentFrame.mvpMapLines[i2] receives
    int SearchByGeomNApearance(const std::vector<PslKeyLine>& klLast, const std::vector<uint8_t>& descLast, const std::vector<PslKeyLine>& klCur,
                               const std::vector<uint8_t>& descCur, const std::vector<uint8_t>& lastHasMapLine, float desc_th, float mnMinX,
                               float mnMaxX, float mnMinY, float mnMaxY, std::vector<int32_t>& matches_12, std::vector<int32_t>& assigned) {
        matches_12.assign(klLast.size(), -1);
        assigned.assign(klCur.size(), -1);
        int n = 0;
        check(pslfe_line_search_by_geom_appearance(ctx_.get(), klLast.data(), descLast.data(), (int)klLast.size(), klCur.data(), descCur.data(),
                                                   (int)klCur.size(), lastHasMapLine.data(), desc_th, mnMinX, mnMaxX, mnMinY, mnMaxY,
                                                   matches_12.data(), assigned.data(), &n), "pslfe_line_search_by_geom_appearance");
        return n;
    }
    // FrameBFMatch(ldesc1, ldesc2, LineMatches, TH), add_src/LSDmatcher.cpp:492-516
    void FrameBFMatch(const std::vector<uint8_t>& ldesc1, const std::vector<uint8_t>& ldesc2, std::vector<int>& LineMatches, float TH) {
        LineMatches.assign(ldesc1.size() / 32, -1);
        check(pslfe_line_frame_bf_match(ctx_.get(), ldesc1.data(), (int)ldesc1.size() / 32, ldesc2.data(), (int)ldesc2.size() / 32, mfNNratio, TH,
                                        LineMatches.data()), "pslfe_line_frame_bf_match");
    }
    // SearchByProjection(CurrentFrame, LastFrame, th) :112-215 (mode 0) / SearchByProjection(F, vpMapLines, eval_orient, th) :260-352 (mode 1)
    int SearchByProjection(const std::vector<PslKeyLine>& kls, const std::vector<uint8_t>& ldesc, const std::vector<double>& keyLineFunctions,
                           const double* lines3dDir, float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, const std::vector<PslLineQuery>& queries,
                           const std::vector<uint8_t>& qdesc, const uint8_t* taken, int mode, std::vector<int32_t>& match,
                           std::vector<int32_t>* assigned = nullptr) {
        match.assign(queries.size(), -1);
        if (assigned) assigned->assign(kls.size(), -1);
        int n = 0;
        check(pslfe_line_search_by_projection(ctx_.get(), kls.data(), ldesc.data(), keyLineFunctions.data(), lines3dDir, (int)kls.size(), mnMinX,
                                              mnMinY, mnMaxX, mnMaxY, queries.data(), qdesc.data(), (int)queries.size(), taken, mode, mfNNratio,
                                              match.data(), assigned ? assigned->data() : nullptr, &n, nullptr, nullptr, 0, nullptr),
              "pslfe_line_search_by_projection");
        return n;
    }
private:
    Context& ctx_;
public:
    float mfNNratio;
    bool mbCheckOrientation;
};

// The part of Frame::ExtractLSD after the extractor (src/Frame.cc:490-660): isLineGood, convertFansToKeyLines, planes.
class FrameGlue {
public:
    struct Result {
        std::vector<double> lines3d;   // mvLines3D, n x 6
        std::vector<float> lineEq;     // mvLineEq, n x 3
        std::vector<int32_t> pair;     // intersection_lines_plane: line indices, k x 2
        std::vector<float> xy;         // ... 2-D crossing, k x 2
        std::vector<double> cross;     // ... 3-D crossing, k x 3
        std::vector<double> le_l;      // mvle_l, k x 6
        std::vector<float> planes;     // mvPlanes, p x 4
        std::vector<double> normals;   // mvPlaneNormal, p x 3
        std::vector<int32_t> lineNo;   // mvPlaneLineNo, p x 2
        std::vector<double> cross3d;   // CrossPoint_3D, p x 3
        std::vector<double> cross2d;   // CrossPoint_2D, p x 2
    };
    FrameGlue(Context& ctx, int maxLines = 1024, int maxFans = 4096, int maxBatch = 1) : maxFans_(maxFans) {
        check(pslfe_glue_create(ctx.get(), maxLines, maxFans, maxBatch, &h_), "pslfe_glue_create");
    }
    ~FrameGlue() { pslfe_glue_destroy(h_); }
    FrameGlue(const FrameGlue&) = delete;
    FrameGlue& operator=(const FrameGlue&) = delete;
    // fans: the n x 4 matrix of CPartiallyRecoverConnectivity; seed: srand(seed) of the frame (convention H7)
    Result run(const std::vector<PslKeyLine>& keylines, const std::vector<float>& fans, const float* depth, int cols, int rows, int strideFloats,
               const PslCamera& cam, uint32_t seed) {
        const int n = (int)keylines.size(), nf = (int)fans.size() / 4;
        check(pslfe_glue_run(h_, keylines.data(), n, fans.data(), nf, depth, cols, rows, strideFloats, &cam, seed), "pslfe_glue_run");
        return fetch(0, n);
    }
    // results of frame `frame` of the last run / pslfe_glue_run_batch_device; n = that frame's keyline count
    Result fetch(int frame, int n) {
        const int cap = maxFans_;
        Result r;
        r.lines3d.resize((size_t)n * 6); r.lineEq.resize((size_t)n * 3);
        r.pair.resize((size_t)cap * 2); r.xy.resize((size_t)cap * 2); r.cross.resize((size_t)cap * 3); r.le_l.resize((size_t)cap * 6);
        r.planes.resize((size_t)cap * 4); r.normals.resize((size_t)cap * 3); r.lineNo.resize((size_t)cap * 2);
        r.cross3d.resize((size_t)cap * 3); r.cross2d.resize((size_t)cap * 2);
        int k = 0, p = 0;
        check(pslfe_glue_fetch(h_, frame, n, r.lines3d.data(), r.lineEq.data(), r.pair.data(), r.xy.data(), r.cross.data(), r.le_l.data(), cap, &k,
                               r.planes.data(), r.normals.data(), r.lineNo.data(), r.cross3d.data(), r.cross2d.data(), cap, &p),
              "pslfe_glue_fetch");
        r.pair.resize((size_t)k * 2); r.xy.resize((size_t)k * 2); r.cross.resize((size_t)k * 3); r.le_l.resize((size_t)k * 6);
        r.planes.resize((size_t)p * 4); r.normals.resize((size_t)p * 3); r.lineNo.resize((size_t)p * 2);
        r.cross3d.resize((size_t)p * 3); r.cross2d.resize((size_t)p * 2);
        return r;
    }
    pslfe_glue* get() const { return h_; }
private:
    pslfe_glue* h_ = nullptr;
    int maxFans_;
};

// == Look-ahead extraction for the Tracking loop.  The reference's caller is a sequential loop over the frames of a recording
//    (Examples/RGB-D/rgbd_tum.cc:88-130) that constructs one Frame per image (src/Tracking.cc:240 -> src/Frame.cc:133-208); nothing the
//    Frame constructor computes depends on the pose of an earlier frame, so the extraction of frames t+1 .. t+K can run in ONE batched
//    launch while frame t is being tracked.  push() stages a frame (gray + CV_32F depth, copied to HBM); a full batch of `lookahead`
//    frames is launched at once -
//        pslfe_orb_extract_batch_device, pslfe_line_extract_batch_device, pslfe_line_pair_batch_device,
//        pslfe_glue_run_batch_device, pslfe_frame_set_from_orb_rgbd, pslfe_record_pack_device
//    - asynchronously, on one of TWO lanes (each with its own context / stream, extractor objects and frame grid): while the tracker
//    works through the frames of one lane, the next batch is extracted on the other.  pop() hands out the oldest frame's Frame
//    members (the packed per-frame records come back in one copy per batch).  A frame's grid slot - grid() after its pop(), or
//    frame.grid / frame.slot - stays valid until `lookahead` further frames have been popped: call the matchers for a frame before
//    popping the next one (as a Tracking thread does).  isLineGood's rand() stream is seeded with 1 + the frame's running index
//    (convention H7), so the results are those of the one-frame-at-a-time path, bit for bit, whatever the look-ahead.
//    A live camera pays up to 2 K - 1 frames of latency for this; a recording pays nothing.
class FramePrefetcher {
public:
    struct Frame {   // the members of ORB_SLAM2::Frame this path fills
        uint64_t index = 0;          // running index of the frame (push order)
        int slot = 0;                // its slot in *grid: SearchByProjection(*frame.grid, frame.slot, ...)
        FrameGrid* grid = nullptr;
        std::vector<PslKeyPoint> mvKeys, mvKeysUn;
        std::vector<uint8_t> mDescriptors;
        std::vector<float> mvDepth, mvuRight;
        std::vector<PslKeyLine> mvKeylinesUn;
        std::vector<uint8_t> mLdesc;
        std::vector<double> mvKeyLineFunctions;
        std::vector<float> fans;
        FrameGlue::Result glue;
    };

    FramePrefetcher(Context& ctx, int cols, int rows, int lookahead, int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                    int nLSDFeature, const PslCamera& cam, float pairRadius = 20.0f, float fanThr = 0.78539816339744830962f)
        : w_(cols), h_(rows), K_(lookahead < 1 ? 1 : lookahead), cam_(cam), radius_(pairRadius), fanThr_(fanThr) {
        for (int l = 0; l < 2; ++l) {
            Lane& L = lane_[l];
            L.own = new Context(ctx.device());   // a context (= a stream) of its own per lane: the caller's context stays free for the per-frame
            L.ctx = L.own;                       // calls of the tracker (LSDmatcher, plane association ...) while a lane extracts
            L.orb = new ORBextractor(*L.ctx, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, K_);
            L.lsd = new LINEextractor(*L.ctx, 1, 1.2f, (unsigned)nLSDFeature, 0.0, K_);
            kpCap_ = pslfe_orb_max_keypoints(L.orb->get(), cols, rows);
            if (kpCap_ < 0) throw Error(kpCap_, "pslfe_orb_max_keypoints");
            L.grid = new FrameGrid(*L.ctx, kpCap_ > 0 ? kpCap_ : 1, K_);
            check(pslfe_device_alloc(L.ctx->get(), (size_t)K_ * w_ * h_, &L.d_gray), "pslfe_device_alloc");
            check(pslfe_device_alloc(L.ctx->get(), (size_t)K_ * w_ * h_ * sizeof(float), &L.d_depth), "pslfe_device_alloc");
            caps_.kp_cap = kpCap_; caps_.kl_cap = nLSDFeature; caps_.fan_cap = 4096; caps_.plane_cap = 0;
            check(pslfe_record_layout(&caps_, &lay_), "pslfe_record_layout");
            check(pslfe_device_alloc(L.ctx->get(), (size_t)K_ * lay_.bytes, &L.d_rec), "pslfe_device_alloc");
            L.rec.resize((size_t)K_ * lay_.bytes);
        }
    }
    ~FramePrefetcher() {
        for (int l = 0; l < 2; ++l) {
            Lane& L = lane_[l];
            if (L.ctx) {
                pslfe_ctx_synchronize(L.ctx->get());
                pslfe_device_free(L.ctx->get(), L.d_gray); pslfe_device_free(L.ctx->get(), L.d_depth); pslfe_device_free(L.ctx->get(), L.d_rec);
            }
            delete L.glue; delete L.grid; delete L.lsd; delete L.orb; delete L.own;
        }
    }
    FramePrefetcher(const FramePrefetcher&) = delete;
    FramePrefetcher& operator=(const FramePrefetcher&) = delete;

    int lookahead() const { return K_; }
    // frames pushed and not yet popped / of those, the ones whose extraction has been launched or collected
    size_t staged() const { return (size_t)(lane_[0].staged + lane_[1].staged - lane_[0].next - lane_[1].next); }
    size_t ready() const { return (size_t)((lane_[0].collected ? lane_[0].staged - lane_[0].next : 0) + (lane_[1].collected ? lane_[1].staged - lane_[1].next : 0)); }
    FrameGrid& grid() { return *lane_[last_].grid; }   // of the frame popped last
    ORBextractor& orbExtractor() { return *lane_[0].orb; }
    LINEextractor& lineExtractor() { return *lane_[0].lsd; }

    // true when push() would accept a frame now (a lane is free or being filled)
    bool canPush() const {
        const Lane& L = lane_[stage_];
        return !L.launched || (L.collected && L.next == L.staged);
    }

    // Stages one frame: gray 8UC1 (`grayStep` bytes per row), depth CV_32F in metres (`depthStep` floats per row); the batch is launched
    // when `lookahead` frames are staged.  Returns false (and stages nothing) while both lanes hold frames that have not been popped.
    bool push(const uint8_t* gray, int grayStep, const float* depth, int depthStep) {
        if (!gray || !depth) return false;
        Lane& L = lane_[stage_];
        if (L.launched) {
            if (!(L.collected && L.next == L.staged)) return false;   // still being popped (or not popped at all)
            L.launched = L.collected = false; L.staged = L.next = 0;   // every frame of this lane has been handed out: it takes the next batch
        }
        uint8_t* dg = static_cast<uint8_t*>(L.d_gray) + (size_t)L.staged * w_ * h_;
        float* dd = static_cast<float*>(L.d_depth) + (size_t)L.staged * w_ * h_;
        if (grayStep != w_) {   // rows with padding: packed on the host first, one copy either way
            tmp8_.resize((size_t)w_ * h_);
            for (int y = 0; y < h_; ++y) memcpy(tmp8_.data() + (size_t)y * w_, gray + (size_t)y * grayStep, (size_t)w_);
            gray = tmp8_.data();
        }
        if (depthStep != w_) {
            tmp32_.resize((size_t)w_ * h_);
            for (int y = 0; y < h_; ++y) memcpy(tmp32_.data() + (size_t)y * w_, depth + (size_t)y * depthStep, (size_t)w_ * sizeof(float));
            depth = tmp32_.data();
        }
        check(pslfe_device_upload(L.ctx->get(), dg, gray, (size_t)w_ * h_), "pslfe_device_upload");
        check(pslfe_device_upload(L.ctx->get(), dd, depth, (size_t)w_ * h_ * sizeof(float)), "pslfe_device_upload");
        if (++L.staged == K_) { launch(stage_); stage_ ^= 1; }
        return true;
    }

    // Launches the extraction of a partly filled batch now (pop() does it when it runs out of launched frames).
    void flush() {
        Lane& L = lane_[stage_];
        if (!L.launched && L.staged > 0) { launch(stage_); stage_ ^= 1; }
    }

    // The oldest frame not handed out yet; false when nothing is staged.  Lanes are filled and emptied alternately, so the oldest frame is
    // always in lane cur_.
    bool pop(Frame& out) {
        Lane& L = lane_[cur_];
        if (L.launched && L.collected && L.next == L.staged) return false;   // used up, and nothing has been pushed since
        if (!L.launched) {                                                   // a partly filled batch: extract it now
            if (L.staged == 0) return false;
            launch(cur_);
            if (stage_ == cur_) stage_ ^= 1;
        }
        if (!L.collected) {   // the batch's packed records, one copy (waits for the lane's stream)
            check(pslfe_device_download(L.ctx->get(), L.rec.data(), L.d_rec, (size_t)L.staged * lay_.bytes), "pslfe_device_download");
            L.collected = true;
        }
        const int k = L.next++;
        const uint8_t* r = L.rec.data() + (size_t)k * lay_.bytes;
        int32_t hd[8];
        memcpy(hd, r, sizeof(hd));
        const int nkp = hd[0], nkl = hd[2], nfan = hd[4];
        if ((hd[6] & 7) != 0) throw Error(PSLFE_E_CAPACITY, "FramePrefetcher: a frame's results exceed the record capacities");
        out.index = L.index0 + (uint64_t)k; out.slot = k; out.grid = L.grid;
        out.mvKeys.resize(nkp); out.mDescriptors.resize((size_t)nkp * 32);
        out.mvKeylinesUn.resize(nkl); out.mLdesc.resize((size_t)nkl * 32); out.mvKeyLineFunctions.resize((size_t)nkl * 3);
        out.fans.resize((size_t)nfan * 4);
        if (nkp) { memcpy(out.mvKeys.data(), r + lay_.off_kps, (size_t)nkp * sizeof(PslKeyPoint)); memcpy(out.mDescriptors.data(), r + lay_.off_desc, (size_t)nkp * 32); }
        if (nkl) {
            memcpy(out.mvKeylinesUn.data(), r + lay_.off_kls, (size_t)nkl * sizeof(PslKeyLine));
            memcpy(out.mLdesc.data(), r + lay_.off_ldesc, (size_t)nkl * 32);
            memcpy(out.mvKeyLineFunctions.data(), r + lay_.off_lineEq, (size_t)nkl * 3 * sizeof(double));
        }
        if (nfan) memcpy(out.fans.data(), r + lay_.off_fans, (size_t)nfan * 4 * sizeof(float));
        out.glue = L.glue->fetch(k, nkl);                                            // mvLines3D, crossings, mvPlanes ...
        if (nkp) L.grid->fetch(k, out.mvKeysUn, out.mvDepth, out.mvuRight, kpCap_);   // mvKeysUn, mvDepth, mvuRight
        else { out.mvKeysUn.clear(); out.mvDepth.clear(); out.mvuRight.clear(); }
        last_ = cur_;
        if (L.next == L.staged) cur_ ^= 1;   // this lane is used up: the next frame is the other lane's first
        return true;
    }

private:
    struct Lane {
        Context* own = nullptr;
        Context* ctx = nullptr;
        ORBextractor* orb = nullptr;
        LINEextractor* lsd = nullptr;
        FrameGrid* grid = nullptr;
        FrameGlue* glue = nullptr;   // created after the first batch: its row stride is the line extractor's (pslfe_line_results_device)
        void* d_gray = nullptr; void* d_depth = nullptr; void* d_rec = nullptr;
        std::vector<uint8_t> rec;
        int staged = 0, next = 0;            // frames staged in this lane / handed out
        bool launched = false, collected = false;
        uint64_t index0 = 0;
    };

    // every staged frame of the lane through the extractors, asynchronously on the lane's stream
    void launch(int l) {
        Lane& L = lane_[l];
        const int F = L.staged;
        const uint8_t* dg = static_cast<const uint8_t*>(L.d_gray);
        const float* dd = static_cast<const float*>(L.d_depth);
        check(pslfe_orb_extract_batch_device(L.orb->get(), dg, F, w_, h_, w_, (size_t)w_ * h_), "pslfe_orb_extract_batch_device");       // ExtractORB
        check(pslfe_line_extract_batch_device(L.lsd->get(), dg, F, w_, h_, w_, (size_t)w_ * h_), "pslfe_line_extract_batch_device");   // ExtractLSD: extractor
        check(pslfe_line_pair_batch_device(L.lsd->get(), radius_, fanThr_), "pslfe_line_pair_batch_device");                              // src/Frame.cc:505
        PslRecordSources S;
        memset(&S, 0, sizeof(S));
        check(pslfe_orb_results_device(L.orb->get(), &S.d_kps, &S.d_desc, &S.d_kp_counts, &S.kp_stride), "pslfe_orb_results_device");
        check(pslfe_line_results_device(L.lsd->get(), &S.d_kls, &S.d_ldesc, &S.d_lineEq, &S.d_kl_counts, &S.kl_stride), "pslfe_line_results_device");
        check(pslfe_line_fans_device(L.lsd->get(), &S.d_fans, &S.d_fan_counts, &S.fan_stride), "pslfe_line_fans_device");
        if (!L.glue) L.glue = new FrameGlue(*L.ctx, S.kl_stride, S.fan_stride, K_);
        check(pslfe_glue_run_batch_device(L.glue->get(), F, S.d_kls, S.kl_stride, S.d_kl_counts, S.d_fans, S.fan_stride, S.d_fan_counts, dd, w_, h_, &cam_,
                                          (uint32_t)(1u + next_index_)), "pslfe_glue_run_batch_device");                               // isLineGood, fans, planes
        check(pslfe_frame_set_from_orb_rgbd(L.grid->get(), L.orb->get(), dd, w_, h_, &cam_), "pslfe_frame_set_from_orb_rgbd");          // Undistort .. AssignFeaturesToGrid
        check(pslfe_record_pack_device(L.ctx->get(), &caps_, &S, F, L.d_rec), "pslfe_record_pack_device");
        L.index0 = next_index_;
        next_index_ += (uint64_t)F;
        L.launched = true; L.collected = false; L.next = 0;
    }

    int w_, h_, K_;
    PslCamera cam_;
    float radius_, fanThr_;
    int kpCap_ = 0;
    Lane lane_[2];
    PslRecordCaps caps_;
    PslRecordLayout lay_;
    std::vector<uint8_t> tmp8_;
    std::vector<float> tmp32_;
    int stage_ = 0, cur_ = 0, last_ = 0;   // the lane being filled / popped from next / of the frame popped last
    uint64_t next_index_ = 0;
};

// DBoW2 ORBVocabulary as far as Frame::ComputeBoW needs it (src/Frame.cc:1053-1060): the tree as flat arrays, transform on the device.
class ORBVocabulary {
public:
    struct Result {
        std::vector<int32_t> word, nid;        // per feature
        std::vector<double> weight;
        std::vector<int32_t> bowId;            // mBowVec, ascending
        std::vector<double> bowVal;
        std::vector<int32_t> fvNode, fvStart, fvIdx;  // mFeatVec: node g owns fvIdx[fvStart[g] .. fvStart[g+1])
    };
    // node i: children childIds[childBegin[i] .. + childCount[i]) in Node::children order; node 0 = root; L = m_L
    ORBVocabulary(Context& ctx, const std::vector<int32_t>& childBegin, const std::vector<int32_t>& childCount, const std::vector<int32_t>& childIds,
                  const std::vector<uint8_t>& nodeDesc, const std::vector<double>& nodeWeight, const std::vector<int32_t>& nodeWord, int L) {
        check(pslfe_vocab_create(ctx.get(), (int)childBegin.size(), childBegin.data(), childCount.data(), childIds.data(), (int)childIds.size(),
                                 nodeDesc.data(), nodeWeight.data(), nodeWord.data(), L, &h_), "pslfe_vocab_create");
    }
    ~ORBVocabulary() { pslfe_vocab_destroy(h_); }
    ORBVocabulary(const ORBVocabulary&) = delete;
    ORBVocabulary& operator=(const ORBVocabulary&) = delete;
    Result transform(const std::vector<uint8_t>& descriptors, int levelsup = 4) {
        const int n = (int)descriptors.size() / 32;
        Result r;
        r.word.resize(n); r.nid.resize(n); r.weight.resize(n); r.bowId.resize(n); r.bowVal.resize(n);
        r.fvNode.resize(n); r.fvStart.resize(n + 1); r.fvIdx.resize(n);
        int nb = 0, nf = 0;
        check(pslfe_compute_bow(h_, descriptors.data(), n, levelsup, r.word.data(), r.weight.data(), r.nid.data(), r.bowId.data(), r.bowVal.data(),
                                &nb, r.fvNode.data(), r.fvStart.data(), r.fvIdx.data(), &nf), "pslfe_compute_bow");
        r.bowId.resize(nb); r.bowVal.resize(nb); r.fvNode.resize(nf); r.fvStart.resize(nf + 1);
        r.fvIdx.resize(nf ? r.fvStart[nf] : 0);
        return r;
    }
private:
    pslfe_vocab* h_ = nullptr;
};

// == the KeyFrame-rate searches of LocalMapping / LoopClosing (pslfe_kf): ORBmatcher::Fuse (both overloads), SearchBySim3,
//    SearchForTriangulation (src/ORBmatcher.cc:657-1326), the search of LSDmatcher::Fuse (add_src/LSDmatcher.cpp:933-958) and
//    Map{Point,Line}::ComputeDistinctiveDescriptors, from the point where the host has projected its map points.  One per host
//    thread, on that thread's own Context.
class KeyFrameMatcher {
public:
    static constexpr int TH_HIGH = 100, TH_LOW = 50;
    explicit KeyFrameMatcher(Context& ctx) { check(pslfe_kf_create(ctx.get(), &h_), "pslfe_kf_create"); }
    ~KeyFrameMatcher() { pslfe_kf_destroy(h_); }
    KeyFrameMatcher(const KeyFrameMatcher&) = delete;
    KeyFrameMatcher& operator=(const KeyFrameMatcher&) = delete;

    // candidate loop of Fuse / SearchBySim3; invLevelSigma2 != nullptr selects the reprojection gates of Fuse(pKF, vpMapPoints, th)
    void WindowBest(FrameGrid& kf, int slot, const std::vector<PslProjQuery>& q, const std::vector<uint8_t>& qdesc,
                    const std::vector<float>* invLevelSigma2, std::vector<int32_t>& bestIdx, std::vector<int32_t>& bestDist) {
        bestIdx.assign(q.size(), -1); bestDist.assign(q.size(), 0x7fffffff);
        check(pslfe_kf_window_best(h_, kf.get(), slot, q.data(), qdesc.data(), (int)q.size(), invLevelSigma2 ? 1 : 0,
                                   invLevelSigma2 ? invLevelSigma2->data() : nullptr, invLevelSigma2 ? (int)invLevelSigma2->size() : 0,
                                   bestIdx.data(), bestDist.data()), "pslfe_kf_window_best");
    }
    int SearchBySim3(FrameGrid& kf1, int slot1, FrameGrid& kf2, int slot2, const std::vector<PslProjQuery>& q12, const std::vector<uint8_t>& qdesc1,
                     const std::vector<PslProjQuery>& q21, const std::vector<uint8_t>& qdesc2, std::vector<int32_t>& match12) {
        match12.assign(q12.size(), -1);
        int nf = 0;
        check(pslfe_kf_search_by_sim3(h_, kf1.get(), slot1, kf2.get(), slot2, q12.data(), qdesc1.data(), (int)q12.size(), q21.data(), qdesc2.data(),
                                      (int)q21.size(), match12.data(), &nf), "pslfe_kf_search_by_sim3");
        return nf;
    }
    int SearchForTriangulation(FrameGrid& kf2, int slot2, const std::vector<int32_t>& fidx2, const std::vector<uint8_t>& taken2,
                               const std::vector<PslTriQuery>& q, const std::vector<uint8_t>& qdesc, const float F12[9], float ex, float ey,
                               bool bOnlyStereo, bool checkOrientation, const std::vector<float>& scaleFactors, const std::vector<float>& levelSigma2,
                               std::vector<int32_t>& match) {
        match.assign(q.size(), -1);
        int nm = 0;
        check(pslfe_kf_search_for_triangulation(h_, kf2.get(), slot2, fidx2.data(), (int)fidx2.size(), taken2.data(), q.data(), qdesc.data(),
                                                (int)q.size(), F12, ex, ey, bOnlyStereo, checkOrientation, scaleFactors.data(), levelSigma2.data(),
                                                (int)scaleFactors.size(), match.data(), &nm), "pslfe_kf_search_for_triangulation");
        return nm;
    }
    void LineFuse(const std::vector<PslKeyLine>& kls, const std::vector<uint8_t>& desc, const std::vector<PslLineFuseQuery>& q,
                  const std::vector<uint8_t>& qdesc, std::vector<int32_t>& bestIdx, std::vector<int32_t>& bestDist) {
        bestIdx.assign(q.size(), -1); bestDist.assign(q.size(), 256);
        check(pslfe_kf_line_fuse_best(h_, kls.data(), (int)kls.size(), desc.data(), (int)desc.size() / 32, q.data(), qdesc.data(), (int)q.size(),
                                      bestIdx.data(), bestDist.data()), "pslfe_kf_line_fuse_best");
    }
    // offsets: npts + 1 entries; returns the best row of every point relative to its run
    std::vector<int32_t> ComputeDistinctiveDescriptors(const std::vector<uint8_t>& desc, const std::vector<int32_t>& offsets) {
        std::vector<int32_t> best(offsets.empty() ? 0 : offsets.size() - 1, -1);
        check(pslfe_kf_distinctive_descriptors(h_, desc.data(), offsets.data(), (int)best.size(), best.data()), "pslfe_kf_distinctive_descriptors");
        return best;
    }
private:
    pslfe_kf* h_ = nullptr;
};

}  // namespace pslfe
#endif
