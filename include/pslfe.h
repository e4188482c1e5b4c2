/* pslfe — MI355X-native per-frame feature front-end for PSL-SLAM: the C ABI.
 *
 * This is the drop-in boundary.  The reference has no FFI: its seams are plain
 * C++ calls on objects owned by Tracking (ORBextractor::operator(), LINEextractor::operator(),
 * CPartiallyRecoverConnectivity, ORBmatcher / LSDmatcher methods).  Each entry point below names
 * the reference interface it replaces (file:line under the reference tree).  A C++ shim that
 * mirrors those classes over this ABI is in psl-slam_amd/host/pslfe.hpp; the patch a PSL-SLAM
 * maintainer applies is in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes only; every function returns PSLFE_OK (0) or a negative
 * PSLFE_E_* code and never throws; the caller allocates outputs with a capacity and the callee
 * reports counts; structs are POD with the exact layout of the OpenCV types the reference uses
 * (cv::KeyPoint 28 B, line_descriptor::KeyLine 68 B); one context per GPU; calls on one handle are
 * serialised by the caller (the reference has a single Tracking thread), different handles may be
 * used from different threads.  "d_" arguments are device (HBM) pointers, all others are host.
 * The library needs a gfx950 GPU: there is no CPU fallback, ctx creation fails without one.
 */
#ifndef PSLFE_H
#define PSLFE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSLFE_OK 0
#define PSLFE_E_INVALID (-1)   /* bad argument (null, size, unsupported configuration)        */
#define PSLFE_E_NODEVICE (-2)  /* no usable gfx950 device / HIP runtime error at start-up     */
#define PSLFE_E_HIP (-3)       /* HIP runtime error; text in pslfe_last_error()               */
#define PSLFE_E_CAPACITY (-4)  /* caller-provided capacity too small                          */
#define PSLFE_E_STATE (-5)     /* call order (e.g. fetch before extract)                      */

#define PSLFE_MAX_LEVELS 16

/* == cv::KeyPoint as filled by ORBextractor (src/ORBextractor.cc:837-847, 1095-1103). */
typedef struct PslKeyPoint {
    float x, y;      /* pt, level-0 pixel coordinates (pt *= mvScaleFactor[octave])            */
    float size;      /* (int)(31 * mvScaleFactor[octave])                                      */
    float angle;     /* IC_Angle, degrees in [0,360)                                            */
    float response;  /* FAST corner score (not Harris: include/ORBextractor.h:49 is unused)    */
    int32_t octave;
    int32_t class_id; /* -1 */
} PslKeyPoint;

/* == cv::line_descriptor::KeyLine, field order of
 * Thirdparty/line_descriptor/include/line_descriptor/descriptor_custom.hpp:107-146. */
typedef struct PslKeyLine {
    float angle;
    int32_t class_id;
    int32_t octave;
    float pt_x, pt_y;
    float response;
    float size;
    float startPointX, startPointY, endPointX, endPointY;
    float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
    float lineLength;
    int32_t numOfPixels;
} PslKeyLine;

typedef struct pslfe_ctx pslfe_ctx;   /* one per GPU                                           */
typedef struct pslfe_orb pslfe_orb;   /* == ORBextractor object                                 */

const char* pslfe_version(void);
/* Last error text of this thread (empty string if none). */
const char* pslfe_last_error(void);

/* ---- context ------------------------------------------------------------------------------- */
int pslfe_ctx_create(int device, pslfe_ctx** out);
void pslfe_ctx_destroy(pslfe_ctx* ctx);
/* Run all work of this context on an existing HIP stream (hipStream_t passed as void*), e.g. the
 * caller framework's current stream; NULL = the context's own stream. */
int pslfe_ctx_set_stream(pslfe_ctx* ctx, void* hip_stream);
int pslfe_ctx_synchronize(pslfe_ctx* ctx);
/* Per-stage device timing with HIP events on the launch stream (for bench/roofline).
 * enable != 0 starts collecting; pslfe_ctx_stage_time returns accumulated milliseconds and the
 * number of launches of a stage name ("orb.pyramid", "orb.fast", "orb.octree", "orb.blur",
 * "orb.describe", "match.window", "match.knn2", ...) and resets nothing. */
int pslfe_ctx_profile(pslfe_ctx* ctx, int enable);
/* Restrict the timing to one stage (NULL or "" = all stages).  Every timed stage puts two event records
 * between kernels (~10 us of idle GPU each on MI355X); timing only the stage of interest keeps a
 * throughput measurement undisturbed. */
int pslfe_ctx_profile_only(pslfe_ctx* ctx, const char* stage);
int pslfe_ctx_profile_reset(pslfe_ctx* ctx);
int pslfe_ctx_stage_time(pslfe_ctx* ctx, const char* stage, double* ms_total, int* launches);

/* Plain HBM helpers for callers without their own device allocator (synchronous copies). */
int pslfe_device_alloc(pslfe_ctx* ctx, size_t bytes, void** d_ptr);
int pslfe_device_free(pslfe_ctx* ctx, void* d_ptr);
int pslfe_device_upload(pslfe_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int pslfe_device_download(pslfe_ctx* ctx, void* dst, const void* d_src, size_t bytes);

/* ---- input conversions (Tracking::GrabImageRGBD src/Tracking.cc:214-240) --------------------- */
/* == cvtColor(im, gray, CV_RGB2GRAY | CV_BGR2GRAY) src/Tracking.cc:219-232 on 8UC3 frames (the 4-channel
 *    variants :226-231 drop alpha first and are the same arithmetic).  is_rgb: mbRGB.  Frame f at
 *    d_rgb + f*frame_stride, rows `stride` bytes apart; d_gray packed [nframes][h][w]. Asynchronous. */
int pslfe_rgb_to_gray_device(pslfe_ctx* ctx, const uint8_t* d_rgb, int nframes, int w, int h, int stride,
                             size_t frame_stride, int is_rgb, uint8_t* d_gray);
int pslfe_rgb_to_gray(pslfe_ctx* ctx, const uint8_t* rgb, int w, int h, int stride, int is_rgb, uint8_t* gray);
/* == imDepth.convertTo(imDepth, CV_32F, mDepthMapFactor) src/Tracking.cc:234-235 on CV_16U depth. */
int pslfe_depth_to_float_device(pslfe_ctx* ctx, const uint16_t* d_depth, size_t n, float factor, float* d_out);
int pslfe_depth_to_float(pslfe_ctx* ctx, const uint16_t* depth, size_t n, float factor, float* out);

/* ---- ORB extractor -------------------------------------------------------------------------- */
/* == ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 *    src/ORBextractor.cc:410-470; object created once in Tracking (src/Tracking.cc:120).
 *    max_batch = most frames one batched call will carry (>=1). */
int pslfe_orb_create(pslfe_ctx* ctx, int nfeatures, float scaleFactor, int nlevels, int iniThFAST,
                     int minThFAST, int max_batch, pslfe_orb** out);
void pslfe_orb_destroy(pslfe_orb* orb);

/* == GetLevels/GetScaleFactor/GetScaleFactors/GetInverseScaleFactors/GetScaleSigmaSquares/
 *    GetInverseScaleSigmaSquares, include/ORBextractor.h:63-84 (read by Frame ctor src/Frame.cc:141-148).
 *    Arrays hold nlevels floats. */
int pslfe_orb_levels(const pslfe_orb* orb);
float pslfe_orb_scale_factor(const pslfe_orb* orb);
int pslfe_orb_scale_factors(const pslfe_orb* orb, float* scale, float* inv_scale, float* sigma2,
                            float* inv_sigma2);
/* mnFeaturesPerLevel (src/ORBextractor.cc:435-446), nlevels ints. */
int pslfe_orb_features_per_level(const pslfe_orb* orb, int* quota);
/* Upper bound on keypoints one frame can return for a w x h image (quota + octree overshoot). */
int pslfe_orb_max_keypoints(pslfe_orb* orb, int w, int h);

/* == ORBextractor::operator()(image, mask, keypoints, descriptors)
 *    include/ORBextractor.h:59-61, src/ORBextractor.cc:1043-1105; called from Frame::ExtractORB
 *    src/Frame.cc:311-317.  One 8UC1 host image in, host outputs; synchronous.
 *    gray==NULL or w/h<=0 -> PSLFE_OK with *n = 0 (reference returns silently, :1046).
 *    desc: n x 32 bytes row-major.  cap = capacity of kps/desc in keypoints. */
int pslfe_orb_extract(pslfe_orb* orb, const uint8_t* gray, int w, int h, int stride,
                      PslKeyPoint* kps, uint8_t* desc, int cap, int* n);

/* Batched many-frames mode (north_star): nframes independent 8UC1 frames already resident in HBM,
 * frame f at d_gray + f*frame_stride, rows `stride` bytes apart.  Asynchronous on the context's
 * stream; results stay in the handle's HBM buffers until the next call. */
int pslfe_orb_extract_batch_device(pslfe_orb* orb, const uint8_t* d_gray, int nframes, int w, int h,
                                   int stride, size_t frame_stride);
/* Device views of the last batch's results: keypoints [nframes][cap] PslKeyPoint, descriptors
 * [nframes][cap][32] u8, counts [nframes] int32; cap = *kp_cap. Valid until the next extract. */
int pslfe_orb_results_device(pslfe_orb* orb, const PslKeyPoint** d_kps, const uint8_t** d_desc,
                             const int32_t** d_counts, int* kp_cap);
/* Copy frame `frame` of the last batch to the host (synchronises the stream). */
int pslfe_orb_fetch(pslfe_orb* orb, int frame, PslKeyPoint* kps, uint8_t* desc, int cap, int* n);
/* Host-buffer convenience: H2D + batch extract + D2H of everything.  kps [nframes][cap],
 * desc [nframes][cap][32], counts [nframes]. */
int pslfe_orb_extract_batch(pslfe_orb* orb, const uint8_t* gray, int nframes, int w, int h, int stride,
                            size_t frame_stride, PslKeyPoint* kps, uint8_t* desc, int cap, int32_t* counts);

/* Stage taps of the last batch, for parity tests against the oracle (host outputs, synchronous):
 *  level image (blurred = 0: pyramid level == mvImagePyramid[level] ROI; 1: the 7x7 sigma-2 blur),
 *  FAST candidates of a level in reference order (x, y relative to minBorder, score) == the
 *  vToDistributeKeys of src/ORBextractor.cc:779-829, and keypoints after DistributeOctTree. */
int pslfe_orb_debug_level_size(pslfe_orb* orb, int level, int* w, int* h);
int pslfe_orb_debug_level_image(pslfe_orb* orb, int frame, int level, int blurred, uint8_t* out, int out_stride);
int pslfe_orb_debug_candidates(pslfe_orb* orb, int frame, int level, int32_t* xys, int cap, int* n);
int pslfe_orb_debug_level_keypoints(pslfe_orb* orb, int frame, int level, int32_t* xys, int cap, int* n);

/* ---- line extractor ---------------------------------------------------------------------------- */
typedef struct pslfe_line pslfe_line;  /* == LINEextractor object                                  */

/* == LINEextractor::LINEextractor(numOctaves, scale, nLSDFeature, min_line_length)
 *    add_src/LineExtractor.cpp:6-25; object created once in Tracking (src/Tracking.cc:127).
 *    numOctaves must be 1 (PSLFE_E_INVALID otherwise): every reference YAML sets LINEextractor.nLevels: 1, and the
 *    contrib detect() call truncates scale 1.2 to int 1 (add_src/LineExtractor.cpp:336-337), with which the stock
 *    LSDDetector's pyramid for numOctaves > 1 is pyrDown(m, m, Size(cols / 1, rows / 1)) - rejected by pyrDown's size
 *    assertion: the reference's own call throws there, it does not yield lines. */
int pslfe_line_create(pslfe_ctx* ctx, int numOctaves, float scale, int nLSDFeature, double min_line_length,
                      int max_batch, pslfe_line** out);
void pslfe_line_destroy(pslfe_line* line);
/* Refinement mode of the LSD behind the extractor (cv::createLineSegmentDetector(refine), OpenCV 3.x lsd.cpp).  The
 * reference calls the STOCK contrib cv::line_descriptor::LSDDetector (add_src/LineExtractor.cpp:336-337, linked by
 * CMakeLists.txt:96), whose source is not in its tree: upstream constructs the detector with LSD_REFINE_ADV
 * (rect_improve + NFA test, log_eps 0), which is the default here; the vendored, never-called LSDDetectorC
 * (Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:185) uses LSD_REFINE_STD (no NFA test). */
#define PSLFE_LSD_REFINE_STD 1
#define PSLFE_LSD_REFINE_ADV 2
int pslfe_line_set_refine(pslfe_line* line, int refine);
/* == GetLevels / GetScaleFactor / GetScaleFactors ... add_inc/LineExtractor.h:211-233 */
int pslfe_line_levels(const pslfe_line* line);
float pslfe_line_scale_factor(const pslfe_line* line);
int pslfe_line_scale_factors(const pslfe_line* line, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2);

/* == line_descriptor::LSDDetector::detect(image, keylines, scale=1, numOctaves=1) up to the segment
 *    list: cv::createLineSegmentDetector() defaults + checkLineExtremes
 *    (Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:166-205). segments: n x (x1,y1,x2,y2). */
int pslfe_lsd_detect(pslfe_line* line, const uint8_t* gray, int w, int h, int stride, float* segments, int cap, int* n);
/* Tap for parity tests: LSD working image (f64), level-line angle in degrees (f32, -1024 = NOTDEF;
 * the reference's double angle is exactly (double)deg * CV_PI/180) and gradient norm (f64). */
int pslfe_line_debug_gradient(pslfe_line* line, int frame, int* W, int* H, double* scaled, float* angle_deg, double* modgrad);

/* == LINEextractor::operator()(image, mask, keylines, descriptors, lineVec2d)
 *    add_inc/LineExtractor.h:167, add_src/LineExtractor.cpp:325-366; called from Frame::ExtractLSD
 *    src/Frame.cc:494.  LSD -> optimizeAndMergeLines_lsd -> top nLSDFeature by response -> LBD -> 2-D
 *    line equations.  desc: n x 32 bytes; lineEq: n x 3 doubles (sp x ep normalised by |xy|).
 *    gray == NULL or w/h <= 0 -> PSLFE_OK with *n = 0 (:327).  mask is not part of the ABI: the
 *    reference always passes an empty one (src/Frame.cc:493-494). */
int pslfe_line_extract(pslfe_line* line, const uint8_t* gray, int w, int h, int stride, PslKeyLine* kls, uint8_t* desc,
                       double* lineEq, int cap, int* n);
/* Batched many-frames mode, HBM resident, asynchronous on the context's stream. */
int pslfe_line_extract_batch_device(pslfe_line* line, const uint8_t* d_gray, int nframes, int w, int h, int stride,
                                    size_t frame_stride);
/* Device views: keylines [nframes][cap], descriptors [nframes][cap][32], line equations
 * [nframes][cap][3] f64, counts [nframes]. */
int pslfe_line_results_device(pslfe_line* line, const PslKeyLine** d_kls, const uint8_t** d_desc, const double** d_lineEq,
                              const int32_t** d_counts, int* kl_cap);
/* Copy one frame of the last batch to the host. *status (may be NULL): 0, or bit 1 = more raw segments
 * than the merge stage holds, bit 2 = cluster list overflow, bit 4 = more merged lines than cap. */
int pslfe_line_fetch(pslfe_line* line, int frame, PslKeyLine* kls, uint8_t* desc, double* lineEq, int cap, int* n, int* status);

/* == optimizeAndMergeLines_lsd(keylines, img) add_src/uselongline.cpp:449-485 on a segment list
 *    (MergeLines 0.05/5/15 -> drop < 30 px -> MergeLines 0.03/3/30 -> drop < 50 px -> KeyLines). */
int pslfe_line_optimize_and_merge(pslfe_line* line, const float* segments, int nseg, int w, int h, PslKeyLine* kls, int cap, int* n);
/* == BinaryDescriptor::compute(image, keylines, descriptors) for given keylines
 *    (Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp:540-688, 1027-1373). fdesc (may be
 *    NULL): the 72-float LBD vectors before binarisation, n x 72. */
int pslfe_lbd_compute(pslfe_line* line, const uint8_t* gray, int w, int h, int stride, const PslKeyLine* kls, int nkl,
                      uint8_t* desc, float* fdesc);
/* Tap: Sobel dx, dy (s16, w x h) of the LBD pre-processing of the last call. */
int pslfe_line_debug_sobel(pslfe_line* line, int frame, int16_t* dx, int16_t* dy);

/* == CPartiallyRecoverConnectivity(mLines, radius, fans, img, fanThr)
 *    add_inc/PartiallyRecoverConnectivity.h:13, add_src/PartiallyRecoverConnectivity.cpp:14-133; called
 *    from Frame::ExtractLSD src/Frame.cc:505 with radius = 20, fanThr = pi/4.  lines: n x 4
 *    (x1,y1,x2,y2); fans: k x 4 rows (x, y, i, j) after the keep-last de-duplication. */
int pslfe_lil_pair(pslfe_line* line, const float* lines, int nlines, float radius, float fanThr, int imgCols, int imgRows,
                   float* fans, int cap, int* nfans);
/* The same for every frame of the last extracted batch (mLines = keyline end points), HBM resident. */
int pslfe_line_pair_batch_device(pslfe_line* line, float radius, float fanThr);
int pslfe_line_fans_fetch(pslfe_line* line, int frame, float* fans, int cap, int* nfans);
/* Device views of the fans of the last pslfe_line_pair_batch_device: [nframes][fan_stride][4] float, counts [nframes]. */
int pslfe_line_fans_device(pslfe_line* line, const float** d_fans, const int32_t** d_nfans, int* fan_stride);

/* == lmatcher.match(mLastFrame.mLdesc, mCurrentFrame.mLdesc, nnr, matches_12) of src/Tracking.cc:901
 *    (LSDmatcher::match -> matchNNR, add_src/LSDmatcher.cpp:354-413) for every frame f of the last
 *    extracted batch against frame (f - shift) mod nframes, HBM resident, asynchronous.
 *    d_matches12: [nframes][kl_cap] int32, row f indexed by the LAST frame's line; d_nmatches: [nframes]. */
int pslfe_line_match_batch_device(pslfe_line* line, int shift, float nnr, int32_t* d_matches12, int32_t* d_nmatches);

/* ---- descriptor matching --------------------------------------------------------------------- */
/* == cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, k=2) as used by LSDmatcher::matchNNR
 *    add_src/LSDmatcher.cpp:354-376 and FrameBFMatch :492-516.  256-bit descriptors, row-major
 *    32 B.  For query i: idx[2i], idx[2i+1] = best and second-best train rows (lower train index
 *    first on equal distance), dist[...] the Hamming distances; -1 / 0x7fffffff where nt < k. */
int pslfe_hamming_knn2(pslfe_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt,
                       int32_t* idx, int32_t* dist);
/* Same on HBM-resident descriptors, asynchronous on the context's stream. */
int pslfe_hamming_knn2_device(pslfe_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt,
                              int32_t* d_idx, int32_t* d_dist);

/* One projected query of ORBmatcher::SearchByProjection (src/ORBmatcher.cc:1328-1470 and :45-129):
 * what the host-side Tracking code knows about a map point before the window search. */
typedef struct PslProjQuery {
    float u, v;          /* projection into the current frame                                   */
    float radius;        /* th * mvScaleFactors[octave] (:1381) or r * scale (:66)               */
    float ur;            /* expected right coordinate u - mbf*invz (:1415) / mTrackProjXR (:92)  */
    int32_t min_level;   /* GetFeaturesInArea level band (:1385-1390, :66)                       */
    int32_t max_level;
    float angle;         /* LastFrame.mvKeysUn[i].angle, for the rotation histogram (:1433)      */
    int32_t blocks;      /* != 0: the map point has Observations()>0, so a keypoint it takes is
                            skipped by later queries (:1401-1403)                                */
} PslProjQuery;

typedef struct pslfe_frame pslfe_frame;  /* keypoints of up to max_frames frames, each bucketed on
                                             the 64x48 grid of include/Frame.h:45-46            */

/* == Frame::AssignFeaturesToGrid src/Frame.cc:269-284 + PosInGrid :1040-1050 for the keypoints of
 *    one frame, stored in `slot` (0 <= slot < max_frames).  Coordinates are the undistorted ones
 *    (mvKeysUn; PSL-SLAM's RGB-D YAMLs have zero distortion, so these are the extractor outputs).
 *    min_x..max_y = mnMinX..mnMaxY (src/Frame.cc:1135-1168).  uright: mvuRight or NULL (= all -1). */
int pslfe_frame_create(pslfe_ctx* ctx, int max_keypoints, int max_frames, pslfe_frame** out);
void pslfe_frame_destroy(pslfe_frame* f);
int pslfe_frame_set(pslfe_frame* f, int slot, const PslKeyPoint* kps, const uint8_t* desc, const float* uright,
                    int n, float min_x, float min_y, float max_x, float max_y);
/* All frames of the last batch of `orb` -> slots 0..nframes-1, HBM to HBM, asynchronous. */
int pslfe_frame_set_from_orb(pslfe_frame* f, pslfe_orb* orb, float min_x, float min_y, float max_x, float max_y);
/* Pinhole + radial-tangential camera of the settings YAML (Examples/RGB-D/TUM1.yaml; src/Tracking.cc:62-94):
 * mK entries, mDistCoef (k1 k2 p1 p2 k3) and mbf, all as the reference stores them (float). */
typedef struct PslCamera {
    float fx, fy, cx, cy;
    float k1, k2, p1, p2, k3;
    float bf;
} PslCamera;

/* == Frame::ComputeImageBounds src/Frame.cc:1135-1168: bounds = {mnMinX, mnMinY, mnMaxX, mnMaxY}
 *    (the argument order of pslfe_frame_set). */
int pslfe_image_bounds(pslfe_frame* f, const PslCamera* cam, int cols, int rows, float* bounds);

/* == The RGB-D part of the Frame constructor (src/Frame.cc:105-171) for one frame: UndistortKeyPoints
 *    :1062-1092 (cv::undistortPoints with P = K, skipped when k1 == 0 exactly as the reference does),
 *    ComputeStereoFromRGBD :1342-1363 (depth sampled at the DISTORTED keypoint, truncated to integer
 *    pixel; mvuRight = xUn - mbf/d, both -1 where d <= 0), ComputeImageBounds (first-frame statics) and
 *    AssignFeaturesToGrid on the undistorted points.  depth: CV_32F image (metres), host memory,
 *    depth_stride in floats.  The slot then holds mvKeysUn / mvuRight / mvDepth (pslfe_frame_fetch). */
int pslfe_frame_set_rgbd(pslfe_frame* f, int slot, const PslKeyPoint* kps, const uint8_t* desc, int n, const float* depth,
                         int width, int height, int depth_stride, const PslCamera* cam);
/* Same for every frame of the last batch of `orb` (slots 0..nframes-1), HBM to HBM, asynchronous.
 * d_depth: [nframes][height][width] float in HBM. */
int pslfe_frame_set_from_orb_rgbd(pslfe_frame* f, pslfe_orb* orb, const float* d_depth, int width, int height,
                                  const PslCamera* cam);
/* mvKeysUn, mvDepth, mvuRight of a slot (any pointer may be NULL). */
int pslfe_frame_fetch(pslfe_frame* f, int slot, PslKeyPoint* kps_un, float* depth, float* uright, int cap, int* n);

/* Tap for parity tests: CSR of mGrid in the order GetFeaturesInArea visits it (cell = ix*48+iy):
 * start[64*48+1], idx[n]. */
int pslfe_frame_debug_grid(pslfe_frame* f, int slot, int32_t* start, int32_t* idx, int cap, int* n);

/* == ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) src/ORBmatcher.cc:1328-1470
 *    after the host has projected the last frame's map points: queries[i] / qdesc[i] (32 B) describe
 *    map point i.  taken[c] != 0 (or NULL = none): current keypoint c already holds a map point with
 *    Observations()>0 and is skipped (:1401-1403).  Outputs: match[i] = current keypoint given to
 *    query i or -1 (TH_HIGH = 100 gate, then - if check_orientation - the rotation-histogram filter
 *    :1448-1467); assigned[c] (may be NULL) = query whose map point ends up in
 *    CurrentFrame.mvpMapPoints[c] or -1; *nmatches = the function's return value.  The reference's
 *    sequential first-come-first-served behaviour is reproduced exactly. */
int pslfe_orb_search_by_projection_last(pslfe_frame* cur, int slot, const PslProjQuery* queries, const uint8_t* qdesc,
                                        int nq, const uint8_t* taken, int check_orientation, int32_t* match,
                                        int32_t* assigned, int* nmatches);
/* == ORBmatcher::SearchByProjection(F, vpMapPoints, th) src/ORBmatcher.cc:45-129: best and second
 *    best in the window, ratio test `nnratio` only when both lie on the same octave (:118-125). */
int pslfe_orb_search_by_projection_map(pslfe_frame* cur, int slot, const PslProjQuery* queries, const uint8_t* qdesc,
                                       int nq, const uint8_t* taken, float nnratio, int32_t* match, int32_t* assigned,
                                       int* nmatches);
/* == ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) src/ORBmatcher.cc:1472-1599
 *    (relocalisation) after the host has projected the keyframe's map points (skipping bad ones and those in
 *    sAlreadyFound): window search as above, but every occupied keypoint is skipped (taken[c] != 0 <=>
 *    CurrentFrame.mvpMapPoints[c] != NULL), every match occupies its keypoint, there is no stereo gate and the
 *    distance gate is ORBdist.  queries[i].angle = pKF->mvKeysUn[i].angle; `blocks` is ignored (always 1). */
int pslfe_orb_search_by_projection_kf(pslfe_frame* cur, int slot, const PslProjQuery* queries, const uint8_t* qdesc, int nq,
                                      const uint8_t* taken, int orb_dist, int check_orientation, int32_t* match,
                                      int32_t* assigned, int* nmatches);

/* == ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) src/ORBmatcher.cc:159-288, from the point where the two DBoW2
 *    FeatureVectors are walked.  DBoW2 (vocabulary, FeatureVector) stays on the host; the caller passes
 *    fidx: the frame's FeatureVector flattened in node order (F.mFeatVec: for node ascending, its vIndicesF);
 *    one query per keyframe feature in the reference's iteration order (common nodes ascending, vIndicesKF order,
 *    NULL / bad map points dropped): the run [start, start+len) of fidx that is its node's vIndicesF, its descriptor
 *    (qdesc, 32 B) and pKF->mvKeysUn[realIdxKF].angle.
 *    Slot `slot` of `f` holds the frame's keypoints (mvKeys angles) and descriptors.  TH_LOW = 50, ratio test with
 *    mfNNratio against the second best (256 when there is none), a frame feature matched by an earlier query is skipped
 *    (:206-207), rotation histogram as elsewhere.  match[i] = frame feature of query i or -1; assigned[f] = query whose
 *    map point ends up in vpMapPointMatches[f]; *nmatches = the return value. */
typedef struct PslBowQuery {
    int32_t start, len;
    float angle;
} PslBowQuery;
int pslfe_orb_search_by_bow(pslfe_frame* f, int slot, const int32_t* fidx, int nfidx, const PslBowQuery* queries,
                            const uint8_t* qdesc, int nq, float nnratio, int check_orientation, int32_t* match,
                            int32_t* assigned, int* nmatches);

/* Batched, HBM-resident form of pslfe_orb_search_by_projection_last: pair p searches slot
 * slot0 + p with d_nq[p] queries at d_queries + p*qstride (descriptors at d_qdesc + p*qstride*32),
 * writes d_match + p*qstride and d_nmatches[p].  Asynchronous on the context's stream. */
int pslfe_orb_search_by_projection_last_device(pslfe_frame* cur, int slot0, int npairs, const PslProjQuery* d_queries,
                                               const uint8_t* d_qdesc, const int32_t* d_nq, int qstride,
                                               int check_orientation, int32_t* d_match, int32_t* d_nmatches);

/* == LSDmatcher::matchNNR add_src/LSDmatcher.cpp:354-376 (and LSDmatcher::match :378-413, whose
 *    live branch is matchNNR): matches12[i] = best train row if d0 < d1 * nnr (float compare on
 *    DMatch.distance) else -1; *nmatches = return value.  n2 < 2 is UB in the reference
 *    (:369); defined here as "no match". */
int pslfe_line_match_nnr(pslfe_ctx* ctx, const uint8_t* desc1, int n1, const uint8_t* desc2, int n2, float nnr,
                         int32_t* matches12, int* nmatches);

/* == LSDmatcher::SearchByGeomNApearance(CurrentFrame, LastFrame, desc_th) add_src/LSDmatcher.cpp:36-110
 *    (TrackWithMotionModel, src/Tracking.cc:1183): matchNNR, then for last-frame lines that own a map
 *    line (has_mapline[i1] != 0) the 20-degree direction gate and the 10 %-of-image end-point gate.
 *    matches12[n1] as the reference leaves it; assigned[n2] = last-frame line whose map line is given to
 *    current line i2 (or -1); *lmatches = return value. */
int pslfe_line_search_by_geom_appearance(pslfe_ctx* ctx, const PslKeyLine* kl_last, const uint8_t* desc_last, int n1,
                                         const PslKeyLine* kl_cur, const uint8_t* desc_cur, int n2, const uint8_t* has_mapline,
                                         float desc_th, float min_x, float max_x, float min_y, float max_y, int32_t* matches12,
                                         int32_t* assigned, int* lmatches);
/* == LSDmatcher::FrameBFMatch(ldesc1, ldesc2, LineMatches, TH) add_src/LSDmatcher.cpp:492-516 with
 *    lineDescriptorMAD :660-685: kNN-2, gap d1-d0 above half its MAD, d0 < TH, d0 < mfNNratio*d1. */
int pslfe_line_frame_bf_match(pslfe_ctx* ctx, const uint8_t* desc1, int n1, const uint8_t* desc2, int n2, float nnratio, float TH,
                              int32_t* line_matches);
/* == Map::AssociatePlanesByBoundary(pF, dTh, aTh) src/Map.cc:204-272 (live != 0; called from
 *    src/Tracking.cc:967,1209,1329,1531) / InsectLineMatch::SearchMapInsectline
 *    add_src/InsectlineMatch.cpp:9-59 (live == 0; no live caller upstream).  planes: n x 4 world planes of
 *    the frame's LIL pairs (ComputeWorldPlane); points: n x 5 x 3 doubles = start/end of line i, start/end
 *    of line j, 3-D intersection; map_planes: m x 4 in the caller's iteration order (upstream iterates a
 *    std::set of pointers); map_bad: isBad() flags (dead variant only, may be NULL).  The live variant
 *    keeps upstream's running threshold (dTh = dis, shared by all planes) and counts every association. */
int pslfe_associate_planes(pslfe_ctx* ctx, const float* planes, const double* points, int nplanes, const float* map_planes,
                           const uint8_t* map_bad, int nmap, float dTh, float aTh, int live, int32_t* assoc, int* nmatches);

/* One projected map line of LSDmatcher::SearchByProjection: what Tracking knows after isInFrustum. */
typedef struct PslLineQuery {
    float x1, y1, x2, y2;   /* mTrackProjX1, mTrackProjY1, mTrackProjX2, mTrackProjY2                     */
    float radius;           /* th (add_src/LSDmatcher.cpp:151) or RadiusByViewingCos * th (:282-285)       */
    float th_cos;           /* TH of GetFeaturesInAreaForLine: 0.96 (:155) or the default 0.998           */
    float vx, vy;           /* mode 0: LastFrame line direction ePointInOctave - sPointInOctave (:181-183) */
    float length;           /* mode 0: LastFrame.mvKeylinesUn[i].lineLength (:192-196)                     */
    int32_t blocks;         /* the map line has Observations() > 0                                         */
    double wdir[3];         /* mode 1: MapLine::GetNormal() (:293)                                         */
} PslLineQuery;

/* == LSDmatcher::SearchByProjection(CurrentFrame, LastFrame, th) add_src/LSDmatcher.cpp:112-215 (mode 0)
 *    and LSDmatcher::SearchByProjection(F, vpMapLines, eval_orient, th) :260-352 (mode 1), after the host
 *    has projected the map lines; includes Frame::AssignFeaturesToGridForLine (src/Frame.cc:286-309, with
 *    the Bresenham iterator of add_src/lineIterator.cpp) and Frame::GetFeaturesInAreaForLine (:752-826).
 *    kls/desc/lineEq: the frame's mvKeylinesUn, mLdesc, mvKeyLineFunctions; dir3d (mode 1): n x 3 doubles
 *    mvLines3D[i].first - .second.  taken (may be NULL), match, assigned, *nmatches as for the ORB
 *    matchers.  grid_start (CELLS+1) / grid_idx / grid_n (may be NULL): tap of mGridForLine as CSR with
 *    cell = ix*48+iy, for parity tests. */
int pslfe_line_search_by_projection(pslfe_ctx* ctx, const PslKeyLine* kls, const uint8_t* desc, const double* lineEq,
                                    const double* dir3d, int n, float min_x, float min_y, float max_x, float max_y,
                                    const PslLineQuery* queries, const uint8_t* qdesc, int nq, const uint8_t* taken, int mode,
                                    float nnratio, int32_t* match, int32_t* assigned, int* nmatches, int32_t* grid_start,
                                    int32_t* grid_idx, int grid_cap, int* grid_n);

/* ---- Frame::ComputeBoW ------------------------------------------------------------- */
typedef struct pslfe_vocab pslfe_vocab;
/* The DBoW2 vocabulary (ORBvoc.txt: k = 10, L = 6, TF_IDF weighting, L1_NORM scoring) as flat arrays, uploaded once:
 * node i has children child_ids[child_begin[i] .. + child_count[i]) in the order of Node::children
 * (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:297-329), a 32-byte descriptor, its weight and word id; a node without
 * children is a leaf (a word); node 0 is the root; L = depth of the tree (m_L). */
int pslfe_vocab_create(pslfe_ctx* ctx, int nnodes, const int32_t* child_begin, const int32_t* child_count,
                       const int32_t* child_ids, int nchild, const uint8_t* node_desc, const double* node_weight,
                       const int32_t* node_word, int L, pslfe_vocab** out);
void pslfe_vocab_destroy(pslfe_vocab* v);
/* == Frame::ComputeBoW src/Frame.cc:1053-1060 -> mpORBvocabulary->transform(descriptors, mBowVec, mFeatVec, levelsup = 4)
 *    (TemplatedVocabulary.h:1124-1195, 1218-1260).  desc: n x 32.  Per feature (any pointer may be NULL): word id,
 *    weight (features with weight <= 0 are dropped, as the reference drops stopped words) and the node at level
 *    L - levelsup.  mBowVec: nbow ascending (bow_id, bow_val) pairs, L1-normalised.  mFeatVec: nfv ascending node ids
 *    fv_node, node g owning fv_idx[fv_start[g] .. fv_start[g+1]) (feature indices ascending) - exactly the `fidx` /
 *    runs that pslfe_orb_search_by_bow takes. */
int pslfe_compute_bow(pslfe_vocab* v, const uint8_t* desc, int n, int levelsup, int32_t* f_word, double* f_weight,
                      int32_t* f_nid, int32_t* bow_id, double* bow_val, int* nbow, int32_t* fv_node, int32_t* fv_start,
                      int32_t* fv_idx, int* nfv);
/* Same for nframes frames in HBM: descriptors [nframes][stride][32], counts [nframes]; outputs [nframes][stride]
 * (bow_start / fv_start: [nframes][stride + 1]; bow_start is scratch), counts [nframes].  Asynchronous. */
int pslfe_compute_bow_device(pslfe_vocab* v, const uint8_t* d_desc, const int32_t* d_counts, int nframes, int stride,
                             int levelsup, int32_t* d_fword, double* d_fweight, int32_t* d_fnid, int32_t* d_bow_id,
                             double* d_bow_val, int32_t* d_bow_start, int32_t* d_nbow, int32_t* d_fv_node,
                             int32_t* d_fv_start, int32_t* d_fv_idx, int32_t* d_nfv);

/* ---- KeyFrame-rate matchers of LocalMapping / LoopClosing ------------------------------ */
/* These run on other threads than Tracking in the reference (src/LocalMapping.cc:336, 580, 796-872,
 * src/LoopClosing.cc:599, 245-330): give them their own pslfe_ctx (own stream) and one pslfe_kf handle per thread.  A frame
 * slot they read must be complete (its owner's stream synchronised) when it belongs to another context.  All entry points
 * take host pointers and return after the results have arrived. */
typedef struct pslfe_kf pslfe_kf;
int pslfe_kf_create(pslfe_ctx* ctx, pslfe_kf** out);
void pslfe_kf_destroy(pslfe_kf* k);

/* The candidate loop shared by ORBmatcher::Fuse(pKF, vpMapPoints, th) src/ORBmatcher.cc:825-966 (chi2 = 1: the
 * reprojection gates :907-934, 7.8 with a right coordinate and 5.99 without), ORBmatcher::Fuse(pKF, Scw, vpPoints, th,
 * vpReplacePoint) :968-1100 (chi2 = 0) and one direction of SearchBySim3 :1165-1232: KeyFrame::GetFeaturesInArea(u, v,
 * radius) src/KeyFrame.cc:685-724 on the grid of slot `slot`, octaves max_level-1 .. max_level (max_level =
 * nPredictedLevel; min_level, angle, blocks are ignored), the smallest DescriptorDistance, first visited on ties.
 * The host projects (it owns the map points); a query with radius < 0 is one the reference dropped before the search.
 * best_idx[i] = keypoint or -1, best_dist[i] = its distance or INT_MAX: the caller applies bestDist <= TH_LOW and
 * mutates the map (:950-964).  inv_level_sigma2: pKF->mvInvLevelSigma2 (nlevels <= 16 entries; chi2 = 1 only). */
int pslfe_kf_window_best(pslfe_kf* k, pslfe_frame* f, int slot, const PslProjQuery* queries, const uint8_t* qdesc, int nq,
                         int chi2, const float* inv_level_sigma2, int nlevels, int32_t* best_idx, int32_t* best_dist);
/* == ORBmatcher::SearchBySim3 src/ORBmatcher.cc:1102-1326 after the projections: q12[i1] = map point i1 of KF1 in KF2's
 *    image (radius < 0: none / already matched / bad / a gate failed), qdesc1 its descriptor, n1 = N1; q21 / qdesc2 / n2
 *    likewise.  Both directions, TH_HIGH, then the agreement check :1307-1323: match12[i1] = idx2 or -1. */
int pslfe_kf_search_by_sim3(pslfe_kf* k, pslfe_frame* f1, int slot1, pslfe_frame* f2, int slot2, const PslProjQuery* q12,
                            const uint8_t* qdesc1, int n1, const PslProjQuery* q21, const uint8_t* qdesc2, int n2,
                            int32_t* match12, int* nfound);
/* One feature of KF1 in ORBmatcher::SearchForTriangulation, in the reference's iteration order (common vocabulary nodes
 * ascending, f1it->second order; features that have a map point and, under bOnlyStereo, those without a right coordinate
 * are dropped by the caller, :699-711). */
typedef struct PslTriQuery {
    int32_t start, len; /* the run of KF2's flattened FeatureVector under the shared node */
    float x, y, angle;  /* pKF1->mvKeysUn[idx1].pt, .angle                                  */
    int32_t stereo;     /* pKF1->mvuRight[idx1] >= 0                                       */
} PslTriQuery;
/* == ORBmatcher::SearchForTriangulation src/ORBmatcher.cc:657-823 with CheckDistEpipolarLine :140-157.  Slot `slot2` of f2
 *    holds KF2 (mvKeysUn, mvuRight, descriptors); fidx2: its FeatureVector flattened in node order; taken2[idx2] != 0 <=>
 *    pKF2->GetMapPoint(idx2) != NULL; F12 row-major 3x3; (ex, ey) the epipole :664-671; scale_factors / level_sigma2 =
 *    pKF2->mvScaleFactors / mvLevelSigma2.  TH_LOW = 50, the LAST candidate wins among equal distances (:738), vbMatched2
 *    is never set by the reference (:686) so queries are independent; rotation histogram :764-775, :793-811.
 *    match[i] = idx2 of query i or -1; *nmatches = the return value. */
int pslfe_kf_search_for_triangulation(pslfe_kf* k, pslfe_frame* f2, int slot2, const int32_t* fidx2, int nfidx2,
                                      const uint8_t* taken2, const PslTriQuery* queries, const uint8_t* qdesc, int nq,
                                      const float* F12, float ex, float ey, int only_stereo, int check_orientation,
                                      const float* scale_factors, const float* level_sigma2, int nlevels, int32_t* match,
                                      int* nmatches);
/* One projected map line of LSDmatcher::Fuse (add_src/LSDmatcher.cpp:885-931). */
typedef struct PslLineFuseQuery {
    float x1, y1, x2, y2; /* u1, v1, u2, v2                                 */
    float radius;         /* th * mvScaleFactorsLine[level]; < 0: dropped   */
    int32_t level;        /* nPredictedLevel                                */
} PslLineFuseQuery;
/* == the search of LSDmatcher::Fuse add_src/LSDmatcher.cpp:933-958: KeyFrame::GetLinesInArea(u1, v1, u2, v2, radius,
 *    TH = 0.998) src/KeyFrame.cc:857-891 over kls = pKF->mvKeyLines, octaves level-1 .. level, smallest distance to
 *    desc row idx, first on ties; best_dist = 256 when none.  `desc` (ndesc rows) is the matrix the caller reads rows
 *    from: the reference indexes pKF->mDescriptors (the ORB matrix) with the line index at :945; a line without a row is
 *    skipped. */
int pslfe_kf_line_fuse_best(pslfe_kf* k, const PslKeyLine* kls, int n, const uint8_t* desc, int ndesc,
                            const PslLineFuseQuery* queries, const uint8_t* qdesc, int nq, int32_t* best_idx,
                            int32_t* best_dist);
/* == MapPoint::ComputeDistinctiveDescriptors src/MapPoint.cc:242-304 and MapLine::ComputeDistinctiveDescriptors
 *    add_src/MapLine.cpp:250-310 for npts map points / lines at once: the observed descriptors of point p are rows
 *    offsets[p] .. offsets[p+1] of desc (at most 1024 per point); best[p] = the row (relative to offsets[p]) with the
 *    least median Hamming distance to the others (median = sorted[0.5*(N-1)], first on ties), -1 for an empty run. */
int pslfe_kf_distinctive_descriptors(pslfe_kf* k, const uint8_t* desc, const int32_t* offsets, int npts, int32_t* best);

/* ---- RGB-D line glue of the Frame constructor ------------------------------ */
typedef struct pslfe_glue pslfe_glue;
/* Buffers for up to max_batch frames of max_lines keylines and max_fans LIL rows each. */
int pslfe_glue_create(pslfe_ctx* ctx, int max_lines, int max_fans, int max_batch, pslfe_glue** out);
void pslfe_glue_destroy(pslfe_glue* g);
/* == the part of Frame::ExtractLSD after the extractor (src/Frame.cc:490-660) for one frame, host pointers:
 *    Frame::isLineGood(im, imDepth, K) :662-750 with LINEextractor::compPt3dCov / extract3dline_mahdist
 *    (add_src/LineExtractor.cpp:40-322): per keyline <= 21 depth samples, back-projection, RANSAC 3-D line by
 *    Mahalanobis distance -> mvLines3D, mvLineEq;
 *    Frame::convertFansToKeyLines(fans, mvKeylinesUn) :426-472 with Frame_shortestDistance :381-424 ->
 *    intersection_lines_plane; the plane loop :505-660 with Frame::OldPlane :474-488 -> mvPlanes, mvPlaneNormal,
 *    mvPlaneLineNo, CrossPoint_3D, CrossPoint_2D, mvle_l.
 *    kls: mvKeylinesUn (n); fans: the n x 4 matrix of CPartiallyRecoverConnectivity (x, y, index1, index2);
 *    depth: CV_32F image (metres), stride in floats; cam: fx, fy, cx, cy are used.
 *    rand() is glibc's generator seeded as srand(seed) at the start of the frame (the reference never seeds: its
 *    stream position depends on the process history; convention H7).  Asynchronous; results via pslfe_glue_fetch. */
int pslfe_glue_run(pslfe_glue* g, const PslKeyLine* kls, int nlines, const float* fans, int nfans, const float* depth,
                   int width, int height, int depth_stride, const PslCamera* cam, uint32_t seed);
/* Same for nframes frames resident in HBM: keylines [nframes][kl_stride] (kl_stride == max_lines), counts
 * [nframes], fans [nframes][fan_stride][4], counts [nframes], depth [nframes][height][width] float; frame f is
 * seeded with seed0 + f.  (pslfe_line_results_device / pslfe_line_pair_batch_device give exactly these views.) */
int pslfe_glue_run_batch_device(pslfe_glue* g, int nframes, const PslKeyLine* d_kls, int kl_stride, const int32_t* d_nkl,
                                const float* d_fans, int fan_stride, const int32_t* d_nfans, const float* d_depth,
                                int width, int height, const PslCamera* cam, uint32_t seed0);
/* Results of frame `frame` (any pointer may be NULL):
 *   lines3d [nlines][6] f64 = mvLines3D (start, end; zeros when the line failed), lineEq [nlines][3] = mvLineEq
 *   (-1,-1,-1 when failed); crossings (intersection_lines_plane, fan order): pair [k][2], xy [k][2], cross [k][3]
 *   f64, le_l [k][6] f64 = mvle_l; planes: planes [p][4] = mvPlanes, normals [p][3] f64 = mvPlaneNormal, lineNo
 *   [p][2] = mvPlaneLineNo, cross3d [p][3] = CrossPoint_3D, cross2d [p][2] f64 = CrossPoint_2D. */
int pslfe_glue_fetch(pslfe_glue* g, int frame, int nlines, double* lines3d, float* lineEq, int32_t* pair, float* xy,
                     double* cross, double* le_l, int int_cap, int* nint, float* planes, double* normals, int32_t* lineNo,
                     double* cross3d, double* cross2d, int plane_cap, int* nplanes);

/* ---- batched many-frames mode across the GPUs of one node (BASELINE configs[3]) -----------------------
 * The reference has no counterpart (it is single-process, single-camera: src/System.cc:91-101); north_star asks for
 * independent frames / streams sharded over the 8 GPUs with RCCL over xGMI for the result gather.  Stream s -> rank
 * s mod world, no data-path collective; the one exchange is a gather (to the consuming rank, or to all) of fixed-size per-frame
 * RESULT RECORDS holding what Tracking.cc reads of a Frame on this path.  Record (little endian, sections 16-byte aligned, zero padded):
 *   header  8 x int32: n_kp, n_match, n_kl, n_lmatch, n_fan, n_planes (true counts), flags (bit0 kps / bit1 lines / bit2 fans /
 *           bit3 planes truncated to the capacity), frame index inside the rank's batch
 *   kps     [kp_cap] PslKeyPoint    = mvKeys               desc   [kp_cap][32] = mDescriptors
 *   match   [kp_cap] int32          = ORBmatcher::SearchByProjection result (query i -> keypoint of this frame, -1 none; all kp_cap rows of the
 *                                     caller's buffer, which holds -1 beyond the query count; rows beyond match_stride read -1)
 *   kls     [kl_cap] PslKeyLine     = mvKeylinesUn         ldesc  [kl_cap][32] = mLdesc
 *   lineEq  [kl_cap][3] f64         = mvKeyLineFunctions   lmatch [kl_cap] int32 = LSDmatcher::match result
 *   fans    [fan_cap][4] f32        = CPartiallyRecoverConnectivity rows (x, y, i, j)
 *   planes  [plane_cap][4] f32      = mvPlanes             plane_lines [plane_cap][2] int32 = mvPlaneLineNo          */
typedef struct PslRecordCaps { int32_t kp_cap, kl_cap, fan_cap, plane_cap; } PslRecordCaps;
typedef struct PslRecordLayout {
    int64_t bytes;  /* size of one record, a multiple of 256 */
    int64_t off_kps, off_desc, off_match, off_kls, off_ldesc, off_lineEq, off_lmatch, off_fans, off_planes, off_plane_lines;
} PslRecordLayout;
/* Pure host arithmetic (no GPU needed): the offsets every consumer of a record uses. */
int pslfe_record_layout(const PslRecordCaps* caps, PslRecordLayout* out);
/* Where the results of a batch live in HBM (the *_results_device / *_fans_device views and the caller's match buffers);
 * a NULL pointer leaves its section empty.  Strides are in rows per frame. */
typedef struct PslRecordSources {
    const PslKeyPoint* d_kps; const uint8_t* d_desc; const int32_t* d_kp_counts; int32_t kp_stride;
    const int32_t* d_match; const int32_t* d_nmatches; int32_t match_stride;
    const PslKeyLine* d_kls; const uint8_t* d_ldesc; const double* d_lineEq; const int32_t* d_kl_counts; int32_t kl_stride;
    const int32_t* d_lmatch; const int32_t* d_nlmatches; int32_t lmatch_stride;
    const float* d_fans; const int32_t* d_fan_counts; int32_t fan_stride;
    const float* d_planes; const int32_t* d_plane_lines; const int32_t* d_plane_counts; int32_t plane_stride;
} PslRecordSources;
/* Packs the records of nframes frames into d_records ([nframes][layout.bytes]), asynchronously on the context's stream. */
int pslfe_record_pack_device(pslfe_ctx* ctx, const PslRecordCaps* caps, const PslRecordSources* src, int nframes, void* d_records);
/* mvPlanes / mvPlaneLineNo / their counts of the last pslfe_glue_run_batch_device, HBM resident ([nframes][plane_stride][..]). */
int pslfe_glue_planes_device(pslfe_glue* g, const float** d_planes, const int32_t** d_plane_lines, const int32_t** d_plane_counts,
                             int* plane_stride);

/* RCCL gather of the records, one communicator per context.  RCCL is loaded at run time (librccl.so.1).
 *   pslfe_gather_unique_id  rank 0 obtains the 128-byte ncclUniqueId and hands it to the other ranks by whatever channel the
 *                           host has (MPI, a socket, torch.distributed, a file);
 *   pslfe_gather_create     ncclCommInitRank - collective: every rank calls it with the same id;
 *   pslfe_gather_all        d_recv[world][bytes_per_rank] <- every rank's d_send[bytes_per_rank]; runs on the gather's own
 *                           stream AFTER everything issued on the context's stream so far, so the next batch's kernels
 *                           overlap it; one exchange may be in flight;
 *   pslfe_gather_to_root    the same towards ONE consuming rank (SURVEY.md §8e "ncclGather-by-send/recv"): rank `root` receives
 *                           d_recv[world][bytes_per_rank] (its own part included), the other ranks only send and need no receive
 *                           buffer (d_recv may be NULL there) - one group of ncclSend / ncclRecv; collective: every rank of the
 *                           communicator calls it with the same root and bytes_per_rank;
 *   pslfe_gather_wait       host_blocking != 0: the host waits for the exchange; 0: the context's stream waits for it (the host
 *                           may read the received records only after a host-blocking wait). */
typedef struct pslfe_gather pslfe_gather;
int pslfe_gather_unique_id(uint8_t id[128]);
int pslfe_gather_create(pslfe_ctx* ctx, int rank, int world, const uint8_t id[128], pslfe_gather** out);
void pslfe_gather_destroy(pslfe_gather* g);
int pslfe_gather_all(pslfe_gather* g, const void* d_send, size_t bytes_per_rank, void* d_recv);
int pslfe_gather_to_root(pslfe_gather* g, const void* d_send, size_t bytes_per_rank, int root, void* d_recv);
int pslfe_gather_wait(pslfe_gather* g, int host_blocking);
int pslfe_gather_world(const pslfe_gather* g, int* rank, int* world);

#ifdef __cplusplus
}
#endif
#endif
