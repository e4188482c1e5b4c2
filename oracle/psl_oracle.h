// ORACLE C interface (test infrastructure only; see header of orb_oracle.cpp).
#ifndef PSL_ORACLE_H
#define PSL_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

// == cv::KeyPoint (28 B): SURVEY.md Appendix B
typedef struct PsoKeyPoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} PsoKeyPoint;

// == line_descriptor::KeyLine (68 B), field order of
// Thirdparty/line_descriptor/include/line_descriptor/descriptor_custom.hpp:107-146
typedef struct PsoKeyLine {
    float angle;
    int32_t class_id, octave;
    float pt_x, pt_y, response, size;
    float startPointX, startPointY, endPointX, endPointY;
    float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
    float lineLength;
    int32_t numOfPixels;
} PsoKeyLine;

typedef struct PsoProjQuery {
    float u, v, radius, ur;
    int32_t min_level, max_level;
    float angle;
    int32_t blocks;
} PsoProjQuery;

/* one feature of KF1 in ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:693-775) */
typedef struct PsoTriQuery {
    int32_t start, len;  /* the run of KF2's flattened FeatureVector under the shared node */
    float x, y, angle;   /* pKF1->mvKeysUn[idx1] */
    int32_t stereo;      /* pKF1->mvuRight[idx1] >= 0 */
} PsoTriQuery;

/* one projected map line of LSDmatcher::Fuse (add_src/LSDmatcher.cpp:885-931) */
typedef struct PsoLineFuseQuery {
    float x1, y1, x2, y2, radius;  /* radius < 0: dropped by a gate before the search */
    int32_t level;                 /* nPredictedLevel */
} PsoLineFuseQuery;

typedef struct PsoLineQuery {
    float x1, y1, x2, y2;   /* mTrackProjX1.. : projected end points */
    float radius, th_cos;   /* r and TH of GetFeaturesInAreaForLine */
    float vx, vy, length;   /* mode 0: last frame's line direction (ePoint - sPoint) and lineLength */
    int32_t blocks;
    double wdir[3];         /* mode 1: MapLine::GetNormal() */
} PsoLineQuery;

void* pso_orb_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
void pso_orb_destroy(void* h);
int pso_orb_extract(void* h, const uint8_t* gray, int w, int hh, int stride, PsoKeyPoint* kps, uint8_t* desc, int cap);
int pso_orb_quota(void* h, int level);
float pso_orb_scale(void* h, int level);
int pso_orb_umax(void* h, int v);
int pso_orb_level_size(void* h, int level, int* w, int* hh);
const uint8_t* pso_orb_level_ptr(void* h, int level);
const uint8_t* pso_orb_blur_ptr(void* h, int level);
int pso_orb_level_candidates(void* h, int level, int* xys, int cap);
int pso_orb_level_keypoints(void* h, int level, PsoKeyPoint* out, int cap);

int pso_distribute_octree(const int* xys, int n, int minX, int maxX, int minY, int maxY, int N, int* out_xys, int cap);
void pso_resize_linear_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh);
void pso_gaussian_blur_u8(const uint8_t* src, int w, int h, int ksize, double sigma, uint8_t* dst);
void pso_gaussian_kernel_q8(int ksize, double sigma, int* K);
int pso_fast_subimage(const uint8_t* img, int w, int h, int x0, int y0, int x1, int y1, int threshold, int* xys, int cap);
float pso_fast_atan2_f(float y, float x);
float pso_sinf_f(float x);
float pso_cosf_f(float x);
float pso_libm_sinf(float x);
float pso_libm_cosf(float x);
int pso_cvround_d(double v);
const int8_t* pso_orb_pattern(void);

int pso_hamming256(const uint8_t* a, const uint8_t* b);
int pso_grid_build(const PsoKeyPoint* kps, int n, float minX, float minY, float maxX, float maxY, int* start, int* idx);
int pso_search_by_projection_last(const PsoKeyPoint* kps, const uint8_t* desc, const float* uright, int n, float minX,
                                  float minY, float maxX, float maxY, const PsoProjQuery* q, const uint8_t* qdesc, int nq,
                                  const uint8_t* taken, int checkOri, int* match, int* assigned);
int pso_search_by_projection_map(const PsoKeyPoint* kps, const uint8_t* desc, const float* uright, int n, float minX,
                                 float minY, float maxX, float maxY, const PsoProjQuery* q, const uint8_t* qdesc, int nq,
                                 const uint8_t* taken, float nnratio, int* match, int* assigned);
void pso_hamming_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int* idx, int* dist);
void pso_window_best(const PsoKeyPoint* kps, const uint8_t* desc, const float* uright, int n, const float* bounds, const PsoProjQuery* q,
                     const uint8_t* qdesc, int nq, int chi2, const float* invSigma2, int* best_idx, int* best_dist);
int pso_search_by_sim3(const PsoKeyPoint* kps1, const uint8_t* desc1, int n1, const float* bounds1, const PsoKeyPoint* kps2,
                       const uint8_t* desc2, int n2, const float* bounds2, const PsoProjQuery* q12, const uint8_t* qdesc1,
                       const PsoProjQuery* q21, const uint8_t* qdesc2, int* match12);
int pso_search_for_triangulation(const PsoKeyPoint* kps2, const uint8_t* desc2, const float* uright2, const uint8_t* taken2,
                                 const int32_t* fidx2, const PsoTriQuery* q, const uint8_t* qdesc, int nq, const float* F12, float ex,
                                 float ey, int bOnlyStereo, int checkOri, const float* scaleFactors, const float* levelSigma2, int* match);
void pso_line_fuse_best(const PsoKeyLine* kls, int n, const uint8_t* desc, int ndesc, const PsoLineFuseQuery* q, const uint8_t* qdesc, int nq,
                        int* best_idx, int* best_dist);
void pso_distinctive_descriptors(const uint8_t* desc, const int32_t* offsets, int npts, int* best);
int pso_line_match_nnr(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnr, int* matches12);

int pso_set_lsd_refine(int mode);   /* 1 = LSD_REFINE_STD, 2 = LSD_REFINE_ADV (default) */
int pso_set_nfa_math(int restated); /* 0 = host libm (default), 1 = psl_f64math.h */
double pso_lsd_nfa(int n, int k, double p, int W, int H);
int pso_lsd_rects(const uint8_t* gray, int w, int h, int stride, double* rects, int cap);
int pso_lsd_detect(const uint8_t* gray, int w, int h, int stride, float* lines, int cap);
int pso_lsd_gradient(const uint8_t* gray, int w, int h, int stride, double* scaled, double* angles, double* modgrad, int* W, int* H);
int pso_merge_lines(const float* src, int n, float ang, float dist, float ep, float* dst, int cap);
int pso_optimize_and_merge(const float* src, int n, int w, int h, PsoKeyLine* out, int cap);
int pso_line_iterator_count(int w, int h, float x1, float y1, float x2, float y2);
int pso_line_iterator_walk(double x1, double y1, double x2, double y2, int* xy, int cap);
int pso_lbd_compute(const uint8_t* gray, int w, int h, int stride, const PsoKeyLine* kls, int n, uint8_t* desc, float* fdesc);
int pso_lbd_sobel(const uint8_t* gray, int w, int h, int stride, short* dx, short* dy);
int pso_line_extract(const uint8_t* gray, int w, int h, int stride, int nLSDFeature, PsoKeyLine* kls, uint8_t* desc,
                     double* lineEq, int cap);
int pso_lil_pair(const float* L, int rows, float radius, float fanThr, int imgCols, int imgRows, float* fans, int cap);

int pso_search_by_geom_appearance(const PsoKeyLine* kl_last, const uint8_t* d_last, int n1, const PsoKeyLine* kl_cur, const uint8_t* d_cur,
                                  int n2, const uint8_t* has_mapline, float desc_th, float minX, float maxX, float minY, float maxY,
                                  int* matches12, int* assigned);
void pso_frame_bf_match(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnratio, float TH, int* lineMatches);
int pso_associate_planes(const float* planes, const double* pts, int N, const float* map, const uint8_t* bad, int M, float dTh, float aTh,
                         int live, int* assoc);

int pso_line_grid_build(const PsoKeyLine* k, int n, float minX, float minY, float maxX, float maxY, int* start, int* idx, int cap);
int pso_line_search_by_projection(const PsoKeyLine* k, const uint8_t* desc, const double* eq, const double* dir3d, int n, float minX,
                                  float minY, float maxX, float maxY, const PsoLineQuery* q, const uint8_t* qdesc, int nq,
                                  const uint8_t* taken, int mode, float nnratio, int* match, int* assigned);

void pso_rgb_to_gray(const uint8_t* rgb, int w, int h, int stride, int is_rgb, uint8_t* gray);
void pso_depth_to_float(const uint16_t* d, int n, float factor, float* out);
void pso_image_bounds(int cols, int rows, const float* K, const float* dist, float* bounds);
void pso_frame_post_rgbd(const PsoKeyPoint* kps, int n, const float* depth, int w, int h, int dstride, const float* K, const float* dist,
                         float mbf, PsoKeyPoint* kpsUn, float* mvDepth, float* mvuRight);

int pso_search_by_projection_kf(const PsoKeyPoint* kps, const uint8_t* desc, int n, float minX, float minY, float maxX, float maxY,
                                const PsoProjQuery* q, const uint8_t* qdesc, int nq, const uint8_t* taken, int orbDist, int checkOri,
                                int* match, int* assigned);
int pso_search_by_bow(const uint8_t* fdesc, const float* fangle, int nf, const int32_t* fidx, const int32_t* run, const uint8_t* qdesc,
                      const float* qangle, int nq, float nnratio, int checkOri, int* match, int* assigned);

/* Frame::ComputeBoW = DBoW2 transform on a flat vocabulary (bow_oracle.cpp) */
int pso_compute_bow(const int32_t* child_begin, const int32_t* child_count, const int32_t* child_ids, const uint8_t* node_desc,
                    const double* node_weight, const int32_t* node_word, int L, int levelsup, const uint8_t* desc, int n, int32_t* f_word,
                    double* f_weight, int32_t* f_nid, int32_t* bow_id, double* bow_val, int32_t* fv_node, int32_t* fv_start, int32_t* fv_idx,
                    int* n_fv);

/* RGB-D line glue of the Frame constructor (glue_oracle.cpp) */
void pso_line_good(const PsoKeyLine* kls, int n, const float* depth, int cols, int rows, int dstride, const float* cam, uint32_t seed,
                   double* lines3d, float* lineEq);
int pso_fans_to_intersections(const float* fans, int nfans, const double* lines3d, int32_t* pair, float* xy, double* cross, int cap);
int pso_planes_from_pairs(const PsoKeyLine* kls, const float* lineEq, const double* lines3d, const int32_t* pair, const float* xy,
                          const double* cross, int nint, float* planes, double* normals, int32_t* lineNo, double* cross3d, double* cross2d,
                          double* le_l, int cap);
int pso_glibc_rand(uint32_t seed, int n, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif
