/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See the header of each oracle .c file.
 * PARITY UNPINNED (no golden vectors exist in the reference; saiga submodule absent).
 */
#ifndef SNK_ORACLE_H
#define SNK_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_DIST_INF 256

typedef struct orc_knn2
{
    int32_t idx1, dist1, idx2, dist2;
} orc_knn2;

typedef struct orc_kp64
{
    double x, y;
    float angle;
    int32_t octave;
} orc_kp64;

/* ---- match_oracle.c ---- */
int orc_hamming(const uint64_t a[4], const uint64_t b[4]);
void orc_bf_knn2(const uint64_t (*q)[4], int nq, const uint64_t (*t)[4], int nt, orc_knn2* out, int threads);
int orc_bf_filter(const orc_knn2* knn, int nq, int threshold, float ratio, int32_t (*pairs)[2]);
int orc_stereo_match(const orc_kp64* left, const uint64_t (*dl)[4], int nl, const orc_kp64* right,
                     const uint64_t (*dr)[4], int nr, double bf, const float* level_scale, int relaxed,
                     float* right_points, float* depth);

/* ---- orb_oracle.c ---- */
#define ORC_MAX_LEVELS 16
#define ORC_LEVEL_CAP 8192
#define ORC_CELL_SLOTS 64

typedef struct orc_orb_params
{
    int32_t nfeatures;
    float scale_factor;
    int32_t n_levels;
    int32_t ini_th;
    int32_t min_th;
} orc_orb_params;

typedef struct orc_keypoint
{
    float x, y, size, angle, response;
    int32_t octave;
} orc_keypoint;

typedef struct orc_orb_layout_t
{
    int32_t n_levels;
    float scale[ORC_MAX_LEVELS];
    int32_t w[ORC_MAX_LEVELS], h[ORC_MAX_LEVELS], nfeat[ORC_MAX_LEVELS];
} orc_orb_layout_t;

typedef struct orc_cell_grid_t
{
    int32_t n_cols, n_rows, w_cell, h_cell;
} orc_cell_grid_t;

typedef struct orc_cand
{
    uint16_t x, y, score, cell;
} orc_cand;

extern const int8_t orc_brief_pattern[1024];
int orc_orb_layout(const orc_orb_params* p, int w, int h, orc_orb_layout_t* L);
void orc_resize_coords(int src, int dst, int32_t* ofs, int32_t* w1);
void orc_resize(const uint8_t* src, int sw, int sh, int spitch, uint8_t* dst, int dw, int dh, int dpitch);
int orc_fast_score(const uint8_t* img, int pitch, int x, int y);
void orc_cell_grid(int w, int h, orc_cell_grid_t* g);
int orc_orb_candidates(const uint8_t* img, int w, int h, int pitch, int ini_th, int min_th, orc_cand* out, int cap);
uint64_t orc_point_key(int x, int y, int W, int H);
int orc_orb_distribute_bound(int w, int h, int N); /* capacity of out_idx below */
int orc_orb_distribute(const orc_cand* pts, int n, int w, int h, int N, int* out_idx);
int orc_orb_distribute_ranked(const orc_cand* pts, const uint32_t* rank, int n, int w, int h, int N, int* out_idx);
void orc_harris_abc(const uint8_t* img, int pitch, int x, int y, int32_t* a, int32_t* b, int32_t* c);
float orc_harris_response(const uint8_t* img, int pitch, int x, int y);
uint32_t orc_harris_rank(float r);
int orc_orb_set_response(int v); /* "orb.response": 0 = FAST score, 1 = Harris (set through orc_set_definition) */
void orc_umax(int* umax);
void orc_ic_moments(const uint8_t* img, int pitch, int x, int y, int* m10, int* m01);
float orc_fast_atan2(float y, float x);
void orc_sincos_deg(float deg, float* s_out, float* c_out);
int orc_blur_at(const uint8_t* img, int w, int h, int pitch, int x, int y);
void orc_blur_image(const uint8_t* img, int w, int h, int pitch, uint8_t* dst, int dpitch);
void orc_descriptor_blurred(const uint8_t* blurred, int pitch, int x, int y, float angle_deg, uint64_t out[4]);
void orc_descriptor(const uint8_t* img, int w, int h, int pitch, int x, int y, float angle_deg, uint64_t out[4]);
int orc_orb_pyramid(const orc_orb_params* p, const uint8_t* img, int w, int h, int pitch, uint8_t** levels,
                    orc_orb_layout_t* L);
int orc_orb_detect(const orc_orb_params* p, const uint8_t* img, int w, int h, int pitch, orc_keypoint* kps,
                   uint64_t (*desc)[4], int capacity, int level_cap, int threads);

/* ---- preprocess_oracle.c ---- */
typedef struct orc_rectification
{
    double K_src[4]; /* fx fy cx cy */
    double D_src[8]; /* k1 k2 k3 k4 k5 k6 p1 p2 */
    double R[9];     /* row-major */
    double K_dst[4];
    double bf;
} orc_rectification;
void orc_undistort_gn(const double* D, double px, double py, double* ox, double* oy);
void orc_rectify(const orc_rectification* R, const orc_keypoint* kps, int n, orc_kp64* out, double (*normalized)[2]);
int orc_rgbd_stereo(const orc_kp64* und, int n, const double* K, const double* D_depth, const double* K_depth, double bf,
                    const float* depth_image, int w, int h, int pitch_floats, float* right_points, float* depth);

/* ---- ba_oracle.c ---- */
typedef struct orc_ba_options
{
    int32_t max_iterations;     /* LM iterations (reference: 3) */
    int32_t max_pcg_iterations; /* reference: 30 */
    double pcg_tol;             /* reference: 1e-10 */
    double huber_mono, huber_stereo;
    double lambda_init; /* 0 -> 1e-4 */
} orc_ba_options;

typedef struct orc_ba_rpc /* Saiga RelPoseConstraint, LocalBundleAdjustment.cpp:294-346 */
{
    int32_t img1, img2;
    double rel_pose[7]; /* T_img2 * T_img1^-1 */
    double weight_rotation, weight_translation;
} orc_ba_rpc;

typedef struct orc_ba_problem
{
    int32_t n_img, n_pt, n_obs;
    double (*pose)[7]; /* qx qy qz qw tx ty tz, world -> camera; updated in place */
    const uint8_t* img_const;
    double (*pt)[3]; /* updated in place */
    const uint8_t* pt_const;
    const int32_t* obs_img;
    const int32_t* obs_pt;
    const double (*obs_uv)[2];
    const double* obs_depth; /* > 0 => stereo observation */
    const double* obs_weight;
    const uint8_t* obs_outlier; /* may be NULL */
    double K[4];                /* fx fy cx cy */
    double bf;
    int32_t n_rpc;
    int32_t pad;
    const orc_ba_rpc* rpc;
} orc_ba_problem;

void orc_se3_update(const double* pose, const double* d, double* out);
int orc_ba_rpc_linearize(const double* pose1, const double* pose2, const orc_ba_rpc* c, double* r, double* J1);
void orc_ba_chi2(const orc_ba_problem* P, double* chi2);
/* 0 = the restatement's summation order (default), 1 = reversed, 2 = pairwise: the oracle against a re-ordered copy of itself
 * (the control behind the truncated-PCG rule of tests/ba_parity.py); process-wide, test infrastructure */
void orc_ba_set_sum_order(int mode);
int orc_ba_solve(orc_ba_problem* P, const orc_ba_options* O, int iterations, double* cost_initial, double* cost_final,
                 int* pcg_iterations_total);

/* ---- track_oracle.c ---- */
typedef struct orc_grid_bounds
{
    double min_x, min_y, max_x, max_y;
} orc_grid_bounds;

typedef struct orc_frame_view
{
    int32_t n;
    int32_t cols, rows;
    const orc_kp64* kps;        /* undistorted keypoints, grid order */
    const uint64_t (*desc)[4];
    const float* right_points;
    const uint8_t* taken;       /* mvpMapPoints[i] != nullptr */
    const int32_t* cell_start;  /* cols*rows + 1, x-major cells */
    orc_grid_bounds bounds;
} orc_frame_view;

typedef struct orc_camera
{
    double fx, fy, cx, cy, bf;
} orc_camera;

typedef struct orc_lm_coarse
{
    double pos[3], normal[3];
    uint64_t desc[4];
    int32_t octave;
    float angle;
} orc_lm_coarse;

typedef struct orc_lm_fine
{
    double pos[3], normal[3];
    uint64_t desc[4];
    float reference_depth;
    int32_t reference_scale_level;
    uint8_t valid;
    uint8_t pad[7];
} orc_lm_fine;

double orc_det_log(double x);
double orc_det_exp(double y);
void orc_grid_dims(const orc_grid_bounds* b, int* cols, int* rows);
void orc_feature_grid(const orc_kp64* kps, int n, const orc_grid_bounds* b, int32_t* perm, int32_t* cell_start);
int orc_set_definition(const char* key, int value); /* mirror of snk_set_definition (0 = ok, 1 = unknown key / bad value) */
void orc_set_match_threads(int n); /* threads of the per-point phase of orc_match_coarse / orc_match_fine */
int orc_match_coarse(const orc_frame_view* f, const orc_camera* cam, const double* pose, const orc_lm_coarse* pts, int m,
                     float th, int feature_error, int direction, const float* level_scale, int n_levels, int32_t* match_idx);
int orc_match_fine(const orc_frame_view* f, const orc_camera* cam, const double* pose, orc_lm_fine* pts, int m, float th,
                   float ratio, const float* level_scale, int n_levels, int32_t* match_idx, uint8_t* visible);
int orc_match_keyframe(const orc_frame_view* f, const orc_camera* cam, const double* pose, const double (*pos)[3],
                       const uint64_t (*desc)[4], const uint8_t* skip, int m, float th, int feature_error, int32_t* match_idx);

typedef struct orc_fusion_point /* Snake/Map/LocalMap.h:57-80 */
{
    double pos[3], normal[3];
    uint64_t desc[4];
    float reference_depth;
    int32_t reference_scale_level;
    int32_t observations;
    int32_t id;
} orc_fusion_point;

int orc_match_fuse(const orc_frame_view* f, const orc_camera* cam, const double* pose, const orc_fusion_point* pts,
                   const uint8_t* point_mask, int m, float th, float obs_factor, int feature_th, const float* level_scale,
                   int n_levels, int32_t* best_idx);
int orc_match_triangulation_project(const double* depth_grid, int grid_rows, int grid_cols, const double* pose1,
                                    const double* pose2, const orc_camera* cam, const orc_kp64* kps1,
                                    const double (*np1)[2], const uint64_t (*desc1)[4], const uint8_t* has_mp1, int n1,
                                    const orc_frame_view* f2, const double (*np2)[2], const double* E12,
                                    float epipolar_distance, int feature_distance, int32_t* match_idx2);
typedef struct orc_relink_query /* one (keyframe feature, map point) observation — DeferredMapper.cpp:61-99 */
{
    double pos[3];        /* mp->getPosition() */
    uint64_t desc[4];     /* mp->descriptor */
    uint64_t alt_desc[4]; /* descriptor of the first observation of mp in another keyframe (:90-98) */
    int32_t feature;      /* i */
    int32_t has_alt;
} orc_relink_query;
int orc_match_relink(const orc_frame_view* f, const orc_camera* cam, const double* pose, const orc_relink_query* queries, int n,
                     float radius, double outlier_threshold, int feature_threshold, int32_t* action, int32_t* best_idx);
int orc_match_triangulation_bow(const orc_camera* cam, const double* E12, const double (*np1)[2], const uint64_t (*desc1)[4],
                                const uint8_t* has_mp1, int n_nodes1, const uint32_t* node_id1, const int32_t* node_start1,
                                const int32_t* feat1, const double (*np2)[2], const uint64_t (*desc2)[4], const uint8_t* has_mp2,
                                int n_nodes2, const uint32_t* node_id2, const int32_t* node_start2, const int32_t* feat2,
                                float epipolar_distance, int feature_distance, int32_t (*pairs)[2]);
int orc_match_triangulation_bf(const orc_camera* cam, const double* E12, const double (*np1)[2], const uint64_t (*desc1)[4],
                               const uint8_t* has_mp1, int n1, const double (*np2)[2], const uint64_t (*desc2)[4],
                               const uint8_t* has_mp2, int n2, int feature_distance, int32_t* match_idx2);

/* ---- pose_oracle.c ---- */
typedef struct orc_pose_obs /* Saiga ObsBase<double> as PoseRefinement.h:47-55 fills it */
{
    double x, y;   /* undistorted keypoint */
    double depth;  /* > 0 => stereo observation */
    double weight; /* sqrt(InverseSquaredScale(octave)) */
} orc_pose_obs;

typedef struct orc_pose_options
{
    double th_mono, th_stereo; /* reprojectionErrorThreshold{Mono,Stereo} * errorFactor */
    int32_t outer_iterations;  /* 4 */
    int32_t inner_iterations;  /* 10 */
    int32_t robust_rounds;     /* 3: Huber in rounds 0..2 */
    int32_t pad;
    double lambda; /* 1e-4 */
} orc_pose_options;

void orc_se3_log_rel(const double* pose, const double* pred, double* e);
void orc_pose_chi2(const double* pose, const orc_camera* cam, const double (*wps)[3], const orc_pose_obs* obs, int n,
                   double* chi2);
int orc_pose_refine(double* pose, const orc_camera* cam, const orc_pose_options* opt, const double (*wps)[3],
                    const orc_pose_obs* obs, int n, const double* prediction, double w_rot, double w_trans,
                    uint8_t* outlier);

#ifdef __cplusplus
}
#endif
#endif
