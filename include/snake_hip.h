/*
 * snake_hip.h — C ABI of the MI355X (gfx950) implementation of Snake-SLAM's
 * per-frame feature pipeline and local bundle adjustment.
 *
 * Every entry point replaces one seam of the reference (darglein/Snake-SLAM);
 * the seam is cited as `path:line` relative to the reference checkout.  The
 * reference has no FFI layer: its seams are C++ member calls into the (absent)
 * saiga submodule, so a C++ adaptor on the Snake side converts std::vector /
 * Eigen to the plain pointers used here (see INTEGRATION.md and
 * snake_slam_amd/cpp/snake_hip.hpp).
 *
 * Conventions
 *  - all functions return an snk_status (0 = ok); nothing aborts or throws;
 *  - "host" entry points take host pointers, are synchronous, and copy through
 *    buffers owned by the handle (the reference's call shape);
 *  - "_dev" entry points take device pointers, enqueue on the handle's stream and
 *    return without synchronising (batched throughput path; inputs resident in HBM);
 *  - a handle is used by one thread at a time; different handles are independent
 *    (reference threading: FeatureDetection || Preprocess || Tracking || LBA);
 *  - a descriptor is 256 bits = 4 x uint64_t, little-endian bit order: bit b of the
 *    descriptor is bit (b & 63) of word (b >> 6).
 */
#ifndef SNAKE_HIP_H
#define SNAKE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNK_API __attribute__((visibility("default")))

typedef enum snk_status
{
    SNK_OK                = 0,
    SNK_ERR_INVALID_ARG   = 1,
    SNK_ERR_NO_DEVICE     = 2, /* no HIP device / HIP runtime failure at create */
    SNK_ERR_HIP           = 3, /* a HIP call failed; see snk_last_error() */
    SNK_ERR_CAPACITY      = 4, /* caller-provided capacity too small */
    SNK_ERR_NOT_CONFIGURED = 5,
    SNK_ERR_TIMEOUT       = 6  /* snk_frontend_collect: no frame arrived within the time given */
} snk_status;

/* "infinite" Hamming distance: the Snake side initialises its running minima with 256
 * (Snake/Tracking/SnakeORBMatcher.cpp:121,280,462) and an absent neighbour is reported so. */
#define SNK_DIST_INF 256

SNK_API const char* snk_last_error(void);  /* thread-local text of the last failure */
SNK_API const char* snk_version(void);
SNK_API int snk_device_count(void);        /* number of visible HIP devices (0 = none) */

/* Run-time switches for choices this library had to DEFINE because the arithmetic lives in the absent saiga submodule
 * (reference .gitmodules:1-3): a maintainer who can read saiga selects the matching rule without editing a kernel.
 * Process-wide, read by every later call; also settable at start-up with SNK_DEFINITIONS="key=value,key=value".
 *   "bf_filter.threshold_strict"  Saiga::BruteForceMatcher::filterMatches(th, ratio), call site
 *                                 Snake/Tracking/TrackingCoarse.cpp:352: 0 = keep d1 <= th (default), 1 = keep d1 < th
 *   "bf_filter.ratio_strict"      same call: 0 = keep d1 <= ratio * d2 (default), 1 = keep d1 < ratio * d2
 *   "iround.mode"                 Saiga::iRound as Preprocess::StereoMatching uses it (Snake/Preprocess/Preprocess.cpp:155,165):
 *                                 0 = floor(x + 0.5) (default), 1 = round half away from zero, 2 = round half to even
 *   "orb.response"                the corner measure of Saiga::ORBExtractor::Detect (ctor call Snake/Preprocess/FeatureDetector.cpp:31-41;
 *                                 the extractor itself is in saiga): 0 = the FAST-9 score (default; published ORB-SLAM2), 1 = the Harris
 *                                 response of the FAST corners (7 x 7 block, k = 0.04: OpenCV's ORB HARRIS_SCORE form).  Under 1 the
 *                                 FAST stage (corners, non-maximum suppression, thresholds, candidate budgets) is unchanged; the Harris
 *                                 response picks the point a quadtree node keeps and is KeyPoint::response.
 * Unknown key / out-of-range value: SNK_ERR_INVALID_ARG, nothing changes.  The oracle mirrors every key (orc_set_definition). */
SNK_API int snk_set_definition(const char* key, int value);
SNK_API int snk_get_definition(const char* key, int* value);

/* ------------------------------------------------------------------------------------------
 * Descriptor matching
 * ------------------------------------------------------------------------------------------ */

/* A rectified / undistorted keypoint as the matchers consume it.  Mirrors the fields of
 * Saiga::KeyPoint<double> that Snake reads: .point, .angle (degrees, float), .octave
 * (Snake/Preprocess/Preprocess.cpp:129-130,214-215; Snake/Map/Features.cpp:24,44). */
typedef struct snk_kp64
{
    double x, y;
    float angle;
    int32_t octave;
} snk_kp64;

/* kNN-2 result for one query descriptor: {idx1, dist1, idx2, dist2}; idx = -1 / dist =
 * SNK_DIST_INF when the train set has fewer than 1 / 2 entries.  A train descriptor at distance
 * 256 (all bits differ) is never reported: 256 is "infinite" on the Snake side. */
typedef struct snk_knn2
{
    int32_t idx1, dist1, idx2, dist2;
} snk_knn2;

typedef struct snk_matcher snk_matcher;

/* device: HIP device ordinal.  stream: a hipStream_t to enqueue on, or NULL for a stream
 * owned by the handle. */
SNK_API int snk_matcher_create(int device, void* stream, snk_matcher** out);
SNK_API int snk_matcher_destroy(snk_matcher* m);
SNK_API int snk_matcher_sync(snk_matcher* m); /* hipStreamSynchronize on the handle's stream */

/* Replaces Saiga::BruteForceMatcher<DescriptorORB>::matchKnn2 / matchKnn2_omp —
 * call site Snake/Tracking/TrackingCoarse.cpp:350-351 (also LoopClosing/LoopORBMatcher.cpp:103-105,
 * Tracking/Initialization/MonoInitializer.cpp:589-592).  For every query the two nearest train
 * descriptors by Hamming distance, ties resolved towards the lower train index. */
SNK_API int snk_bf_knn2(snk_matcher* m, const uint64_t (*query)[4], int nq, const uint64_t (*train)[4], int nt,
                        snk_knn2* out /* nq */);

/* Replaces BruteForceMatcher::filterMatches(threshold, ratio) + the public `matches` vector —
 * Snake/Tracking/TrackingCoarse.cpp:352,373-387.  Keeps (query, idx1) when dist1 <= threshold and
 * dist1 <= ratio * dist2 (float compare), in ascending query order.  pairs has room for nq entries. */
SNK_API int snk_bf_filter(snk_matcher* m, const snk_knn2* knn, int nq, int threshold, float ratio,
                          int32_t (*pairs)[2], int* n_pairs);

/* Batched, device-resident forms.  Layout: batch b owns rows [b*cap, b*cap + n[b]) of each array;
 * counts live on the device (they are produced there by the extractor).  n_pairs_dev[b] receives
 * the number of pairs of batch b. */
SNK_API int snk_bf_knn2_batch_dev(snk_matcher* m, const uint64_t* query_dev, const int32_t* nq_dev, int nq_cap,
                                  const uint64_t* train_dev, const int32_t* nt_dev, int nt_cap, int batch,
                                  snk_knn2* out_dev);
SNK_API int snk_bf_filter_batch_dev(snk_matcher* m, const snk_knn2* knn_dev, const int32_t* nq_dev, int nq_cap,
                                    int batch, int threshold, float ratio, int32_t* pairs_dev /* [batch][nq_cap][2] */,
                                    int32_t* n_pairs_dev);

/* Replaces Snake::Preprocess::StereoMatching(Frame&) — Snake/Preprocess/Preprocess.cpp:122-242
 * (declared Snake/Preprocess/Preprocess.h:33).  left / right are the RECTIFIED keypoints
 * (rect_left.Forward / rect_right.Forward already applied, Preprocess.cpp:140-150); left is in
 * feature-grid order, right in extractor order (Preprocess.cpp:41-49).  level_scale[o] =
 * scalePyramid.Scale(o) (float), n_levels entries.  bf = rect_left.bf.  relaxed =
 * settings.fd_relaxed_stereo.  right_points / depth are in/out (pre-filled with -1000 by
 * Frame::allocateTmp, Snake/Map/Frame.cpp:25-26); only matched entries are overwritten.
 * *n_matches receives the return value of StereoMatching. */
SNK_API int snk_stereo_match(snk_matcher* m, const snk_kp64* left, const uint64_t (*desc_left)[4], int nl,
                             const snk_kp64* right, const uint64_t (*desc_right)[4], int nr, double bf,
                             const float* level_scale, int n_levels, int relaxed, float* right_points, float* depth,
                             int* n_matches);

SNK_API int snk_stereo_match_batch_dev(snk_matcher* m, const snk_kp64* left_dev, const uint64_t* desc_left_dev,
                                       const int32_t* nl_dev, int nl_cap, const snk_kp64* right_dev,
                                       const uint64_t* desc_right_dev, const int32_t* nr_dev, int nr_cap, int batch,
                                       double bf, const float* level_scale_host, int n_levels, int relaxed,
                                       float* right_points_dev, float* depth_dev, int32_t* n_matches_dev);

/* ------------------------------------------------------------------------------------------
 * ORB extractor
 * ------------------------------------------------------------------------------------------ */

/* Constructor arguments of Saiga::ORBExtractor / ORBExtractorGPU as Snake passes them
 * (Snake/Preprocess/FeatureDetector.cpp:31-33,40-41: fd_features, fd_scale_factor, fd_levels,
 * fd_iniThFAST, fd_minThFAST; the CPU extractor's thread count has no meaning here).
 * level_cap bounds the FAST candidates considered per pyramid level (0 = 8192; a power of two
 * in [256, 8192]); see DESIGN.md "candidate budget". */
typedef struct snk_orb_params
{
    int32_t nfeatures;
    float scale_factor;
    int32_t n_levels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
    int32_t level_cap;
} snk_orb_params;

/* Saiga::KeyPoint<float> fields (Snake reads .point, .octave, .angle; FeatureDetector.cpp:129-132). */
typedef struct snk_keypoint
{
    float x, y;     /* level-0 pixel coordinates */
    float size;     /* 31 * scale(octave) */
    float angle;    /* degrees, [0, 360] */
    float response; /* FAST corner score */
    int32_t octave;
} snk_keypoint;

typedef struct snk_orb snk_orb;

SNK_API int snk_orb_create(const snk_orb_params* params, int device, void* stream, snk_orb** out);
SNK_API int snk_orb_destroy(snk_orb* o);
SNK_API int snk_orb_sync(snk_orb* o);

/* Size the device buffers for width x height images and up to max_batch images per call.
 * snk_orb_detect (re)configures itself on the first image / a size change. */
SNK_API int snk_orb_configure(snk_orb* o, int width, int height, int max_batch);
/* Upper bound of keypoints per image for the configured size (output capacity to provide). */
SNK_API int snk_orb_max_keypoints(const snk_orb* o, int* out);

/* Replaces ORBExtractor::Detect / ORBExtractorGPU::Detect(ImageView<uchar>, vector<KeyPoint<float>>&,
 * vector<DescriptorORB>&) — Snake/Preprocess/FeatureDetector.cpp:119,124,149,154.  One synchronous
 * call per image; keypoints and descriptors are index-aligned (FeatureDetector.cpp:169). */
SNK_API int snk_orb_detect(snk_orb* o, const uint8_t* img, int width, int height, int pitch_bytes, snk_keypoint* kps,
                           uint64_t (*desc)[4], int capacity, int* n_out);

/* Batched, device-resident form: image b starts at images_dev + b*image_stride (row pitch
 * pitch_bytes); outputs kps_dev[b*out_cap ...], desc_dev[(b*out_cap + i)*4 ...], n_dev[b].
 * Every image must span pitch_bytes * height readable bytes (image_stride >= that): rows are
 * read in aligned 4-byte units, so up to 3 bytes of a row's padding right of `width` may be
 * read (never used).  Base, pitch and stride that are multiples of 4 take the fast path. */
SNK_API int snk_orb_detect_batch_dev(snk_orb* o, const uint8_t* images_dev, int pitch_bytes, size_t image_stride,
                                     int batch, snk_keypoint* kps_dev, uint64_t* desc_dev, int32_t* n_dev,
                                     int out_cap);

/* Optional per-stage timing with HIP events recorded on the handle's stream around the kernel
 * stages of every detect call: ms[0] = the per-level streaming passes (blurred level + next pyramid
 * level in one pass; plus the stand-alone resize when scale_factor > 2), ms[1] = reserved (the
 * blur used to be a stage of its own; now ~0), ms[2] = FAST cells, ms[3] = distribution,
 * ms[4] = descriptors.  With snk_orb_set_chains(o, 2) a batch of >= 8 images runs as two half-batch
 * launch chains on two streams; each chain is timed on its own stream and counts as one entry of
 * n_calls ("launch chains"), so ms[k] / n_calls is the average duration of one launch and the images
 * per launch are (images processed) / n_calls.
 * snk_orb_stage_times synchronises, returns the summed milliseconds per stage over the launch chains
 * since the last query (ms[5]) and their number, and resets the accumulation. */
SNK_API int snk_orb_set_profiling(snk_orb* o, int enable);

/* Launch chains per snk_orb_detect_batch_dev call (1..4, default 1; environment SNK_ORB_PARTS sets the
 * default).  With more than one chain a batch of >= 8 images is split into equal parts whose kernel
 * sequences run on the handle's stream and on streams owned by the handle, forked from and joined on
 * the handle's stream with events, so the caller-visible ordering is unchanged.  Measured on MI355X:
 * 2 chains +3 % frames/s (the tail of one chain's launch overlaps the other chain), 3-4 chains slower;
 * per-kernel durations are no longer additive.  No reference counterpart. */
SNK_API int snk_orb_set_chains(snk_orb* o, int chains);
/* Staggered schedule of a batch (>= 2 * parts images): the batch is cut into `parts` (2..16; 0 = off) ranges, the front halves
 * (pyramid / blur passes, FAST cells -- bound by vector-instruction issue) run back to back on the handle's stream and the back
 * half of range p (distribution, descriptors -- bound by LDS / cache latency) on a second stream beside the front half of range
 * p + 1.  Same results; the per-kernel times of snk_orb_stage_times then overlap. */
SNK_API int snk_orb_set_stagger(snk_orb* o, int parts);
SNK_API int snk_orb_stage_times(snk_orb* o, float* ms, int* n_calls);

/* Intermediate results of the last call (tests / debugging); no counterpart in the reference — the stages are
 * those of the absent Saiga::ORBExtractor behind Snake/Preprocess/FeatureDetector.cpp:119. */
enum
{
    SNK_ORB_DEBUG_PYRAMID         = 1, /* u8 rows of the level (level >= 1), `pitch` bytes each */
    SNK_ORB_DEBUG_CELL_COUNTS     = 2, /* u16 per FAST cell */
    SNK_ORB_DEBUG_CELL_CANDIDATES = 3, /* u32[64] per cell: score<<12 | (63-dy)<<6 | (63-dx), strongest first */
    SNK_ORB_DEBUG_SELECTED        = 4, /* u32 per slot: x | y<<16 (level pixels) */
    SNK_ORB_DEBUG_SELECTED_COUNT  = 5, /* int */
    SNK_ORB_DEBUG_LEVEL_INFO      = 6  /* int[8]: w, h, pitch, ncols, nrows, wcell, hcell, nfeat */
};
SNK_API int snk_orb_debug_fetch(snk_orb* o, int what, int image, int level, void* out, size_t cap_bytes,
                                size_t* n_bytes);

/* ------------------------------------------------------------------------------------------
 * Preprocess: undistort / rectify keypoints
 * ------------------------------------------------------------------------------------------ */

/* The fields of Saiga::Rectification that Snake uses (rect_left / rect_right,
 * Snake/System/SnakeGlobal.h:107-108; filled by Snake/Preprocess/StereoTransforms.cpp:58-68).
 * D_src is the rational radial-tangential model in the order k1 k2 k3 k4 k5 k6 p1 p2. */
typedef struct snk_rectification
{
    double K_src[4]; /* fx fy cx cy */
    double D_src[8];
    double R[9];     /* row-major 3x3 */
    double K_dst[4]; /* fx fy cx cy */
    double bf;
} snk_rectification;

/* Replaces the per-keypoint body of Snake::Preprocess::undistortKeypoints
 * (Snake/Preprocess/Preprocess.cpp:55-77) and Rectification::Forward as applied in
 * StereoMatching (Preprocess.cpp:140-150): out[i] = rectified pixel position with angle / octave
 * copied, normalized[i] = the point stored in frame.normalized_points (may be NULL). */
SNK_API int snk_rectify(snk_matcher* m, const snk_rectification* rect, const snk_keypoint* kps, int n, snk_kp64* out,
                        double (*normalized)[2]);
SNK_API int snk_rectify_batch_dev(snk_matcher* m, const snk_rectification* rect, const snk_keypoint* kps_dev,
                                  const int32_t* n_dev, int cap, int batch, snk_kp64* out_dev, double* normalized_dev);

/* Replaces Snake::Preprocess::ComputeStereoFromRGBD(Frame&) -- Snake/Preprocess/Preprocess.cpp:79-120, the RGB-D branch of
 * Preprocess::Process (:43-46).  The globals it reads (Snake/System/SnakeGlobal.h): K (fx fy cx cy of the undistorted image),
 * rgbd_intrinsics.depthModel.dis (D_depth, rational radial-tangential order k1..k6 p1 p2), rgbd_intrinsics.depthModel.K (K_depth)
 * and rgbd_intrinsics.bf. */
typedef struct snk_rgbd_model
{
    double K[4];
    double D_depth[8];
    double K_depth[4];
    double bf;
} snk_rgbd_model;

/* undistorted = frame.undistorted_keypoints (n), depth_image = frame.depth_image (float metres, row pitch in floats).  Per keypoint:
 * unproject with K, distort with D_depth, project with K_depth, nearest pixel `(int)(v + 0.5)`; depth > 0: depth[i] = depth,
 * right_points[i] = x - bf / depth; else both -1 (:106-114).  *n_matches = the function's return value.  Where the reference aborts
 * (SAIGA_ASSERT: outside the depth image, depth < 0 or >= 20, :100-104) the call returns SNK_ERR_INVALID_ARG with
 * *n_matches = -(index + 1) of the first such keypoint and leaves the outputs untouched.  The _batch_dev form never fails for data:
 * status_dev[b] = 0x7FFFFFFF (fine) or index + 1 of the first offending keypoint (that frame's outputs are then partial). */
SNK_API int snk_rgbd_stereo(snk_matcher* m, const snk_rgbd_model* model, const snk_kp64* undistorted, int n, const float* depth_image,
                            int width, int height, int pitch_floats, float* right_points, float* depth, int* n_matches);
SNK_API int snk_rgbd_stereo_batch_dev(snk_matcher* m, const snk_rgbd_model* model, const snk_kp64* undistorted_dev, const int32_t* n_dev,
                                      int cap, int batch, const float* depth_images_dev, int width, int height, int pitch_floats,
                                      size_t image_stride_floats, float* right_points_dev, float* depth_dev, int32_t* n_matches_dev,
                                      int32_t* status_dev);

/* ------------------------------------------------------------------------------------------
 * Feature grid and projection-guided tracking matchers
 * ------------------------------------------------------------------------------------------ */

/* Saiga::FeatureGridBounds2<double, 20> as Snake uses it (featureGridBounds,
 * Snake/System/SnakeGlobal.h:115): the extent of the undistorted image; 20-px cells. */
typedef struct snk_grid_bounds
{
    double min_x, min_y, max_x, max_y;
} snk_grid_bounds;

/* Replaces frame.grid.create(featureGridBounds, undistorted_keypoints) —
 * Snake/Preprocess/Preprocess.cpp:246: perm[i] = new index of feature i (the caller scatters its
 * arrays as at :254-260); cell_start[cols*rows + 1] = first feature of every cell, cells ordered
 * x-major (id = cx*rows + cy, the order Features::GetFeaturesInArea walks them,
 * Snake/Map/Features.cpp:17-21). */
SNK_API int snk_feature_grid(snk_matcher* m, const snk_kp64* undistorted, int n, const snk_grid_bounds* bounds,
                             int32_t* perm, int32_t* cell_start, int* cols, int* rows);

/* Batched, device-resident Preprocess::computeFeatureGrid (Preprocess.cpp:244-266): builds the grid
 * of every image and scatters its rectified keypoints and descriptors into grid order
 * (kps_out / desc_out); perm_dev [batch][cap], cell_start_dev [batch][cols*rows + 1]. */
SNK_API int snk_feature_grid_batch_dev(snk_matcher* m, const snk_grid_bounds* bounds, const snk_kp64* kps_dev,
                                       const uint64_t* desc_dev, const int32_t* n_dev, int cap, int batch,
                                       snk_kp64* kps_out_dev, uint64_t* desc_out_dev, int32_t* perm_dev,
                                       int32_t* cell_start_dev);

/* ------------------------------------------------------------------------------------------
 * One stereo frame through the front-end in one call
 * ------------------------------------------------------------------------------------------ */

/* What FeatureDetector + Preprocess are constructed with (Snake/Preprocess/FeatureDetector.cpp:31-41: the extractor arguments;
 * Snake/System/SnakeGlobal.h:107-108,115: rect_left / rect_right, featureGridBounds; Snake/System/Settings.h:123: fd_relaxed_stereo). */
typedef struct snk_frontend_params
{
    snk_orb_params orb;
    snk_rectification rect_left, rect_right; /* undistortKeypoints uses rect_left, StereoMatching's Forward of the right keypoints rect_right */
    snk_grid_bounds bounds;                  /* featureGridBounds */
    double bf;                               /* rect_left.bf / stereo_cam.bf as StereoMatching reads it (Preprocess.cpp:137) */
    int32_t relaxed_stereo;                  /* settings.fd_relaxed_stereo */
    int32_t stereo;                          /* 1 = settings.inputType == Stereo (left + right image, StereoMatching); 0 = mono: left only */
} snk_frontend_params;

/* The members of Snake::Frame that FeatureDetector::Detect and Preprocess::Process fill (Snake/Map/Frame.h, Features.h), as
 * caller-owned arrays of `capacity` entries each (any pointer may be NULL = not wanted).  After the call the left arrays are in
 * FEATURE-GRID order (Preprocess::computeFeatureGrid scatters keypoints, descriptors, undistorted_keypoints and
 * normalized_points, Preprocess.cpp:254-260), the right arrays in extractor order (Preprocess.cpp:41-49). */
typedef struct snk_frontend_frame
{
    int32_t capacity;                 /* in: entries per array (snk_frontend_max_keypoints) */
    int32_t n, n_right, n_stereo;     /* out: frame.N, keypoints_right.size(), the return value of StereoMatching */
    int32_t cols, rows;               /* out: the grid's cell counts (cell id = cx * rows + cy) */
    snk_keypoint* keypoints;          /* frame.keypoints (the reference widens them to KeyPoint<double>; same values) */
    uint64_t (*descriptors)[4];       /* frame.descriptors */
    snk_kp64* undistorted_keypoints;  /* frame.undistorted_keypoints (.point, .angle, .octave) */
    double (*normalized_points)[2];   /* frame.normalized_points */
    int32_t* permutation;             /* what frame.grid.create returned: new index of extractor feature i */
    int32_t* cell_start;              /* cols * rows + 1 entries: first feature of every cell */
    float* right_points;              /* frame.right_points (-1000 where unmatched, Frame.cpp:25) */
    float* depth;                     /* frame.depth (-1000 where unmatched, Frame.cpp:26) */
    snk_keypoint* keypoints_right;    /* frame.keypoints_right */
    uint64_t (*descriptors_right)[4]; /* frame.descriptors_right */
} snk_frontend_frame;

typedef struct snk_frontend snk_frontend;

/* One handle = the reference's FeatureDetector + Preprocess pair for one camera rig: its own stream, extractor and scratch. */
SNK_API int snk_frontend_create(const snk_frontend_params* params, int device, snk_frontend** out);
SNK_API int snk_frontend_destroy(snk_frontend* f);
/* Capacity to provide for images of this size (configures the handle for it). */
SNK_API int snk_frontend_max_keypoints(snk_frontend* f, int width, int height, int* out);
SNK_API int snk_frontend_grid_dims(const snk_frontend* f, int* cols, int* rows);

/* FeatureDetector::Detect (left, right: Snake/Preprocess/FeatureDetector.cpp:116-156) + Preprocess::Process (allocateTmp,
 * undistortKeypoints, computeFeatureGrid, StereoMatching: Snake/Preprocess/Preprocess.cpp:35-53) for ONE frame in ONE synchronous
 * call: one upload of the two images, the extractor as one two-image launch chain, rectification / grid / StereoMatching enqueued
 * behind it, one download, one synchronisation; from the second frame of an image size on the launch sequence is replayed as
 * a hipGraph.  Results are bit for bit those of snk_orb_detect x 2 + snk_rectify x 2 + snk_feature_grid + snk_stereo_match (the same
 * kernels).  level_scale of StereoMatching = the extractor's own pyramid scales (scale[l] = scale[l - 1] * scale_factor in
 * float).  right / pitch_right are ignored by a mono handle.  SNK_ERR_CAPACITY (n / n_right set) when capacity is too small. */
SNK_API int snk_frontend_process(snk_frontend* f, const uint8_t* left, int pitch_left, const uint8_t* right, int pitch_right,
                                 int width, int height, snk_frontend_frame* out);

/* The same frame work PIPELINED, mirroring the reference's blocking single-slot stage queues (SynchronizedSlot<FramePtr>
 * output_buffer: Snake/Preprocess/FeatureDetector.h:39 between FeatureDetection and Preprocess, Preprocess.h:36 between Preprocess and
 * Tracking): the handle owns `depth` independent slots (stream, extractor, scratch, pinned staging, hipGraph, completion event; default 3,
 * snk_frontend_set_depth 1..8 while nothing is in flight).
 *   snk_frontend_submit   stages the images, enqueues upload -> launch chain -> download on the next slot's stream and RETURNS;
 *                         blocks only while all `depth` slots hold uncollected frames (SynchronizedSlot::set).
 *   snk_frontend_collect  waits for the OLDEST submitted frame and fills `out` exactly as snk_frontend_process would (frames
 *                         come back in submission order; bit-identical results).  timeout_ms < 0: wait for a submission as
 *                         long as it takes (SynchronizedSlot::get), 0: do not wait, > 0: that long; SNK_ERR_TIMEOUT when none came.
 *                         SNK_ERR_CAPACITY consumes the frame.
 * One thread may submit while another collects (the reference's FeatureDetection / Preprocess threads); at most one thread on
 * each side.  snk_frontend_process must not be mixed with frames in flight (SNK_ERR_INVALID_ARG); all frames in flight have
 * one image size. */
SNK_API int snk_frontend_set_depth(snk_frontend* f, int depth);
SNK_API int snk_frontend_submit(snk_frontend* f, const uint8_t* left, int pitch_left, const uint8_t* right, int pitch_right,
                                int width, int height);
SNK_API int snk_frontend_collect(snk_frontend* f, snk_frontend_frame* out, int timeout_ms);
SNK_API int snk_frontend_in_flight(snk_frontend* f, int* n);
/* Waits (like snk_frontend_collect: timeout_ms < 0 / 0 / > 0, SNK_ERR_TIMEOUT) until a submitted frame EXISTS -- not until it is
 * finished -- and reports the image size it was submitted with and the capacity its arrays need; the frame stays queued.  This is
 * how a collecting thread sizes its arrays without looking at what the submitting thread is doing (any pointer may be NULL). */
SNK_API int snk_frontend_peek(snk_frontend* f, int timeout_ms, int* width, int* height, int* capacity);
/* snk_frontend_submit without the host-side staging copy (about half of a submit's host time): the upload reads the CALLER's
 * buffers, which must be page-locked (snk_pinned_alloc, hipHostMalloc, hipHostRegister) for the copy to be asynchronous and must
 * stay valid and unmodified until the frame has been collected -- the reference's Input thread owns its image buffers the same
 * way (Snake/Preprocess/Input.h:48, FeatureDetector.h:39).  Rows keep the caller's pitch on the device when pitch_left is a
 * multiple of 4, equals pitch_right and does not exceed the width rounded up to 64 (one copy when right == left + pitch * height,
 * two otherwise); other pitches go through a 2-D copy per image.  Results are bit for bit those of snk_frontend_submit. */
SNK_API int snk_frontend_submit_pinned(snk_frontend* f, const uint8_t* left, int pitch_left, const uint8_t* right, int pitch_right,
                                       int width, int height);
/* Page-locked host memory for the images of snk_frontend_submit_pinned (hipHostMalloc / hipHostFree), for callers that do not link HIP. */
SNK_API int snk_pinned_alloc(size_t bytes, void** out);
SNK_API int snk_pinned_free(void* p);

/* The per-frame data the tracking matchers read (Snake/Map/Features.h:18-41, Frame.h:44-46), in
 * feature-grid order.  taken[i] != 0 <=> frame.mvpMapPoints[i] != nullptr. */
typedef struct snk_frame_view
{
    int32_t n;
    int32_t cols, rows;
    const snk_kp64* kps; /* undistorted_keypoints */
    const uint64_t (*desc)[4];
    const float* right_points;
    const uint8_t* taken;
    const int32_t* cell_start;
    snk_grid_bounds bounds;
} snk_frame_view;

/* K (fx fy cx cy) and stereo_cam.bf (Snake/System/SnakeGlobal.h:103-104). */
typedef struct snk_camera
{
    double fx, fy, cx, cy, bf;
} snk_camera;

/* CoarseTrackingPoint / FineTrackingPoint without the MapPoint* (Snake/Map/LocalMap.h:17-55). */
typedef struct snk_lm_coarse
{
    double pos[3], normal[3];
    uint64_t desc[4];
    int32_t octave;
    float angle;
} snk_lm_coarse;

typedef struct snk_lm_fine
{
    double pos[3], normal[3];
    uint64_t desc[4];
    float reference_depth;
    int32_t reference_scale_level;
    uint8_t valid; /* in/out: cleared by the frustum / distance / viewing-angle culls */
    uint8_t pad[7];
} snk_lm_fine;

/* The Tracking thread looks at one frame with 1-2 coarse calls and one fine call (TrackingCoarse.cpp:234,
 * TrackingFine.cpp:149): snk_match_bind_frame uploads the frame view ONCE and keeps it bound to the handle; the matchers
 * that take a `const snk_frame_view* frame` (snk_match_project_coarse / _fine / _keyframe, snk_match_fuse,
 * snk_match_triangulation_project's frame2; not snk_match_relink, which validates its queries against the view)
 * called with frame == NULL then use the bound frame instead of uploading a view per call.  The view's arrays may be changed or freed after the call.  frame == NULL unbinds. */
SNK_API int snk_match_bind_frame(snk_matcher* m, const snk_frame_view* frame);
/* New `taken` mask (n bytes) for the bound frame: mvpMapPoints changed between two matcher calls. */
SNK_API int snk_match_bound_taken(snk_matcher* m, const uint8_t* taken);

/* Replaces SnakeORBMatcher::SearchByProjectionFrameFrame2 — Snake/Tracking/SnakeORBMatcher.cpp:191-354
 * (call site Snake/Tracking/TrackingCoarse.cpp:234).  pose = CurrentFrame.Pose() (qx qy qz qw tx ty tz).
 * direction: 0 none, 1 bForward, 2 bBackward (:210-212).  match_idx[i] = feature matched to
 * local-map point i or -1 (the adaptor sets mvpMapPoints[match_idx[i]] = lm.points[i].mp);
 * *n_matches = the function's return value. */
SNK_API int snk_match_project_coarse(snk_matcher* m, const snk_frame_view* frame, const snk_camera* cam,
                                     const double pose[7], const snk_lm_coarse* pts, int n_pts, float th, int feature_error,
                                     int direction, const float* level_scale, int n_levels, int32_t* match_idx,
                                     int* n_matches);

/* Replaces SnakeORBMatcher::SearchByProjection2 — Snake/Tracking/SnakeORBMatcher.cpp:365-526 (call
 * site Snake/Tracking/TrackingFine.cpp:149).  pts[i].valid is updated like lmp.valid; visible[i] = 1
 * where the reference calls lmp.mp->IncreaseVisible() (:431). */
SNK_API int snk_match_project_fine(snk_matcher* m, const snk_frame_view* frame, const snk_camera* cam, const double pose[7],
                                   snk_lm_fine* pts, int n_pts, float th, float ratio, const float* level_scale, int n_levels,
                                   int32_t* match_idx, uint8_t* visible, int* n_matches);

/* Device-resident, batched forms of the two per-frame tracking matchers: frame b of the batch is what the batched
 * front-end leaves in HBM -- keypoints and descriptors in feature-grid order and cell_start
 * (snk_feature_grid_batch_dev), right_points (snk_stereo_match_batch_dev) -- so the matchers consume it without a
 * host round trip.  All per-feature arrays are [batch][cap]; cell_start is [batch][cols * rows + 1] for the 20-px grid
 * of `bounds`; taken[b][i] != 0 <=> mvpMapPoints[i] != nullptr (all zero for a fresh frame; snk_match_mark_taken_batch_dev
 * applies the adaptor's `mvpMapPoints[idx] = mp` between the coarse and the fine call). */
typedef struct snk_frames_dev
{
    int32_t batch, cap;
    const int32_t* n;           /* [batch] features per frame */
    const snk_kp64* kps;        /* [batch][cap] undistorted keypoints, grid order */
    const uint64_t* desc;       /* [batch][cap][4] */
    const float* right_points;  /* [batch][cap] */
    const uint8_t* taken;       /* [batch][cap] */
    const int32_t* cell_start;  /* [batch][cols * rows + 1] */
    snk_grid_bounds bounds;
} snk_frames_dev;

/* SearchByProjectionFrameFrame2 for every frame of the batch (same rules and results as snk_match_project_coarse).
 * poses_dev: [batch][7] doubles ON THE DEVICE (qx qy qz qw tx ty tz; e.g. left there by the previous frame's pose
 * refinement); pts_dev [batch][pts_cap], n_pts_dev [batch]; outputs match_idx_dev [batch][pts_cap] (-1 = none, also for
 * entries past n_pts) and n_matches_dev [batch].  Asynchronous on the handle's stream. */
SNK_API int snk_match_project_coarse_batch_dev(snk_matcher* m, const snk_frames_dev* frames, const snk_camera* cam,
                                               const double* poses_dev, const snk_lm_coarse* pts_dev,
                                               const int32_t* n_pts_dev, int pts_cap, float th, int feature_error,
                                               int direction, const float* level_scale, int n_levels,
                                               int32_t* match_idx_dev, int32_t* n_matches_dev);

/* SearchByProjection2 for every frame of the batch (same rules and results as snk_match_project_fine; pts_dev[..].valid is
 * updated in place, visible_dev [batch][pts_cap]). */
SNK_API int snk_match_project_fine_batch_dev(snk_matcher* m, const snk_frames_dev* frames, const snk_camera* cam,
                                             const double* poses_dev, snk_lm_fine* pts_dev, const int32_t* n_pts_dev,
                                             int pts_cap, float th, float ratio, const float* level_scale, int n_levels,
                                             int32_t* match_idx_dev, uint8_t* visible_dev, int32_t* n_matches_dev);

/* The same with the local-map records READ-ONLY: pts_dev[..].valid is read, never written.  The flag the reference leaves in lmp.valid
 * (SnakeORBMatcher.cpp:397-428) is visible_dev[b][i]: a point keeps valid = 1 exactly when it passes every cull, which is where
 * IncreaseVisible() is called (:431).  For callers that match a local map again (or share its records between frames) without restoring
 * it, and 6 x fewer bytes written: the in-place flag is one byte into each 96-byte record. */
SNK_API int snk_match_project_fine_batch_ro_dev(snk_matcher* m, const snk_frames_dev* frames, const snk_camera* cam,
                                                const double* poses_dev, const snk_lm_fine* pts_dev, const int32_t* n_pts_dev,
                                                int pts_cap, float th, float ratio, const float* level_scale, int n_levels,
                                                int32_t* match_idx_dev, uint8_t* visible_dev, int32_t* n_matches_dev);

/* taken_dev[b][match_idx_dev[b][i]] = 1 for every matched point: `CurrentFrame.mvpMapPoints[idx] = mp`
 * (SnakeORBMatcher.cpp:330, :522) applied on the device between two matcher calls on the same frames. */
SNK_API int snk_match_mark_taken_batch_dev(snk_matcher* m, const int32_t* match_idx_dev, const int32_t* n_pts_dev, int pts_cap,
                                           int batch, uint8_t* taken_dev, int cap);

/* Replaces SnakeORBMatcher::SearchByProjectionFrameToKeyframe — Snake/Tracking/SnakeORBMatcher.cpp:71-188.
 * positions / descriptors: mp->getPosition() / mp->GetDescriptor() of kf.GetMapPointMatches();
 * skip[i] != 0 where the keyframe has no point or the frame already holds it (:102-103).
 * Greedy and sequential like the reference: a feature given to point i is unavailable to i+1. */
SNK_API int snk_match_project_keyframe(snk_matcher* m, const snk_frame_view* frame, const snk_camera* cam,
                                       const double pose[7], const double (*positions)[3],
                                       const uint64_t (*descriptors)[4], const uint8_t* skip, int n_pts, float th,
                                       int feature_error, int32_t* match_idx, int* n_matches);

/* ------------------------------------------------------------------------------------------
 * Keyframe-rate matchers of local mapping
 * ------------------------------------------------------------------------------------------ */

/* FusionPoint, all fields (Snake/Map/LocalMap.h:57-80). */
typedef struct snk_fusion_point
{
    double pos[3], normal[3];
    uint64_t desc[4];
    float reference_depth;
    int32_t reference_scale_level;
    int32_t observations;
    int32_t id;
} snk_fusion_point;

/* Replaces MappingORBMatcher::Fuse(kf, pose, point_mask, LocalMap<FusionPoint>, fuseCandidates, th,
 * obs_factor, feature_th) — Snake/LocalMapping/MappingORBMatcher.cpp:359-480 (call sites
 * Snake/LocalMapping/NeighbourSearch.cpp:177,188).  frame = kf->frame (grid order; `taken` unused),
 * point_mask may be NULL.  best_idx[i] = keyframe feature point i would be fused into, or -1; the
 * caller emplaces (best_idx[i], pts[i].id) in point order.  *n_fused = the return value. */
SNK_API int snk_match_fuse(snk_matcher* m, const snk_frame_view* frame, const snk_camera* cam, const double pose[7],
                           const snk_fusion_point* pts, const uint8_t* point_mask, int n_pts, float th, float obs_factor,
                           int feature_th, const float* level_scale, int n_levels, int32_t* best_idx, int* n_fused);

/* Replaces MappingORBMatcher::SearchForTriangulationProject — MappingORBMatcher.cpp:168-249 (call
 * site Snake/LocalMapping/Triangulator.cpp:170).  depth_grid: row-major [grid_rows][grid_cols], one
 * depth per 4 x 4 feature-grid cells (:194-195); kps1 / np1 / desc1 / has_mp1: keyframe 1
 * (undistorted_keypoints, normalized_points, descriptors, GetMapPoint(i) != nullptr), any order;
 * frame2: keyframe 2 in grid order with taken[i] = GetMapPoint(i) != nullptr, np2 its
 * normalized_points; E12 row-major.  Saiga::EpipolarDistanceSquared (absent) is [DEFINED] as the
 * squared distance of np2 to the line E12 * (np1, 1).  match_idx2[i] = feature of keyframe 2 paired
 * with feature i of keyframe 1, or -1. */
SNK_API int snk_match_triangulation_project(snk_matcher* m, const double* depth_grid, int grid_rows, int grid_cols,
                                            const double pose1[7], const double pose2[7], const snk_camera* cam,
                                            const snk_kp64* kps1, const double (*np1)[2], const uint64_t (*desc1)[4],
                                            const uint8_t* has_mp1, int n1, const snk_frame_view* frame2,
                                            const double (*np2)[2], const double E12[9], float epipolar_distance,
                                            int feature_distance, int32_t* match_idx2, int* n_matches);

/* frame->bow_feature_vec (an ordered map vocabulary node -> feature indices, filled by the reference's
 * bag-of-words transform) flattened: node_id strictly ascending, node_start[n_nodes + 1] offsets into
 * `features`, each node's features in the order the reference iterates them. */
typedef struct snk_bow_features
{
    int32_t n_nodes;
    int32_t pad;
    const uint32_t* node_id;
    const int32_t* node_start;
    const int32_t* features;
} snk_bow_features;

/* Replaces MappingORBMatcher::SearchForTriangulation2 — Snake/LocalMapping/MappingORBMatcher.cpp:14-99
 * (call site Snake/LocalMapping/Triangulator.cpp:164).  np / desc / has_mp: normalized_points,
 * descriptors, GetMapPoint(i) != nullptr of each keyframe (any order; the bag-of-words vectors index
 * them).  pairs must hold bow1->node_start[n_nodes] entries; it receives (idx1, idx2) in the order the
 * reference emplaces them; *n_matches = the return value.  tmp_flags of the reference is never set
 * (:26-27, :59), so a keyframe-2 feature may be paired with several keyframe-1 features, as there. */
SNK_API int snk_match_triangulation_bow(snk_matcher* m, const snk_camera* cam, const double E12[9], const double (*np1)[2],
                                        const uint64_t (*desc1)[4], const uint8_t* has_mp1, int n1,
                                        const snk_bow_features* bow1, const double (*np2)[2], const uint64_t (*desc2)[4],
                                        const uint8_t* has_mp2, int n2, const snk_bow_features* bow2,
                                        float epipolar_distance, int feature_distance, int32_t (*pairs)[2], int* n_matches);

/* Replaces MappingORBMatcher::SearchForTriangulationBF — MappingORBMatcher.cpp:102-165: all pairs,
 * epipolar gate of 10 px (:107; the reference ignores its epipolarDistance argument), then the
 * descriptor gate.  match_idx2[i] = feature of keyframe 2 paired with feature i of keyframe 1, or -1. */
SNK_API int snk_match_triangulation_bf(snk_matcher* m, const snk_camera* cam, const double E12[9], const double (*np1)[2],
                                       const uint64_t (*desc1)[4], const uint8_t* has_mp1, int n1, const double (*np2)[2],
                                       const uint64_t (*desc2)[4], const uint8_t* has_mp2, int n2, int feature_distance,
                                       int32_t* match_idx2, int* n_matches);

/* One (keyframe feature, map point) observation of DeferredMapper::Relink — Snake/Optimizer/DeferredMapper.cpp:61-99. */
typedef struct snk_relink_query
{
    double pos[3];        /* mp->getPosition() */
    uint64_t desc[4];     /* mp->descriptor */
    uint64_t alt_desc[4]; /* descriptor of mp's first observation in another keyframe (:90-98), if has_alt */
    int32_t feature;      /* i: the feature of the keyframe that holds mp (grid order) */
    int32_t has_alt;
} snk_relink_query;

#define SNK_RELINK_KEEP 0
#define SNK_RELINK_ERASE 1  /* behind the camera or reprojection error > outlier_threshold (:75-81) */
#define SNK_RELINK_MOVE 2   /* best_idx = a closer feature with a strictly better descriptor (:101-138) */

/* Replaces the per-observation search of DeferredMapper::Relink — Snake/Optimizer/DeferredMapper.cpp:39-165
 * (call site :32); reference constants: radius 0.8, outlier_threshold = reprojectionErrorThresholdMono
 * = 2.1, feature_threshold 25 (:41-43).  frame = kf->frame (grid order; `taken` unused), pose =
 * kf->Pose().  The map edits stay on the caller's side, in feature order as in the reference: ERASE ->
 * EraseMapPointMatch / EraseObservation; MOVE -> if kf->GetMapPoint(best_idx) is a good point at that
 * moment erase (:143-152) else relink (:155-161).  A point moved to a LATER feature is visited again
 * by the reference's loop with its recomputed descriptor: the caller queries it again (n = 1).
 * stereo_cam.LeftPointToRight(x, z) (absent saiga) is [DEFINED] as x - bf / z.
 * *n_changed = number of queries with action != KEEP. */
SNK_API int snk_match_relink(snk_matcher* m, const snk_frame_view* frame, const snk_camera* cam, const double pose[7],
                             const snk_relink_query* queries, int n, float radius, double outlier_threshold,
                             int feature_threshold, int32_t* action, int32_t* best_idx, int* n_changed);

/* ------------------------------------------------------------------------------------------
 * Pose refinement (the step after every projection matcher)
 * ------------------------------------------------------------------------------------------ */

/* Saiga ObsBase<double> as the reference fills it per match — Snake/Tracking/PoseRefinement.h:47-55,
 * PoseRefinement.cpp:46-52: ip = undistorted keypoint, weight = sqrt(InverseSquaredScale(octave)),
 * depth = frame.depth[i] (> 0 => stereo observation, u_r = u - bf / depth). */
typedef struct snk_pose_obs
{
    double x, y;
    double depth;
    double weight;
} snk_pose_obs;

/* th_mono / th_stereo = reprojectionErrorThreshold{Mono,Stereo} * errorFactor (PoseRefinement.cpp:13-15,
 * Snake/System/SnakeGlobal.h:145-146).  The optimiser the reference calls
 * (Saiga::RobustPoseOptimization::optimizePoseRobust, absent submodule) is [DEFINED] as "snk-pose v1"
 * (DESIGN.md 3c): outer_iterations rounds of inner_iterations damped Gauss-Newton steps over the
 * current inliers, Huber (delta = threshold) in the first robust_rounds rounds, every match
 * re-classified after each round (outlier <=> |r|^2 > threshold^2).  Defaults of the Python / C++
 * mirrors: 4, 10, 3, lambda = 1e-4. */
typedef struct snk_pose_options
{
    double th_mono, th_stereo;
    int32_t outer_iterations, inner_iterations, robust_rounds, pad;
    double lambda;
} snk_pose_options;

/* One frame: wps[i] = position of the map point matched to feature idx[i] (lm.points[lid].position
 * or pMP->getPosition()), obs[i] as above; pose = frame.Pose() in, optimised pose out (world ->
 * camera, qx qy qz qw tx ty tz); outlier[i] -> frame.mvbOutlier[idx[i]]; inliers = the return value
 * of optimizePoseRobust.  prediction / w_rot / w_trans = frame.prediction and
 * frame.prediction_weight_{rotation,translation} (RobustSmoothPoseOptimization branch,
 * PoseRefinement.h:68-73); both weights 0 => no prior. */
typedef struct snk_pose_problem
{
    int32_t n;
    int32_t inliers; /* out */
    const double (*wps)[3];
    const snk_pose_obs* obs;
    uint8_t* outlier; /* out [n] */
    double pose[7];   /* in / out */
    double prediction[7];
    double w_rot, w_trans;
} snk_pose_problem;

/* Replaces rpo.optimizePoseRobust / rpo_smooth.optimizePoseRobust for a batch of frames (one
 * wavefront per frame, one launch).  Host pointers, synchronous. */
SNK_API int snk_pose_refine(snk_matcher* m, const snk_camera* cam, const snk_pose_options* opt,
                            snk_pose_problem* problems, int n_problems);

/* Device-resident PoseRefinement::RefinePoseWithMatches (Snake/Tracking/PoseRefinement.cpp:25-79) for every frame of a batch,
 * fed by the result of a batched projection matcher: for local-map point i with match_idx[b][i] = f >= 0 the pair
 * (world point = 3 doubles at pts_dev + (b * pts_cap + i) * pts_stride -- snk_lm_coarse / snk_lm_fine both start with it --,
 * observation = frames->kps[b][f], depth_dev[b][f] (> 0 => stereo), weight sqrt(InverseSquaredScale(octave))) enters in
 * point order.  poses_dev [batch][7] is the start pose and receives the refined one (untouched with fewer than 3 pairs);
 * outlier_dev [batch][pts_cap] gets the flag of every matched point (0 elsewhere), inliers_dev [batch] the return value.
 * Asynchronous on the handle's stream; with snk_match_project_coarse_batch_dev before and _fine_batch_dev after it a frame's
 * tracking chain (TrackingCoarse.cpp:234-270, TrackingFine.cpp:149-158) runs without a host round trip. */
SNK_API int snk_pose_refine_matches_batch_dev(snk_matcher* m, const snk_frames_dev* frames, const float* depth_dev,
                                              const snk_camera* cam, const snk_pose_options* opt, const void* pts_dev,
                                              int pts_stride, const int32_t* match_idx_dev, const int32_t* n_pts_dev,
                                              int pts_cap, const float* level_scale, int n_levels, double* poses_dev,
                                              uint8_t* outlier_dev, int32_t* inliers_dev);

/* The same refinement fed the way the reference feeds it: frame_pt_dev [batch][frames->cap] is `frame.mvpMapPoints` as indices
 * into the frame's point table (pts_dev [batch][pts_cap] records of pts_stride bytes starting with the position; -1 = nullptr,
 * entries >= n_pts_dev[b] are ignored) and the pairs enter in FEATURE order -- the loop `for (auto i : frame.featureRange())` of
 * PoseRefinement.cpp:37-57.  outlier_dev [batch][frames->cap] is `frame.mvbOutlier` (flag of every feature with a point, 0
 * elsewhere); frames->n is required.  Everything else as snk_pose_refine_matches_batch_dev. */
SNK_API int snk_pose_refine_frame_batch_dev(snk_matcher* m, const snk_frames_dev* frames, const float* depth_dev,
                                            const snk_camera* cam, const snk_pose_options* opt, const void* pts_dev,
                                            int pts_stride, const int32_t* frame_pt_dev, const int32_t* n_pts_dev,
                                            int pts_cap, const float* level_scale, int n_levels, double* poses_dev,
                                            uint8_t* outlier_dev, int32_t* inliers_dev);

/* Several sequences per GPU in lockstep (BASELINE.json config 5, "sequences batched"): the two glue steps that keep frame t of
 * every sequence on the device between the batched BF matcher and the batched pose refinement.
 * snk_track_bf_matches_batch_dev -- Tracking::TrackBruteForce, Snake/Tracking/TrackingCoarse.cpp:342-387: the reference matches
 * `matchKnn2_omp(frame.descriptors, ref->frame->descriptors)` (:351) -- the CURRENT frame is the query set -- and sets
 * `frame.mvpMapPoints[m.first] = ref->GetMapPoint(m.second)` for the filtered matches whose reference feature has a point
 * (:373-377).  pairs_dev [batch][cap][2] = (current feature f, reference feature r), n_pairs_dev [batch]: the output of
 * snk_bf_filter_batch_dev with the current frame as query set; ref_has_dev [batch][cap] = the reference feature has a map point;
 * output frame_pt_dev[b][f] = r for those matches, -1 (nullptr) everywhere else -- mvpMapPoints as indices, the form
 * snk_pose_refine_frame_batch_dev consumes with the reference frame's points as the point table.
 * snk_track_backproject_batch_dev -- the stereo points of every frame in the world, p_w = R^T (p_c - t) with
 * p_c = ((x - cx) / fx * z, (y - cy) / fy * z, z), z = depth (the inverse of `currentPose * wp`,
 * Snake/Tracking/SnakeORBMatcher.cpp:229): world_dev [batch][cap][3], has_dev [batch][cap] = depth > 0.  Both asynchronous on the
 * handle's stream. */
SNK_API int snk_track_bf_matches_batch_dev(snk_matcher* m, const int32_t* pairs_dev, const int32_t* n_pairs_dev,
                                           const uint8_t* ref_has_dev, int cap, int batch, int32_t* frame_pt_dev);
SNK_API int snk_track_backproject_batch_dev(snk_matcher* m, const snk_frames_dev* frames, const float* depth_dev,
                                            const snk_camera* cam, const double* poses_dev, double* world_dev, uint8_t* has_dev);

/* ------------------------------------------------------------------------------------------
 * Local bundle adjustment
 * ------------------------------------------------------------------------------------------ */

/* The members of Saiga::OptimizationOptions / BAOptions that Snake sets
 * (Snake/Optimizer/LocalBundleAdjustment.cpp:47-64): maxIterations = 3, maxIterativeIterations = 30,
 * iterativeTolerance = 1e-10, solverType = Iterative with buildExplizitSchur, huberMono / huberStereo
 * = reprojectionErrorThreshold{Mono,Stereo} * lbaErrorFactor (Snake/System/SnakeGlobal.h:145-150).
 * lambda_init: initial LM damping (0 = 1e-4). */
typedef struct snk_ba_options
{
    int32_t max_iterations;
    int32_t max_pcg_iterations;
    double pcg_tol;
    double huber_mono, huber_stereo;
    double lambda_init;
} snk_ba_options;

/* One Saiga::Scene as MakeLocalScene fills it (LocalBundleAdjustment.cpp:187-293), flattened:
 * images[i] = {se3 (world -> camera; quaternion x y z w, translation), constant};
 * worldPoints[j] = {p, constant}; one entry per StereoImagePoint = {image, wp, point (pixels),
 * depth (> 0 => stereo observation; u_r = u - bf/depth), weight}; intrinsics[0] = K; scene.bf. */
/* Saiga RelPoseConstraint as MakeLocalScene fills it when the IMU is enabled
 * (LocalBundleAdjustment.cpp:294-346): img1 / img2 = id_in_scene of two consecutive keyframes,
 * rel_pose = the pre-integrated relative pose, weight_rotation = gyro weight / dt,
 * weight_translation = acc weight / dt (0 for dt > 2 s).  [DEFINED] (the residual lives in the absent
 * saiga): rel_pose is T_img2 * T_img1^-1 for world -> camera poses; e = log(T2 T1^-1 rel^-1) = (rho,
 * omega); r = (weight_translation * rho, weight_rotation * omega); cost += |r|^2 (no robust kernel);
 * Gauss-Newton Jacobians with J_l^-1 ~ I: d r / d delta2 = W, d r / d delta1 = -W Ad(T2 T1^-1). */
typedef struct snk_ba_rpc
{
    int32_t img1, img2;
    double rel_pose[7];
    double weight_rotation, weight_translation;
} snk_ba_rpc;

typedef struct snk_ba_problem
{
    int32_t n_img, n_pt, n_obs;
    double (*pose)[7];
    const uint8_t* img_const;
    double (*pt)[3];
    const uint8_t* pt_const;
    const int32_t* obs_img;
    const int32_t* obs_pt;
    const double (*obs_uv)[2];
    const double* obs_depth;
    const double* obs_weight;
    double K[4]; /* fx fy cx cy */
    double bf;
    int32_t n_rpc; /* scene.rel_pose_constraints (0 without IMU) */
    int32_t pad;
    const snk_ba_rpc* rpc;
} snk_ba_problem;

typedef struct snk_ba snk_ba;

SNK_API int snk_ba_create(const snk_ba_options* options, int device, void* stream, snk_ba** out);
SNK_API int snk_ba_destroy(snk_ba* h);
SNK_API int snk_ba_sync(snk_ba* h);

/* Replaces BARecRel::create(scene) — LocalBundleAdjustment.cpp:359: analyse the structure, copy
 * the problem to the device.  set_problems loads `count` independent windows that are solved side
 * by side (one launch sequence for all).  The arrays are copied; they need not stay alive.
 * Limits: count <= 65535; n_img <= 32767 per problem (SNK_ERR_INVALID_ARG beyond). */
SNK_API int snk_ba_set_problem(snk_ba* h, const snk_ba_problem* problem);
SNK_API int snk_ba_set_problems(snk_ba* h, const snk_ba_problem* problems, int count);

/* Observations flagged here are ignored by solve / residuals, like StereoImagePoint::outlier
 * (LocalBundleAdjustment.cpp:382,391).  obs_outlier has n_obs entries in caller order; NULL clears. */
SNK_API int snk_ba_set_outliers(snk_ba* h, int problem, const uint8_t* obs_outlier);

/* Replaces BARecRel::initAndSolve() / solve() — LocalBundleAdjustment.cpp:365,407: `iterations`
 * LM iterations on every loaded problem (state stays on the device; fetch it with
 * snk_ba_get_state).  cost_initial / cost_final (one per problem, may be NULL) mirror
 * OptimizationResults::cost_initial / cost_final (LocalBundleAdjustment.cpp:412). */
SNK_API int snk_ba_solve(snk_ba* h, int iterations, double* cost_initial, double* cost_final);
SNK_API int snk_ba_solve_async(snk_ba* h, int iterations); /* enqueue only */
SNK_API int snk_ba_reset(snk_ba* h);                       /* restore the poses / points given to set_problems */

/* Optimised scene.images[i].se3 / scene.worldPoints[j].p (LocalBundleAdjustment.cpp:485-498). */
SNK_API int snk_ba_get_state(snk_ba* h, int problem, double (*pose)[7], double (*pt)[3], int* pcg_iterations);

/* Replaces the per-observation Scene::residual3 / residual2 squared norms of the chi-square outlier
 * passes (LocalBundleAdjustment.cpp:372-395, 423-457): chi2_per_obs[o] = |weighted residual|^2 at the
 * current state (0 for skipped observations), n_obs entries in caller order. */
SNK_API int snk_ba_residuals(snk_ba* h, int problem, double* chi2_per_obs);

/* LocalBundleAdjustment::SolveLocalScene after scene creation in ONE call (LocalBundleAdjustment.cpp:357-410): initAndSolve with
 * the handle's max_iterations (:357-365), the chi-square pass on the device (:368-397: every observation that is valid, not
 * yet an outlier and has chi2 > (depth > 0 ? chi2_stereo : chi2_mono) becomes an outlier), `extra_iterations` more iterations
 * when anything was marked (:399-410; the reference uses 1).  One 4-byte count crosses the bus in between instead of the chi2
 * array down and the mask up.  Results: *n_marked (outlierPoints), the FIRST solve's costs (what :412 returns), the optimised
 * poses / points of `problem` and its observation outlier flags after the pass (n_obs bytes, caller order); each may be NULL
 * except n_marked.  Same results as snk_ba_solve -> snk_ba_residuals -> host threshold -> snk_ba_set_outliers -> snk_ba_solve.
 * With several problems loaded every problem is solved and marked, and each one that had something marked gets the extra
 * iterations (decided on the device: the call synchronises once, at the end); the outputs are those of `problem`. */
SNK_API int snk_ba_solve_local_scene(snk_ba* h, int problem, double chi2_mono, double chi2_stereo, int extra_iterations,
                                     uint8_t* obs_outlier, int* n_marked, double* cost_initial, double* cost_final,
                                     double (*pose)[7], double (*pt)[3]);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU result gather (BASELINE config 5: one sequence per GPU)
 * ------------------------------------------------------------------------------------------
 * The path shards over independent units -- one Snake-SLAM process and one sequence per GPU -- and exchanges nothing while it
 * runs.  When a rank is done it holds what the reference writes per run: the TUM trajectory (Snake/System/System.cpp:552-563,
 * "timestamp tx ty tz qx qy qz qw" per frame) and a few counters.  These entry points gather such fixed-size blocks across the
 * ranks with ONE RCCL all-gather over xGMI, without torch.distributed or MPI in the process.  RCCL is opened at run time (dlopen
 * by SONAME librccl.so.1; SNK_RCCL_LIB overrides): the library has no link-time dependency on it.
 * Replaces: nothing in the reference (it is a single-GPU program); serves the "RCCL/xGMI only for result gather" of north_star.
 * One handle per process / GPU; calls on a handle are serial. */
typedef struct snk_dist snk_dist;
#define SNK_DIST_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */

/* Rank 0 creates the 128-byte communicator id and hands it to the other ranks by any means (environment, file, socket). */
SNK_API int snk_dist_get_unique_id(uint8_t id[SNK_DIST_ID_BYTES]);
/* Collective over all `world` ranks (ncclCommInitRank): rank r runs on HIP device `device`.  Returns when every rank has called. */
SNK_API int snk_dist_init(const uint8_t id[SNK_DIST_ID_BYTES], int rank, int world, int device, snk_dist** out);
/* The same with the id passed through a file all ranks can see: rank 0 removes whatever an earlier job left at `path`, writes
 * the id (atomically: `path`.tmp + rename), the others wait up to timeout_s seconds (<= 0: 60) for it.  Rank 0 removes the file
 * again in snk_dist_destroy and when its init fails.  Use a fresh path per job, or start rank 0 first: a rank that reads a
 * stale file before rank 0 has removed it joins a dead communicator. */
SNK_API int snk_dist_init_file(const char* path, int rank, int world, int device, double timeout_s, snk_dist** out);
SNK_API int snk_dist_destroy(snk_dist* d);
SNK_API int snk_dist_rank(const snk_dist* d, int* rank, int* world);
/* recv[r * bytes .. (r + 1) * bytes) = rank r's `send` block, on every rank; every rank passes the same `bytes`.  Host memory
 * (staged through one pinned buffer: upload, ncclAllGather on the handle's stream, download, one synchronisation). */
SNK_API int snk_dist_all_gather(snk_dist* d, const void* send, size_t bytes, void* recv);
/* Device memory, no staging; returns after the handle's stream has drained. */
SNK_API int snk_dist_all_gather_dev(snk_dist* d, const void* send_dev, size_t bytes, void* recv_dev);
/* Maximum of one value per rank on every rank (the longest trajectory: blocks are padded to it); doubles as a barrier. */
SNK_API int snk_dist_max_i64(snk_dist* d, int64_t value, int64_t* out);
/* ncclGetVersion of the RCCL this process resolved (SNK_ERR_NO_DEVICE when none can be opened). */
SNK_API int snk_dist_rccl_version(int* version);

#ifdef __cplusplus
}
#endif
#endif /* SNAKE_HIP_H */
