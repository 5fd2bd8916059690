// snake_hip.hpp — header-only C++17 adaptor over the C ABI (include/snake_hip.h) with the call
// shapes of the saiga classes Snake-SLAM uses on this path, so the Snake side changes types, not
// call sites.  Error style: the reference aborts through SAIGA_ASSERT /
// SAIGA_EXIT_ERROR (e.g. Snake/Preprocess/FeatureDetector.cpp:35); here every non-zero status
// becomes a std::runtime_error carrying snk_last_error().
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "snake_hip.h"

namespace snake_hip
{
using DescriptorORB = std::array<uint64_t, 4>;  // Saiga::DescriptorORB (256 bit, trivially copyable)
using KeyPointF     = snk_keypoint;             // Saiga::KeyPoint<float>: point, size, angle, response, octave

inline void check(int rc, const char* what)
{
    if (rc != SNK_OK) throw std::runtime_error(std::string(what) + ": " + snk_last_error());
}

// Saiga::ORBExtractor / ORBExtractorGPU — Snake/Preprocess/FeatureDetector.cpp:31-41,119,124
class ORBExtractor
{
   public:
    ORBExtractor(int nfeatures, float scale_factor, int levels, int iniThFAST, int minThFAST, int /*threads*/ = 0,
                 int device = 0)
    {
        snk_orb_params p{nfeatures, scale_factor, levels, iniThFAST, minThFAST, 0};
        check(snk_orb_create(&p, device, nullptr, &h_), "snk_orb_create");
    }
    ~ORBExtractor() { snk_orb_destroy(h_); }
    ORBExtractor(const ORBExtractor&)            = delete;
    ORBExtractor& operator=(const ORBExtractor&) = delete;

    // Detect(ImageView<uchar>, vector<KeyPoint<float>>&, vector<DescriptorORB>&)
    void Detect(const uint8_t* data, int width, int height, int pitch_bytes, std::vector<KeyPointF>& keypoints,
                std::vector<DescriptorORB>& descriptors)
    {
        check(snk_orb_configure(h_, width, height, 1), "snk_orb_configure");
        int cap = 0;
        check(snk_orb_max_keypoints(h_, &cap), "snk_orb_max_keypoints");
        keypoints.resize((size_t)cap);
        descriptors.resize((size_t)cap);
        int n = 0;
        check(snk_orb_detect(h_, data, width, height, pitch_bytes, keypoints.data(),
                             reinterpret_cast<uint64_t(*)[4]>(descriptors.data()), cap, &n),
              "snk_orb_detect");
        keypoints.resize((size_t)n);
        descriptors.resize((size_t)n);
    }

   private:
    snk_orb* h_ = nullptr;
};

// Saiga::BruteForceMatcher<DescriptorORB> — Snake/Tracking/TrackingCoarse.cpp:350-352,373-387
class BruteForceMatcher
{
   public:
    explicit BruteForceMatcher(int device = 0) { check(snk_matcher_create(device, nullptr, &h_), "snk_matcher_create"); }
    ~BruteForceMatcher() { snk_matcher_destroy(h_); }
    BruteForceMatcher(const BruteForceMatcher&)            = delete;
    BruteForceMatcher& operator=(const BruteForceMatcher&) = delete;

    void matchKnn2(const std::vector<DescriptorORB>& d1, const std::vector<DescriptorORB>& d2)
    {
        knn_.resize(d1.size());
        check(snk_bf_knn2(h_, reinterpret_cast<const uint64_t(*)[4]>(d1.data()), (int)d1.size(),
                          reinterpret_cast<const uint64_t(*)[4]>(d2.data()), (int)d2.size(), knn_.data()),
              "snk_bf_knn2");
    }
    void matchKnn2_omp(const std::vector<DescriptorORB>& d1, const std::vector<DescriptorORB>& d2, int /*threads*/)
    {
        matchKnn2(d1, d2);
    }
    int filterMatches(int threshold, float ratio)
    {
        std::vector<std::array<int32_t, 2>> pairs(knn_.size() + 1);
        int n = 0;
        check(snk_bf_filter(h_, knn_.data(), (int)knn_.size(), threshold, ratio, reinterpret_cast<int32_t(*)[2]>(pairs.data()),
                            &n),
              "snk_bf_filter");
        matches.resize((size_t)n);
        for (int i = 0; i < n; ++i) matches[(size_t)i] = {pairs[(size_t)i][0], pairs[(size_t)i][1]};
        return n;
    }
    std::vector<std::pair<int, int>> matches;  // (index into d1, index into d2)
    const std::vector<snk_knn2>& knn() const { return knn_; }

   private:
    snk_matcher* h_ = nullptr;
    std::vector<snk_knn2> knn_;
};

// Snake::Preprocess::undistortKeypoints + StereoMatching — Snake/Preprocess/Preprocess.cpp:55-77,122-242
class Preprocess
{
   public:
    explicit Preprocess(int device = 0) { check(snk_matcher_create(device, nullptr, &h_), "snk_matcher_create"); }
    ~Preprocess() { snk_matcher_destroy(h_); }
    Preprocess(const Preprocess&)            = delete;
    Preprocess& operator=(const Preprocess&) = delete;

    // rect.Forward for every keypoint (angle / octave copied); normalized may be null
    void Rectify(const snk_rectification& rect, const std::vector<KeyPointF>& kps, std::vector<snk_kp64>& out,
                 std::vector<std::array<double, 2>>* normalized = nullptr)
    {
        out.resize(kps.size());
        if (normalized) normalized->resize(kps.size());
        check(snk_rectify(h_, &rect, kps.data(), (int)kps.size(), out.data(),
                          normalized ? reinterpret_cast<double(*)[2]>(normalized->data()) : nullptr),
              "snk_rectify");
    }

    // Preprocess::ComputeStereoFromRGBD (Preprocess.cpp:79-120): right_points / depth from the depth image; returns the match count.
    // Throws where the reference aborts (a keypoint outside the depth image, a depth outside [0, 20)).
    int ComputeStereoFromRGBD(const snk_rgbd_model& model, const std::vector<snk_kp64>& undistorted, const float* depth_image, int width,
                              int height, int pitch_floats, std::vector<float>& right_points, std::vector<float>& depth)
    {
        right_points.resize(undistorted.size());
        depth.resize(undistorted.size());
        int n = 0;
        check(snk_rgbd_stereo(h_, &model, undistorted.data(), (int)undistorted.size(), depth_image, width, height, pitch_floats,
                              right_points.data(), depth.data(), &n),
              "snk_rgbd_stereo");
        return n;
    }

    // returns the number of stereo matches; right_points / depth keep their -1000 fill where unmatched
    int StereoMatching(const std::vector<snk_kp64>& left_rectified, const std::vector<DescriptorORB>& desc_left,
                       const std::vector<snk_kp64>& right_rectified, const std::vector<DescriptorORB>& desc_right, double bf,
                       const std::vector<float>& level_scale, bool relaxed, std::vector<float>& right_points,
                       std::vector<float>& depth)
    {
        right_points.resize(left_rectified.size(), -1000.0f);
        depth.resize(left_rectified.size(), -1000.0f);
        int n = 0;
        check(snk_stereo_match(h_, left_rectified.data(), reinterpret_cast<const uint64_t(*)[4]>(desc_left.data()),
                               (int)left_rectified.size(), right_rectified.data(),
                               reinterpret_cast<const uint64_t(*)[4]>(desc_right.data()), (int)right_rectified.size(), bf,
                               level_scale.data(), (int)level_scale.size(), relaxed ? 1 : 0, right_points.data(),
                               depth.data(), &n),
              "snk_stereo_match");
        return n;
    }

   private:
    snk_matcher* h_ = nullptr;
};

// FeatureDetector::Detect (left + right) and Preprocess::Process for one frame in ONE call and one synchronisation
// (snk_frontend_process) -- Snake/Preprocess/FeatureDetector.cpp:116-156, Snake/Preprocess/Preprocess.cpp:35-53.  The result
// vectors hold what those two modules leave in Snake::Frame: left arrays in feature-grid order, right arrays in extractor order.
struct FrontendResult
{
    std::vector<KeyPointF> keypoints, keypoints_right;
    std::vector<DescriptorORB> descriptors, descriptors_right;
    std::vector<snk_kp64> undistorted_keypoints;
    std::vector<std::array<double, 2>> normalized_points;
    std::vector<int32_t> permutation, cell_start;
    std::vector<float> right_points, depth;
    int cols = 0, rows = 0, stereo_matches = 0;
};
class Frontend
{
   public:
    Frontend(const snk_frontend_params& p, int device = 0) : p_(p)
    {
        check(snk_frontend_create(&p, device, &h_), "snk_frontend_create");
        check(snk_frontend_grid_dims(h_, &cols_, &rows_), "snk_frontend_grid_dims");  // constants of the handle (the bounds / 20 px)
    }
    ~Frontend() { snk_frontend_destroy(h_); }
    Frontend(const Frontend&)            = delete;
    Frontend& operator=(const Frontend&) = delete;

    // right may be null for a mono handle; returns the number of stereo matches
    int Process(const uint8_t* left, int pitch_left, const uint8_t* right, int pitch_right, int width, int height, FrontendResult& r)
    {
        snk_frontend_frame f = bind(width, height, r);
        check(snk_frontend_process(h_, left, pitch_left, right, pitch_right, width, height, &f), "snk_frontend_process");
        return finish(f, r);
    }
    // The pipelined form: Submit returns as soon as the frame is enqueued (it blocks only while `depth` frames are uncollected, like
    // SynchronizedSlot::set of FeatureDetector::output_buffer, Snake/Preprocess/FeatureDetector.h:39); Collect hands out the oldest
    // frame, bit for bit what Process returns (SynchronizedSlot::get).  One thread may Submit while another Collects: the two share
    // nothing in this object (Collect sizes its arrays from what the C layer reports for the frame it is about to receive).
    void SetDepth(int depth) { check(snk_frontend_set_depth(h_, depth), "snk_frontend_set_depth"); }
    void Submit(const uint8_t* left, int pitch_left, const uint8_t* right, int pitch_right, int width, int height)
    {
        check(snk_frontend_submit(h_, left, pitch_left, right, pitch_right, width, height), "snk_frontend_submit");
    }
    // the images are page-locked memory of the caller (PinnedImage below) that stays untouched until the frame has been collected:
    // no staging copy (Snake/Preprocess/Input.h:48 -- the Input thread owns its image buffers)
    void SubmitPinned(const uint8_t* left, int pitch_left, const uint8_t* right, int pitch_right, int width, int height)
    {
        check(snk_frontend_submit_pinned(h_, left, pitch_left, right, pitch_right, width, height), "snk_frontend_submit_pinned");
    }
    int Collect(FrontendResult& r, int timeout_ms = -1)
    {
        int cap = 0;
        check(snk_frontend_peek(h_, timeout_ms, nullptr, nullptr, &cap), "snk_frontend_peek");  // blocks until a frame has been submitted
        snk_frontend_frame f = bind_capacity(cap, r);
        check(snk_frontend_collect(h_, &f, timeout_ms), "snk_frontend_collect");
        return finish(f, r);
    }
    int InFlight()
    {
        int n = 0;
        check(snk_frontend_in_flight(h_, &n), "snk_frontend_in_flight");
        return n;
    }
    const snk_frontend_params& params() const { return p_; }

   private:
    // result vectors at capacity, pointers into them (Process: the calling thread is the only user of the handle)
    snk_frontend_frame bind(int width, int height, FrontendResult& r)
    {
        if (cap_ == 0 || width != cap_w_ || height != cap_h_)
        {
            check(snk_frontend_max_keypoints(h_, width, height, &cap_), "snk_frontend_max_keypoints");
            cap_w_ = width, cap_h_ = height;
        }
        return bind_capacity(cap_, r);
    }
    snk_frontend_frame bind_capacity(int cap, FrontendResult& r) const
    {
        r.cols = cols_, r.rows = rows_;
        const size_t c = (size_t)cap;
        r.keypoints.resize(c), r.keypoints_right.resize(c), r.descriptors.resize(c), r.descriptors_right.resize(c);
        r.undistorted_keypoints.resize(c), r.normalized_points.resize(c), r.permutation.resize(c), r.right_points.resize(c), r.depth.resize(c);
        r.cell_start.resize((size_t)r.cols * r.rows + 1);
        snk_frontend_frame f{};
        f.capacity              = cap;
        f.keypoints             = r.keypoints.data();
        f.descriptors           = reinterpret_cast<uint64_t(*)[4]>(r.descriptors.data());
        f.undistorted_keypoints = r.undistorted_keypoints.data();
        f.normalized_points     = reinterpret_cast<double(*)[2]>(r.normalized_points.data());
        f.permutation           = r.permutation.data();
        f.cell_start            = r.cell_start.data();
        f.right_points          = r.right_points.data();
        f.depth                 = r.depth.data();
        f.keypoints_right       = r.keypoints_right.data();
        f.descriptors_right     = reinterpret_cast<uint64_t(*)[4]>(r.descriptors_right.data());
        return f;
    }
    static int finish(const snk_frontend_frame& f, FrontendResult& r)
    {
        const size_t n = (size_t)f.n, nr = (size_t)f.n_right;
        r.keypoints.resize(n), r.descriptors.resize(n), r.undistorted_keypoints.resize(n), r.normalized_points.resize(n);
        r.permutation.resize(n), r.right_points.resize(n), r.depth.resize(n);
        r.keypoints_right.resize(nr), r.descriptors_right.resize(nr);
        r.stereo_matches = f.n_stereo;
        return f.n_stereo;
    }
    snk_frontend* h_ = nullptr;
    snk_frontend_params p_;
    int cap_ = 0, cap_w_ = 0, cap_h_ = 0, cols_ = 0, rows_ = 0;  // cap_*: Process only; cols_ / rows_: set once in the constructor
};

// Page-locked image memory for Frontend::SubmitPinned (snk_pinned_alloc): what Snake's Input thread would allocate its image buffers from.
class PinnedImage
{
   public:
    PinnedImage(int pitch, int height, int count = 1) : pitch_(pitch), height_(height)
    {
        void* p = nullptr;
        check(snk_pinned_alloc((size_t)pitch * height * count, &p), "snk_pinned_alloc");
        p_ = static_cast<uint8_t*>(p);
    }
    ~PinnedImage() { snk_pinned_free(p_); }
    PinnedImage(const PinnedImage&)            = delete;
    PinnedImage& operator=(const PinnedImage&) = delete;
    uint8_t* data(int image = 0) { return p_ + (size_t)image * pitch_ * height_; }
    int pitch() const { return pitch_; }

   private:
    uint8_t* p_ = nullptr;
    int pitch_ = 0, height_ = 0;
};

// Frame data the tracking matchers read, grid-ordered (Snake/Map/Features.h:18-41, Frame.h:44-46).
struct FrameView
{
    std::vector<snk_kp64> undistorted_keypoints;
    std::vector<DescriptorORB> descriptors;
    std::vector<float> right_points;
    std::vector<uint8_t> taken;  // mvpMapPoints[i] != nullptr
    std::vector<int32_t> cell_start;
    int cols = 0, rows = 0;
    snk_grid_bounds bounds{};

    snk_frame_view view() const
    {
        snk_frame_view v{};
        v.n            = (int)undistorted_keypoints.size();
        v.cols         = cols;
        v.rows         = rows;
        v.kps          = undistorted_keypoints.data();
        v.desc         = reinterpret_cast<const uint64_t(*)[4]>(descriptors.data());
        v.right_points = right_points.data();
        v.taken        = taken.data();
        v.cell_start   = cell_start.data();
        v.bounds       = bounds;
        return v;
    }
};

// frame.grid.create(featureGridBounds, undistorted_keypoints) + the three SnakeORBMatcher searches
// (Snake/Preprocess/Preprocess.cpp:246; Snake/Tracking/SnakeORBMatcher.h:21-30)
class SnakeORBMatcher
{
   public:
    explicit SnakeORBMatcher(int device = 0) { check(snk_matcher_create(device, nullptr, &h_), "snk_matcher_create"); }
    ~SnakeORBMatcher() { snk_matcher_destroy(h_); }
    SnakeORBMatcher(const SnakeORBMatcher&)            = delete;
    SnakeORBMatcher& operator=(const SnakeORBMatcher&) = delete;

    // returns the permutation (new index of every feature); fills frame.cell_start / cols / rows
    std::vector<int32_t> CreateGrid(FrameView& frame, const snk_grid_bounds& bounds)
    {
        frame.bounds = bounds;
        const int n  = (int)frame.undistorted_keypoints.size();
        std::vector<int32_t> perm((size_t)n + 1);
        const int cols = std::max(1, (int)std::ceil((bounds.max_x - bounds.min_x) / 20.0));
        const int rows = std::max(1, (int)std::ceil((bounds.max_y - bounds.min_y) / 20.0));
        frame.cell_start.assign((size_t)cols * rows + 1, 0);
        check(snk_feature_grid(h_, frame.undistorted_keypoints.data(), n, &bounds, perm.data(), frame.cell_start.data(),
                               &frame.cols, &frame.rows),
              "snk_feature_grid");
        perm.resize((size_t)n);
        return perm;
    }
    // The Tracking thread makes 1-2 coarse calls and one fine call on the same frame: BindFrame uploads the frame once and returns
    // a TOKEN; the Search* overloads that take the token send nothing but the points.  The binding is explicit -- a FrameView
    // object that is refilled or re-created at the same address can never be mistaken for the frame on the device (round 2
    // compared addresses) -- and a token of an earlier binding is refused (std::logic_error).  After changing mvpMapPoints
    // (taken) between two calls: UpdateTaken(token, taken).  The Search* overloads that take a FrameView upload it, always.
    struct BoundFrame
    {
        uint64_t id = 0;
        int n       = 0;
    };
    BoundFrame BindFrame(const FrameView& frame)
    {
        const snk_frame_view v = frame.view();
        check(snk_match_bind_frame(h_, &v), "snk_match_bind_frame");
        bound_ = BoundFrame{++bind_counter_, v.n};
        return bound_;
    }
    void UpdateTaken(const BoundFrame& b, const std::vector<uint8_t>& taken)
    {
        require_bound(b);
        if ((int)taken.size() != b.n) throw std::invalid_argument("UpdateTaken: taken mask size differs from the bound frame's");
        check(snk_match_bound_taken(h_, taken.data()), "snk_match_bound_taken");
    }
    void UnbindFrame()
    {
        check(snk_match_bind_frame(h_, nullptr), "snk_match_bind_frame");
        bound_ = BoundFrame{};
    }

    // match[i] = feature index for local-map point i or -1; the caller sets mvpMapPoints[match[i]] = lm.points[i].mp
    int SearchByProjectionFrameFrame2(const FrameView& frame, const snk_camera& K, const double pose[7],
                                      const std::vector<snk_lm_coarse>& lm, float th, int featureError, int direction,
                                      const std::vector<float>& level_scale, std::vector<int32_t>& match)
    {
        const snk_frame_view v = frame.view();
        return coarse(&v, K, pose, lm, th, featureError, direction, level_scale, match);
    }
    int SearchByProjectionFrameFrame2(const BoundFrame& b, const snk_camera& K, const double pose[7],
                                      const std::vector<snk_lm_coarse>& lm, float th, int featureError, int direction,
                                      const std::vector<float>& level_scale, std::vector<int32_t>& match)
    {
        require_bound(b);
        return coarse(nullptr, K, pose, lm, th, featureError, direction, level_scale, match);
    }
    int SearchByProjection2(const FrameView& frame, const snk_camera& K, const double pose[7], std::vector<snk_lm_fine>& lm,
                            float th, float ratio, const std::vector<float>& level_scale, std::vector<int32_t>& match,
                            std::vector<uint8_t>& visible)
    {
        const snk_frame_view v = frame.view();
        return fine(&v, K, pose, lm, th, ratio, level_scale, match, visible);
    }
    int SearchByProjection2(const BoundFrame& b, const snk_camera& K, const double pose[7], std::vector<snk_lm_fine>& lm, float th,
                            float ratio, const std::vector<float>& level_scale, std::vector<int32_t>& match,
                            std::vector<uint8_t>& visible)
    {
        require_bound(b);
        return fine(nullptr, K, pose, lm, th, ratio, level_scale, match, visible);
    }
    int SearchByProjectionFrameToKeyframe(const FrameView& frame, const snk_camera& K, const double pose[7],
                                          const std::vector<std::array<double, 3>>& positions,
                                          const std::vector<DescriptorORB>& descriptors, const std::vector<uint8_t>& skip, float th,
                                          int featureError, std::vector<int32_t>& match)
    {
        const snk_frame_view v = frame.view();
        match.assign(positions.size() + 1, -1);
        int n = 0;
        check(snk_match_project_keyframe(h_, &v, &K, pose, reinterpret_cast<const double(*)[3]>(positions.data()),
                                         reinterpret_cast<const uint64_t(*)[4]>(descriptors.data()), skip.data(),
                                         (int)positions.size(), th, featureError, match.data(), &n),
              "snk_match_project_keyframe");
        match.resize(positions.size());
        return n;
    }

   private:
    void require_bound(const BoundFrame& b) const
    {
        if (b.id == 0 || b.id != bound_.id) throw std::logic_error("stale or empty frame binding (BindFrame again)");
    }
    int coarse(const snk_frame_view* v, const snk_camera& K, const double pose[7], const std::vector<snk_lm_coarse>& lm, float th,
               int featureError, int direction, const std::vector<float>& level_scale, std::vector<int32_t>& match)
    {
        match.assign(lm.size() + 1, -1);
        int n = 0;
        check(snk_match_project_coarse(h_, v, &K, pose, lm.data(), (int)lm.size(), th, featureError, direction, level_scale.data(),
                                       (int)level_scale.size(), match.data(), &n),
              "snk_match_project_coarse");
        match.resize(lm.size());
        return n;
    }
    int fine(const snk_frame_view* v, const snk_camera& K, const double pose[7], std::vector<snk_lm_fine>& lm, float th, float ratio,
             const std::vector<float>& level_scale, std::vector<int32_t>& match, std::vector<uint8_t>& visible)
    {
        match.assign(lm.size() + 1, -1);
        visible.assign(lm.size() + 1, 0);
        int n = 0;
        check(snk_match_project_fine(h_, v, &K, pose, lm.data(), (int)lm.size(), th, ratio, level_scale.data(),
                                     (int)level_scale.size(), match.data(), visible.data(), &n),
              "snk_match_project_fine");
        match.resize(lm.size());
        visible.resize(lm.size());
        return n;
    }
    snk_matcher* h_ = nullptr;
    BoundFrame bound_{};
    uint64_t bind_counter_ = 0;
};

// Snake::MappingORBMatcher (reference Snake/LocalMapping/MappingORBMatcher.h:15-45): the two keyframe-rate
// matchers that need no bag-of-words.  MapPoint / Keyframe pointers stay on the Snake side.
class MappingORBMatcher
{
   public:
    explicit MappingORBMatcher(int device = 0) { check(snk_matcher_create(device, nullptr, &h_), "snk_matcher_create"); }
    ~MappingORBMatcher() { snk_matcher_destroy(h_); }
    MappingORBMatcher(const MappingORBMatcher&)            = delete;
    MappingORBMatcher& operator=(const MappingORBMatcher&) = delete;

    // Fuse(kf, pose, point_mask, LocalMap<FusionPoint>, fuseCandidates, th, obs_factor, feature_th)
    // — MappingORBMatcher.cpp:359-480; kf_frame = kf->frame as a view, point_mask may be empty (= nullptr)
    int Fuse(const FrameView& kf_frame, const snk_camera& K, const double pose[7], const std::vector<uint8_t>& point_mask,
             const std::vector<snk_fusion_point>& points, std::vector<std::pair<int, int>>& fuseCandidates, float th,
             float obs_factor, int feature_th, const std::vector<float>& level_scale)
    {
        fuseCandidates.clear();
        if (!point_mask.empty() && point_mask.size() != points.size()) throw std::invalid_argument("point_mask size");
        const snk_frame_view v = kf_frame.view();
        std::vector<int32_t> best(points.size() + 1, -1);
        int n = 0;
        check(snk_match_fuse(h_, &v, &K, pose, points.data(), point_mask.empty() ? nullptr : point_mask.data(), (int)points.size(),
                             th, obs_factor, feature_th, level_scale.data(), (int)level_scale.size(), best.data(), &n),
              "snk_match_fuse");
        for (size_t i = 0; i < points.size(); ++i)
            if (best[i] >= 0) fuseCandidates.emplace_back(best[i], points[i].id);
        return n;
    }
    // SearchForTriangulationProject(grid, pose1, pose2, kf1, kf2, E12, vMatchedPairs, epipolarDistance,
    // featureDistance) — MappingORBMatcher.cpp:168-249; pairs are APPENDED like the reference does
    int SearchForTriangulationProject(const double* grid, int grid_rows, int grid_cols, const double pose1[7],
                                      const double pose2[7], const snk_camera& K, const std::vector<snk_kp64>& keypoints1,
                                      const std::vector<std::array<double, 2>>& normalized1,
                                      const std::vector<DescriptorORB>& descriptors1, const std::vector<uint8_t>& has_mp1,
                                      const FrameView& kf2_frame, const std::vector<std::array<double, 2>>& normalized2,
                                      const double E12[9], std::vector<std::pair<int, int>>& vMatchedPairs, float epipolarDistance,
                                      int featureDistance)
    {
        const snk_frame_view v = kf2_frame.view();
        std::vector<int32_t> match(keypoints1.size() + 1, -1);
        int n = 0;
        check(snk_match_triangulation_project(h_, grid, grid_rows, grid_cols, pose1, pose2, &K, keypoints1.data(),
                                              reinterpret_cast<const double(*)[2]>(normalized1.data()),
                                              reinterpret_cast<const uint64_t(*)[4]>(descriptors1.data()), has_mp1.data(),
                                              (int)keypoints1.size(), &v, reinterpret_cast<const double(*)[2]>(normalized2.data()),
                                              E12, epipolarDistance, featureDistance, match.data(), &n),
              "snk_match_triangulation_project");
        for (size_t i = 0; i < keypoints1.size(); ++i)
            if (match[i] >= 0) vMatchedPairs.emplace_back((int)i, match[i]);
        return n;
    }

    // frame->bow_feature_vec (std::map<node id, std::vector<feature index>>-like, ordered) flattened for the C ABI
    struct BowFeatureVector
    {
        std::vector<uint32_t> node_id;
        std::vector<int32_t> node_start{0}, features;
        template <typename Map>
        static BowFeatureVector from(const Map& fv)
        {
            BowFeatureVector b;
            for (const auto& node : fv)
            {
                b.node_id.push_back((uint32_t)node.first);
                for (auto f : node.second) b.features.push_back((int32_t)f);
                b.node_start.push_back((int32_t)b.features.size());
            }
            return b;
        }
        snk_bow_features view() const { return {(int32_t)node_id.size(), 0, node_id.data(), node_start.data(), features.data()}; }
    };
    // SearchForTriangulation2(kf1, kf2, E, vMatchedPairs, epipolarDistance, featureDistance) — MappingORBMatcher.cpp:14-99;
    // pairs are APPENDED in the reference's order
    int SearchForTriangulation2(const snk_camera& K, const double E[9], const std::vector<std::array<double, 2>>& normalized1,
                                const std::vector<DescriptorORB>& descriptors1, const std::vector<uint8_t>& has_mp1,
                                const BowFeatureVector& bow1, const std::vector<std::array<double, 2>>& normalized2,
                                const std::vector<DescriptorORB>& descriptors2, const std::vector<uint8_t>& has_mp2,
                                const BowFeatureVector& bow2, std::vector<std::pair<int, int>>& vMatchedPairs,
                                float epipolarDistance, int featureDistance)
    {
        const snk_bow_features b1 = bow1.view(), b2 = bow2.view();
        std::vector<std::array<int32_t, 2>> pairs(bow1.features.size() + 1);
        int n = 0;
        check(snk_match_triangulation_bow(h_, &K, E, reinterpret_cast<const double(*)[2]>(normalized1.data()),
                                          reinterpret_cast<const uint64_t(*)[4]>(descriptors1.data()), has_mp1.data(),
                                          (int)normalized1.size(), &b1, reinterpret_cast<const double(*)[2]>(normalized2.data()),
                                          reinterpret_cast<const uint64_t(*)[4]>(descriptors2.data()), has_mp2.data(),
                                          (int)normalized2.size(), &b2, epipolarDistance, featureDistance,
                                          reinterpret_cast<int32_t(*)[2]>(pairs.data()), &n),
              "snk_match_triangulation_bow");
        for (int i = 0; i < n; ++i) vMatchedPairs.emplace_back(pairs[i][0], pairs[i][1]);
        return n;
    }
    // SearchForTriangulationBF(pose1, pose2, kf1, kf2, E12, vMatchedPairs, epipolarDistance, featureDistance) —
    // MappingORBMatcher.cpp:102-165 (poses and epipolarDistance are unused there)
    int SearchForTriangulationBF(const snk_camera& K, const double E12[9], const std::vector<std::array<double, 2>>& normalized1,
                                 const std::vector<DescriptorORB>& descriptors1, const std::vector<uint8_t>& has_mp1,
                                 const std::vector<std::array<double, 2>>& normalized2,
                                 const std::vector<DescriptorORB>& descriptors2, const std::vector<uint8_t>& has_mp2,
                                 std::vector<std::pair<int, int>>& vMatchedPairs, int featureDistance)
    {
        std::vector<int32_t> match(normalized1.size() + 1, -1);
        int n = 0;
        check(snk_match_triangulation_bf(h_, &K, E12, reinterpret_cast<const double(*)[2]>(normalized1.data()),
                                         reinterpret_cast<const uint64_t(*)[4]>(descriptors1.data()), has_mp1.data(),
                                         (int)normalized1.size(), reinterpret_cast<const double(*)[2]>(normalized2.data()),
                                         reinterpret_cast<const uint64_t(*)[4]>(descriptors2.data()), has_mp2.data(),
                                         (int)normalized2.size(), featureDistance, match.data(), &n),
              "snk_match_triangulation_bf");
        for (size_t i = 0; i < normalized1.size(); ++i)
            if (match[i] >= 0) vMatchedPairs.emplace_back((int)i, match[i]);
        return n;
    }

   private:
    snk_matcher* h_ = nullptr;
};

// The per-observation search of Snake::DeferredMapper::Relink (reference Snake/Optimizer/DeferredMapper.cpp:39-165).
// The caller builds one query per feature i with a good map point (:61-64: position, descriptor, the descriptor of the
// point's first observation in another keyframe), calls RelinkSearch once, then applies the actions in feature order
// under map.LockFull() exactly as :75-163 do (ERASE / MOVE with the live GetMapPoint(best_idx) test).
class DeferredMapper
{
   public:
    static constexpr float relink_reprojection_error_threshold = 0.8f;  // :41
    static constexpr double relink_outlier_threshold           = 2.1;   // :42 reprojectionErrorThresholdMono
    static constexpr int relink_feature_threshold              = 25;    // :43
    explicit DeferredMapper(int device = 0) { check(snk_matcher_create(device, nullptr, &h_), "snk_matcher_create"); }
    ~DeferredMapper() { snk_matcher_destroy(h_); }
    DeferredMapper(const DeferredMapper&)            = delete;
    DeferredMapper& operator=(const DeferredMapper&) = delete;
    // returns the number of observations whose action is not SNK_RELINK_KEEP
    int RelinkSearch(const FrameView& kf_frame, const snk_camera& K, const double pose[7], const std::vector<snk_relink_query>& queries,
                     std::vector<int32_t>& action, std::vector<int32_t>& best_idx)
    {
        const snk_frame_view v = kf_frame.view();
        action.assign(queries.size() + 1, 0);
        best_idx.assign(queries.size() + 1, -1);
        int n = 0;
        check(snk_match_relink(h_, &v, &K, pose, queries.data(), (int)queries.size(), relink_reprojection_error_threshold,
                               relink_outlier_threshold, relink_feature_threshold, action.data(), best_idx.data(), &n),
              "snk_match_relink");
        action.resize(queries.size());
        best_idx.resize(queries.size());
        return n;
    }

   private:
    snk_matcher* h_ = nullptr;
};

// Snake::PoseRefinement (reference Snake/Tracking/PoseRefinement.h:22-99): the robust pose-only
// optimisation after every matcher call.  The caller gathers wps / obs / idx exactly as refinePose
// (:35-60) and RefinePoseWithMatches (PoseRefinement.cpp:37-57) do, then writes outlier[i] to
// frame.mvbOutlier[idx[i]] and the pose to frame.setPose().
class PoseRefinement
{
   public:
    explicit PoseRefinement(double errorFactor = 1.0, int device = 0)
    {
        check(snk_matcher_create(device, nullptr, &h_), "snk_matcher_create");
        options.th_mono          = 2.1 * errorFactor;  // reprojectionErrorThresholdMono, SnakeGlobal.h:145
        options.th_stereo        = 2.3 * errorFactor;  // reprojectionErrorThresholdStereo, SnakeGlobal.h:146
        options.outer_iterations = 4;
        options.inner_iterations = 10;
        options.robust_rounds    = 3;
        options.pad              = 0;
        options.lambda           = 1e-4;
    }
    ~PoseRefinement() { snk_matcher_destroy(h_); }
    PoseRefinement(const PoseRefinement&)            = delete;
    PoseRefinement& operator=(const PoseRefinement&) = delete;

    // rpo.optimizePoseRobust(wps, obs, outlier, pose, cam); with prediction weights > 0 the
    // rpo_smooth variant (PoseRefinement.h:68-73).  Returns the inlier count.
    int optimizePoseRobust(const std::vector<std::array<double, 3>>& wps, const std::vector<snk_pose_obs>& obs,
                           std::vector<uint8_t>& outlier, double pose[7], const snk_camera& cam,
                           const double* prediction = nullptr, double weight_rotation = 0, double weight_translation = 0)
    {
        outlier.assign(obs.size() + 1, 0);
        snk_pose_problem P{};
        P.n       = (int)obs.size();
        P.wps     = reinterpret_cast<const double(*)[3]>(wps.data());
        P.obs     = obs.data();
        P.outlier = outlier.data();
        for (int i = 0; i < 7; ++i) P.pose[i] = pose[i], P.prediction[i] = prediction ? prediction[i] : pose[i];
        P.w_rot   = prediction ? weight_rotation : 0;
        P.w_trans = prediction ? weight_translation : 0;
        check(snk_pose_refine(h_, &cam, &options, &P, 1), "snk_pose_refine");
        for (int i = 0; i < 7; ++i) pose[i] = P.pose[i];
        outlier.resize(obs.size());
        return P.inliers;
    }
    // a whole batch of frames in one launch
    void optimizeBatch(std::vector<snk_pose_problem>& problems, const snk_camera& cam)
    {
        check(snk_pose_refine(h_, &cam, &options, problems.data(), (int)problems.size()), "snk_pose_refine");
    }

    snk_pose_options options{};

   private:
    snk_matcher* h_ = nullptr;
};

// The part of Saiga::Scene that MakeLocalScene fills (LocalBundleAdjustment.cpp:187-293), flattened.
struct Scene
{
    std::vector<std::array<double, 7>> poses;  // images[i].se3: qx qy qz qw tx ty tz
    std::vector<uint8_t> image_constant;
    std::vector<std::array<double, 3>> points;  // worldPoints[j].p
    std::vector<uint8_t> point_constant;
    std::vector<int32_t> obs_image, obs_point;      // StereoImagePoint owner image, .wp
    std::vector<std::array<double, 2>> obs_pixel;   // .point
    std::vector<double> obs_depth, obs_weight;      // .depth (> 0 => stereo), .weight
    std::vector<uint8_t> obs_outlier;               // .outlier
    double K[4] = {1, 1, 0, 0};                     // intrinsics[0]
    double bf   = 0;
    std::vector<snk_ba_rpc> rel_pose_constraints;   // scene.rel_pose_constraints (IMU), LocalBundleAdjustment.cpp:294-346
};

struct OptimizationResults
{
    double cost_initial = 0, cost_final = 0;
};

// Saiga::BARecRel as Snake drives it — Snake/Optimizer/LocalBundleAdjustment.cpp:357-365,403-407
class BARec
{
   public:
    snk_ba_options optimizationOptions{3, 30, 1e-10, 2.1, 2.3, 0.0};  // LocalBundleAdjustment.cpp:47-64

    explicit BARec(int device = 0) : device_(device) {}
    ~BARec() { snk_ba_destroy(h_); }
    BARec(const BARec&)            = delete;
    BARec& operator=(const BARec&) = delete;

    void create(Scene& scene)
    {
        scene_ = &scene;
        // the handle (device buffers, pinned staging) survives from scene to scene; only changed options need a new one
        if (h_ && std::memcmp(&created_with_, &optimizationOptions, sizeof(snk_ba_options)) != 0)
        {
            snk_ba_destroy(h_);
            h_ = nullptr;
        }
        if (!h_)
        {
            check(snk_ba_create(&optimizationOptions, device_, nullptr, &h_), "snk_ba_create");
            created_with_ = optimizationOptions;
        }
        snk_ba_problem p{};
        p.n_img      = (int)scene.poses.size();
        p.n_pt       = (int)scene.points.size();
        p.n_obs      = (int)scene.obs_image.size();
        p.pose       = reinterpret_cast<double(*)[7]>(scene.poses.data());
        p.pt         = reinterpret_cast<double(*)[3]>(scene.points.data());
        // BAPointOnly holds every image, BAPoseOnly every point (GlobalBundleAdjustment.cpp:103-122, 306-316)
        all_const_.assign(std::max(scene.poses.size(), scene.points.size()) + 1, 1);
        p.img_const  = hold_images_ ? all_const_.data() : scene.image_constant.data();
        p.pt_const   = hold_points_ ? all_const_.data() : scene.point_constant.data();
        p.obs_img    = scene.obs_image.data();
        p.obs_pt     = scene.obs_point.data();
        p.obs_uv     = reinterpret_cast<const double(*)[2]>(scene.obs_pixel.data());
        p.obs_depth  = scene.obs_depth.data();
        p.obs_weight = scene.obs_weight.data();
        for (int k = 0; k < 4; ++k) p.K[k] = scene.K[k];
        p.bf    = scene.bf;
        p.n_rpc = (int)scene.rel_pose_constraints.size();
        p.rpc   = scene.rel_pose_constraints.data();
        check(snk_ba_set_problem(h_, &p), "snk_ba_set_problem");
    }
    OptimizationResults initAndSolve() { return solve(); }
    OptimizationResults solve()
    {
        if (!scene_->obs_outlier.empty()) check(snk_ba_set_outliers(h_, 0, scene_->obs_outlier.data()), "snk_ba_set_outliers");
        OptimizationResults r;
        check(snk_ba_solve(h_, optimizationOptions.max_iterations, &r.cost_initial, &r.cost_final), "snk_ba_solve");
        // the reference mutates the scene in place
        check(snk_ba_get_state(h_, 0, reinterpret_cast<double(*)[7]>(scene_->poses.data()),
                               reinterpret_cast<double(*)[3]>(scene_->points.data()), nullptr),
              "snk_ba_get_state");
        return r;
    }
    // LocalBundleAdjustment::SolveLocalScene after create() in one library call (LocalBundleAdjustment.cpp:357-410): initAndSolve,
    // the chi-square pass on the device, one more iteration when anything was marked.  The scene's poses, points and
    // obs_outlier are updated; *outlierPoints = how many observations the pass marked; returns the FIRST solve's costs (:412).
    OptimizationResults solveLocalScene(double chi2Mono, double chi2Stereo, int* outlierPoints)
    {
        if (scene_->obs_outlier.size() != scene_->obs_image.size()) scene_->obs_outlier.assign(scene_->obs_image.size(), 0);
        bool any = false;
        for (uint8_t f : scene_->obs_outlier) any |= f != 0;
        check(snk_ba_set_outliers(h_, 0, any ? scene_->obs_outlier.data() : nullptr), "snk_ba_set_outliers");  // NULL: a device memset
        OptimizationResults r;
        int marked = 0;
        scene_->obs_outlier.push_back(0);  // never hand out a null pointer for an empty scene
        check(snk_ba_solve_local_scene(h_, 0, chi2Mono, chi2Stereo, 1, scene_->obs_outlier.data(), &marked, &r.cost_initial, &r.cost_final,
                                       reinterpret_cast<double(*)[7]>(scene_->poses.data()),
                                       reinterpret_cast<double(*)[3]>(scene_->points.data())),
              "snk_ba_solve_local_scene");
        scene_->obs_outlier.pop_back();
        if (outlierPoints) *outlierPoints = marked;
        return r;
    }
    // squared norms of Scene::residual3 / residual2 for every observation
    std::vector<double> residualsSquared()
    {
        std::vector<double> chi2(scene_->obs_image.size() + 1);
        check(snk_ba_residuals(h_, 0, chi2.data()), "snk_ba_residuals");
        chi2.resize(scene_->obs_image.size());
        return chi2;
    }

   protected:
    bool hold_images_ = false, hold_points_ = false;

   private:
    int device_;
    snk_ba* h_    = nullptr;
    Scene* scene_ = nullptr;
    snk_ba_options created_with_{};
    std::vector<uint8_t> all_const_;
};

// Saiga::BAPointOnly as GlobalBundleAdjustment::PointBA uses it (GlobalBundleAdjustment.cpp:103-122): create(scene),
// initAndSolve() -- world points optimised, every camera held.  [DEFINED]: the BARec iteration with every image constant.
class BAPointOnly : public BARec
{
   public:
    explicit BAPointOnly(int device = 0) : BARec(device)
    {
        optimizationOptions = {4, 40, 1e-10, 2.1, 2.3, 0.0};  // GlobalBundleAdjustment.cpp:32-43
        hold_images_        = true;
    }
};

// Saiga::BAPoseOnly as RealignIntermiediateFrames uses it (GlobalBundleAdjustment.cpp:306-316): camera poses optimised, every
// world point held.  [DEFINED]: the BARec iteration with every point constant.
class BAPoseOnly : public BARec
{
   public:
    explicit BAPoseOnly(int device = 0) : BARec(device)
    {
        optimizationOptions = {4, 40, 1e-10, 2.1, 2.3, 0.0};
        hold_points_        = true;
    }
};

// ------------------------------------------------------------------------------------------------
// The `.features` cache either side of the extractor — reference
// Snake/Preprocess/FeatureDetector.cpp:94-111 (read) and :134-139 / :166-171 (write):
//   BinaryFile << std::vector<Saiga::KeyPoint<double>> << std::vector<DescriptorORB>
// Saiga::BinaryFile (absent submodule) streams a vector as its element count followed by the raw
// elements.  ASSUMED layout (unverified against a file written by a real Snake-SLAM build): count as
// 64-bit little-endian size_t; KeyPoint<double> = {Vec2d point; double size, angle, response;
// int octave;} padded to 48 bytes; DescriptorORB = 32 bytes.
// ------------------------------------------------------------------------------------------------
struct KeyPointD
{
    double x, y;
    double size, angle, response;
    int32_t octave;
    int32_t pad;
};
static_assert(sizeof(KeyPointD) == 48, "KeyPoint<double> layout");

inline void WriteFeatures(const std::string& file, const std::vector<KeyPointD>& keypoints,
                          const std::vector<DescriptorORB>& descriptors)
{
    // The layout is an unverified assumption (Saiga::BinaryFile is absent): a cache written here may be silently misread by a
    // real Snake-SLAM build.  Refuse unless the caller acknowledges that (or has pinned the layout with tools/check_features_dir.py).
    const char* ack = std::getenv("SNK_FEATURES_LAYOUT_ACK");
    if (!ack || std::string(ack) != "1")
        throw std::runtime_error("WriteFeatures: the .features layout is an unverified assumption; set SNK_FEATURES_LAYOUT_ACK=1 to write it anyway");
    FILE* f = std::fopen(file.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot open " + file);
    const uint64_t nk = keypoints.size(), nd = descriptors.size();
    bool ok = std::fwrite(&nk, 8, 1, f) == 1 && (nk == 0 || std::fwrite(keypoints.data(), sizeof(KeyPointD), nk, f) == nk) &&
              std::fwrite(&nd, 8, 1, f) == 1 && (nd == 0 || std::fwrite(descriptors.data(), 32, nd, f) == nd);
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) throw std::runtime_error("short write to " + file);
}

inline void ReadFeatures(const std::string& file, std::vector<KeyPointD>& keypoints, std::vector<DescriptorORB>& descriptors)
{
    FILE* f = std::fopen(file.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + file);
    auto fail = [&](const char* what) {
        std::fclose(f);
        throw std::runtime_error(std::string(what) + " in " + file);
    };
    uint64_t nk = 0, nd = 0;
    if (std::fread(&nk, 8, 1, f) != 1 || nk > (1u << 24)) fail("bad keypoint count");
    keypoints.resize(nk);
    if (nk && std::fread(keypoints.data(), sizeof(KeyPointD), nk, f) != nk) fail("truncated keypoints");
    if (std::fread(&nd, 8, 1, f) != 1 || nd > (1u << 24)) fail("bad descriptor count");
    descriptors.resize(nd);
    if (nd && std::fread(descriptors.data(), 32, nd, f) != nd) fail("truncated descriptors");
    std::fclose(f);
}

// A file of unknown provenance (a real Snake-SLAM build: saiga's BinaryFile layout is not known here): tries the plausible
// layouts -- 64 / 32-bit counts; KeyPoint<double> of 48 bytes, packed 44, KeyPoint<float> of 24 -- and keeps the first one
// that accounts for every byte of the file (same probing as snake_slam_amd/features_io.py::probe_layout).  Returns the layout
// as "count bytes / keypoint bytes", e.g. "8/48".
inline std::string ReadFeaturesAny(const std::string& file, std::vector<KeyPointD>& keypoints, std::vector<DescriptorORB>& descriptors)
{
    FILE* f = std::fopen(file.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + file);
    std::vector<unsigned char> buf;
    unsigned char tmp[65536];
    size_t got;
    while ((got = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
    std::fclose(f);
    auto rd = [&](size_t off, int bytes) {
        uint64_t v = 0;
        for (int i = 0; i < bytes; ++i) v |= (uint64_t)buf[off + (size_t)i] << (8 * i);
        return v;
    };
    for (int cw : {8, 4})
        for (int ks : {48, 44, 24})
        {
            if (buf.size() < 2 * (size_t)cw) continue;
            const uint64_t nk = rd(0, cw);
            const size_t off  = (size_t)cw + (size_t)nk * (size_t)ks;
            if (nk > (1u << 24) || off + (size_t)cw > buf.size()) continue;
            const uint64_t nd = rd(off, cw);
            if (nd != nk || off + (size_t)cw + (size_t)nd * 32 != buf.size()) continue;
            keypoints.resize(nk);
            descriptors.resize(nd);
            for (uint64_t i = 0; i < nk; ++i)
            {
                const unsigned char* p = buf.data() + cw + i * (size_t)ks;
                KeyPointD k{};
                if (ks == 24)
                {
                    float v[5];
                    std::memcpy(v, p, 20);
                    std::memcpy(&k.octave, p + 20, 4);
                    k.x = v[0], k.y = v[1], k.size = v[2], k.angle = v[3], k.response = v[4];
                }
                else
                {
                    double v[5];
                    std::memcpy(v, p, 40);
                    std::memcpy(&k.octave, p + 40, 4);
                    k.x = v[0], k.y = v[1], k.size = v[2], k.angle = v[3], k.response = v[4];
                }
                keypoints[i] = k;
            }
            if (nd) std::memcpy(descriptors.data(), buf.data() + off + cw, (size_t)nd * 32);
            return std::to_string(cw) + "/" + std::to_string(ks);
        }
    throw std::runtime_error("no known layout accounts for the size of " + file);
}

// frame.keypoints.emplace_back(kp.cast<double>()) — FeatureDetector.cpp:128-131
inline KeyPointD cast_double(const snk_keypoint& k)
{
    return KeyPointD{(double)k.x, (double)k.y, (double)k.size, (double)k.angle, (double)k.response, k.octave, 0};
}
// ------------------------------------------------------------------------------------------------
// One Snake-SLAM process per GPU (BASELINE config 5): gather every rank's results with RCCL over xGMI, no MPI / torch in the process.
// A rank's block is what System::writeFrameTrajectory writes for its sequence (Snake/System/System.cpp:546-563): one
// {timestamp, tx, ty, tz, qx, qy, qz, qw} row per frame with a valid pose.
// ------------------------------------------------------------------------------------------------
struct TumPose
{
    double t, tx, ty, tz, qx, qy, qz, qw;
};

class Dist
{
   public:
    // rendezvous through a file all ranks see (snk_dist_init_file): a fresh path per job
    Dist(const std::string& rendezvous_file, int rank, int world, int device, double timeout_s = 60.0)
    {
        check(snk_dist_init_file(rendezvous_file.c_str(), rank, world, device, timeout_s, &h_), "snk_dist_init_file");
        rank_ = rank;
        world_ = world;
        if (rank == 0) file_ = rendezvous_file;
    }
    // the 128-byte id was handed around by the launcher
    Dist(const uint8_t id[SNK_DIST_ID_BYTES], int rank, int world, int device)
    {
        check(snk_dist_init(id, rank, world, device, &h_), "snk_dist_init");
        rank_ = rank;
        world_ = world;
    }
    ~Dist()
    {
        snk_dist_destroy(h_);
        if (!file_.empty()) std::remove(file_.c_str());
    }
    Dist(const Dist&)            = delete;
    Dist& operator=(const Dist&) = delete;
    int rank() const { return rank_; }
    snk_dist* handle() const { return h_; }  // for snk_dist_rank (what the communicator itself reports)
    int world() const { return world_; }

    // all ranks' trajectories on every rank: result[r] = rank r's rows.  Two collectives: the longest trajectory (blocks are padded to
    // it), then ONE all-gather of {n, rows} blocks -- the exchange SURVEY.md section 8e describes (~236 KB per rank for MH_01).
    std::vector<std::vector<TumPose>> GatherTrajectories(const std::vector<TumPose>& mine)
    {
        int64_t longest = 0;
        check(snk_dist_max_i64(h_, (int64_t)mine.size(), &longest), "snk_dist_max_i64");
        const size_t block = 8 + (size_t)longest * sizeof(TumPose);  // {int64 n, rows}
        std::vector<uint8_t> send(block, 0), recv(block * (size_t)world_);
        const int64_t n = (int64_t)mine.size();
        std::memcpy(send.data(), &n, 8);
        if (n) std::memcpy(send.data() + 8, mine.data(), (size_t)n * sizeof(TumPose));
        check(snk_dist_all_gather(h_, send.data(), block, recv.data()), "snk_dist_all_gather");
        std::vector<std::vector<TumPose>> all((size_t)world_);
        for (int r = 0; r < world_; ++r)
        {
            int64_t nr = 0;
            std::memcpy(&nr, recv.data() + (size_t)r * block, 8);
            all[(size_t)r].resize((size_t)nr);
            if (nr) std::memcpy(all[(size_t)r].data(), recv.data() + (size_t)r * block + 8, (size_t)nr * sizeof(TumPose));
        }
        return all;
    }
    // any fixed-size block per rank (counters: frames, frames/s, matches, BA costs)
    template <typename T>
    std::vector<T> GatherBlocks(const T& mine)
    {
        static_assert(std::is_trivially_copyable<T>::value, "plain data only");
        std::vector<T> all((size_t)world_);
        check(snk_dist_all_gather(h_, &mine, sizeof(T), all.data()), "snk_dist_all_gather");
        return all;
    }

   private:
    snk_dist* h_ = nullptr;
    int rank_ = 0, world_ = 1;
    std::string file_;
};
}  // namespace snake_hip
