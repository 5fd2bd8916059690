// snake_hip_reference.hpp — the hot path behind the REFERENCE'S OWN call signatures.
//
// snake_hip.hpp mirrors the saiga classes with flat arrays; this header goes one step further for the seams whose types are
// Snake-SLAM's own: the functions below are templates over the Snake types and read exactly the member names of
//   Snake/Map/Frame.h:20-46, Snake/Map/Features.h:18-41, Snake/Map/LocalMap.h:17-80, Snake/Map/Keyframe.h (GetMapPointMatches),
//   Saiga::Scene as MakeLocalScene fills it (Snake/Optimizer/LocalBundleAdjustment.cpp:187-346),
// so a Snake-SLAM build calls
//   SearchByProjectionFrameFrame2(Frame&, const LocalMap<CoarseTrackingPoint>&, float th, FeatureDistance, int num_threads)
//   SearchByProjection2(Frame&, LocalMap<FineTrackingPoint>&, float th, float ratio, int num_threads)
//   SearchByProjectionFrameToKeyframe(Frame&, const Keyframe&, float th, FeatureDistance)      (Snake/Tracking/SnakeORBMatcher.h:21-30)
//   undistortKeypoints(Frame&), computeFeatureGrid(Frame&), StereoMatching(Frame&)             (Snake/Preprocess/Preprocess.h:32-34)
//   SolveLocalScene on the Saiga::Scene                                                          (LocalBundleAdjustment.cpp:353-413)
// with its own objects and gets the side effects the reference functions have (mvpMapPoints[idx] = lmp.mp, lmp.valid,
// mp->IncreaseVisible(), right_points / depth, the permuted feature arrays, o.outlier, the scene's poses / points).
// No saiga / Eigen / Sophus header is needed to compile this file: tests/cpp/reference_shims_driver.cpp instantiates every
// template with mock structs that carry exactly those member names.
//
// What the reference keeps in globals (System/SnakeGlobal.h: K, stereo_cam, featureGridBounds, scalePyramid, settings.inputType,
// rect_left / rect_right) is handed over once in a `Globals`.
#pragma once
#include <cstring>
#include <tuple>
#include <type_traits>

#include "snake_hip.hpp"

namespace snake_hip
{
namespace ref
{
struct Globals
{
    snk_camera K{};                       // K (fx fy cx cy) and stereo_cam.bf
    double baseline = 0;                  // stereo_cam.baseLine()
    snk_grid_bounds featureGridBounds{};  // featureGridBounds
    std::vector<float> level_scale;       // scalePyramid.Scale(l), l = 0 .. levels - 1
    bool mono           = false;          // settings.inputType == InputType::Mono
    bool rgbd           = false;          // settings.inputType == InputType::RGBD: Preprocess::Process calls ComputeStereoFromRGBD, not StereoMatching (FrontEnd then needs mono = true)
    bool relaxed_stereo = true;           // settings.fd_relaxed_stereo (Settings.h:123)
    snk_rectification rect_left{}, rect_right{};
};

namespace detail
{
// Sophus::SE3d as Snake uses it: unit_quaternion() (Eigen::Quaterniond: x() y() z() w()) and translation() (Vec3, operator()(int))
template <class SE3>
inline void pose7(const SE3& T, double p[7])
{
    const auto& q = T.unit_quaternion();
    const auto& t = T.translation();
    p[0] = q.x(), p[1] = q.y(), p[2] = q.z(), p[3] = q.w();
    p[4] = t(0), p[5] = t(1), p[6] = t(2);
}
template <class D>
inline void desc32(const D& d, uint64_t out[4])
{
    static_assert(sizeof(D) == 32, "FeatureDescriptor must be the 256-bit Saiga::DescriptorORB");
    std::memcpy(out, &d, 32);
}
template <class KP>
inline snk_kp64 kp64(const KP& k)
{
    return snk_kp64{(double)k.point(0), (double)k.point(1), (float)k.angle, (int32_t)k.octave};
}
}  // namespace detail

// Saiga::FeatureGrid2 as Snake uses it (Snake/Map/Features.h:40, Preprocess.cpp:246): `grid.create(bounds, undistorted_keypoints)`
// returns the permutation that makes cell members contiguous.  The member a Snake build declares instead of Saiga::FeatureGrid2.
class FeatureGrid2
{
   public:
    template <class KeyPoints>
    std::vector<int> create(SnakeORBMatcher& m, const snk_grid_bounds& bounds, const KeyPoints& undistorted_keypoints)
    {
        FrameView fv;
        fv.undistorted_keypoints.reserve(undistorted_keypoints.size());
        for (const auto& k : undistorted_keypoints) fv.undistorted_keypoints.push_back(detail::kp64(k));
        const std::vector<int32_t> perm = m.CreateGrid(fv, bounds);
        cell_start = std::move(fv.cell_start);
        cols = fv.cols, rows = fv.rows, this->bounds = bounds;
        return std::vector<int>(perm.begin(), perm.end());
    }
    std::vector<int32_t> cell_start;
    int cols = 0, rows = 0;
    snk_grid_bounds bounds{};
};

// Snake::Preprocess — Snake/Preprocess/Preprocess.cpp:55-77 (undistortKeypoints), :244-266 (computeFeatureGrid), :122-242
// (StereoMatching).  Frame::allocateTmp (Frame.cpp:21-29) stays the caller's: right_points / depth arrive filled with -1000.
class Preprocess
{
   public:
    explicit Preprocess(const Globals& g, int device = 0) : g_(g), pre_(device), grid_(device) {}

    template <class Frame>
    void undistortKeypoints(Frame& frame)
    {
        std::vector<KeyPointF> in((size_t)frame.N);
        for (int i = 0; i < frame.N; ++i)
        {
            const auto& k = frame.keypoints[(size_t)i];
            in[(size_t)i] = KeyPointF{(float)k.point(0), (float)k.point(1), (float)k.size, (float)k.angle, (float)k.response, (int32_t)k.octave};
        }
        std::vector<snk_kp64> out;
        std::vector<std::array<double, 2>> norm;
        pre_.Rectify(g_.rect_left, in, out, &norm);
        frame.undistorted_keypoints.clear();
        for (int i = 0; i < frame.N; ++i)
        {
            frame.undistorted_keypoints.emplace_back(frame.keypoints[(size_t)i]);                         // :60
            frame.normalized_points[(size_t)i](0) = norm[(size_t)i][0];                                   // :73
            frame.normalized_points[(size_t)i](1) = norm[(size_t)i][1];
            frame.undistorted_keypoints[(size_t)i].point(0) = out[(size_t)i].x;                           // :75
            frame.undistorted_keypoints[(size_t)i].point(1) = out[(size_t)i].y;
        }
    }

    template <class Frame>
    void computeFeatureGrid(Frame& frame)
    {
        const auto permutation = frame.grid.create(grid_, g_.featureGridBounds, frame.undistorted_keypoints);  // :246
        const size_t N         = permutation.size();
        auto mvKeys2           = frame.keypoints;
        auto descriptors2      = frame.descriptors;
        auto mvKeysUn2         = frame.undistorted_keypoints;
        auto norm2             = frame.normalized_points;
        for (size_t i = 0; i < N; ++i)  // :254-260
        {
            const size_t p  = (size_t)permutation[i];
            mvKeys2[p]      = frame.keypoints[i];
            descriptors2[p] = frame.descriptors[i];
            mvKeysUn2[p]    = frame.undistorted_keypoints[i];
            norm2[p]        = frame.normalized_points[i];
        }
        frame.keypoints.swap(mvKeys2);
        frame.descriptors.swap(descriptors2);
        frame.undistorted_keypoints.swap(mvKeysUn2);
        frame.normalized_points.swap(norm2);
    }

    template <class Frame>
    int StereoMatching(Frame& frame)
    {
        auto rectify = [&](const auto& kps, const snk_rectification& rect)  // rect.Forward, :140-150
        {
            std::vector<KeyPointF> in(kps.size());
            for (size_t i = 0; i < kps.size(); ++i)
                in[i] = KeyPointF{(float)kps[i].point(0), (float)kps[i].point(1), (float)kps[i].size, (float)kps[i].angle,
                                  (float)kps[i].response, (int32_t)kps[i].octave};
            std::vector<snk_kp64> out;
            pre_.Rectify(rect, in, out, nullptr);
            // Forward is evaluated on the double-precision point: the float round trip above must not lose it
            for (size_t i = 0; i < kps.size(); ++i)
                if ((double)in[i].x != (double)kps[i].point(0) || (double)in[i].y != (double)kps[i].point(1)) exact_ = false;
            return out;
        };
        exact_                 = true;
        const auto left        = rectify(frame.keypoints, g_.rect_left);
        const auto right       = rectify(frame.keypoints_right, g_.rect_right);
        auto descs             = [](const auto& d)
        {
            std::vector<DescriptorORB> o(d.size());
            for (size_t i = 0; i < d.size(); ++i) detail::desc32(d[i], o[i].data());
            return o;
        };
        std::vector<float> rp(frame.right_points.begin(), frame.right_points.end()), dp(frame.depth.begin(), frame.depth.end());
        const int n = pre_.StereoMatching(left, descs(frame.descriptors), right, descs(frame.descriptors_right), g_.rect_left.bf,
                                          g_.level_scale, g_.relaxed_stereo, rp, dp);
        for (size_t i = 0; i < rp.size(); ++i) frame.right_points[i] = rp[i], frame.depth[i] = dp[i];  // :235-236
        return n;
    }
    // ComputeStereoFromRGBD(Frame&) -- Preprocess.cpp:79-120, the RGB-D branch of Preprocess::Process.  frame.depth_image is read
    // through the members of Saiga::ImageView<float> (`data`, `width`, `height`, `pitchBytes`): frame.depth_image.getImageView().
    template <class Frame, class DepthView>
    int ComputeStereoFromRGBD(Frame& frame, const DepthView& depth_image, const snk_rgbd_model& model)
    {
        std::vector<snk_kp64> und;
        und.reserve((size_t)frame.N);
        for (int i = 0; i < frame.N; ++i) und.push_back(detail::kp64(frame.undistorted_keypoints[(size_t)i]));
        std::vector<float> rp, dp;
        const int n = pre_.ComputeStereoFromRGBD(model, und, reinterpret_cast<const float*>(depth_image.data), (int)depth_image.width,
                                                 (int)depth_image.height, (int)(depth_image.pitchBytes / sizeof(float)), rp, dp);
        for (int i = 0; i < frame.N; ++i) frame.right_points[(size_t)i] = rp[(size_t)i], frame.depth[(size_t)i] = dp[(size_t)i];  // :106-114
        return n;
    }
    // false after a StereoMatching whose keypoint coordinates were not exactly representable as float (the extractor's are:
    // kp.cast<double>() of KeyPoint<float>, FeatureDetector.cpp:128-131)
    bool exact() const { return exact_; }

   private:
    const Globals& g_;
    snake_hip::Preprocess pre_;
    SnakeORBMatcher grid_;
    bool exact_ = true;
};

// FeatureDetector::Detect(Frame&) + Preprocess::Process(Frame&) for a stereo (or mono) frame in ONE call and one synchronisation
// (snk_frontend_process): Snake/Preprocess/FeatureDetector.cpp:116-156 (extract left and right, keypoints cast to double, frame.N)
// followed by Snake/Preprocess/Preprocess.cpp:35-53 (allocateTmp, undistortKeypoints, computeFeatureGrid, StereoMatching).  A Snake
// build calls it where FeatureDetector::Detect runs the extractor (the "FeatureDetection" thread) and lets Preprocess::Process
// pass the frame through -- one GPU round trip per frame instead of six.  The images are read through the members of
// Saiga::ImageView<unsigned char> (`data`, `width`, `height`, `pitchBytes`), i.e. frame.image.getImageView() / frame.right_image.getImageView().
class FrontEnd
{
   public:
    FrontEnd(const Globals& g, const snk_orb_params& orb, int device = 0) : g_(g), fe_(make(g, orb), device) {}

    template <class Frame, class ImageViewT>
    int DetectAndProcess(Frame& frame, const ImageViewT& left, const ImageViewT* right)
    {
        const int n_stereo = fe_.Process(reinterpret_cast<const uint8_t*>(left.data), (int)left.pitchBytes,
                                         right ? reinterpret_cast<const uint8_t*>(right->data) : nullptr, right ? (int)right->pitchBytes : 0,
                                         (int)left.width, (int)left.height, r_);
        using KP = typename std::decay<decltype(frame.keypoints[0])>::type;
        auto widen = [](const KeyPointF& k)  // kp.cast<double>() (FeatureDetector.cpp:128-131)
        {
            KP o{};
            o.point(0) = k.x, o.point(1) = k.y, o.size = k.size, o.angle = k.angle, o.response = k.response, o.octave = k.octave;
            return o;
        };
        const size_t N = r_.keypoints.size(), NR = r_.keypoints_right.size();
        frame.keypoints.clear(), frame.keypoints_right.clear();
        for (size_t i = 0; i < N; ++i) frame.keypoints.push_back(widen(r_.keypoints[i]));
        for (size_t i = 0; i < NR; ++i) frame.keypoints_right.push_back(widen(r_.keypoints_right[i]));
        frame.descriptors.resize(N), frame.descriptors_right.resize(NR);
        static_assert(sizeof(frame.descriptors[0]) == 32, "FeatureDescriptor must be the 256-bit Saiga::DescriptorORB");
        if (N) std::memcpy(&frame.descriptors[0], r_.descriptors.data(), N * 32);
        if (NR) std::memcpy(&frame.descriptors_right[0], r_.descriptors_right.data(), NR * 32);
        frame.N = (int)N;      // FeatureDetector.cpp:169
        frame.allocateTmp();   // Preprocess.cpp:40
        frame.undistorted_keypoints.clear();
        for (size_t i = 0; i < N; ++i)
        {
            frame.undistorted_keypoints.emplace_back(frame.keypoints[i]);                    // Preprocess.cpp:60
            frame.undistorted_keypoints[i].point(0) = r_.undistorted_keypoints[i].x;         // :75
            frame.undistorted_keypoints[i].point(1) = r_.undistorted_keypoints[i].y;
            frame.normalized_points[i](0)           = r_.normalized_points[i][0];            // :73
            frame.normalized_points[i](1)           = r_.normalized_points[i][1];
            frame.right_points[i]                   = r_.right_points[i];                    // :235
            frame.depth[i]                          = r_.depth[i];                           // :236
        }
        frame.grid.cell_start = r_.cell_start;  // frame.grid.create (:246)
        frame.grid.cols = r_.cols, frame.grid.rows = r_.rows, frame.grid.bounds = g_.featureGridBounds;
        return n_stereo;
    }

   private:
    static snk_frontend_params make(const Globals& g, const snk_orb_params& orb)
    {
        // Preprocess::Process has three branches (Preprocess.cpp:43-49): RGBD -> ComputeStereoFromRGBD, Stereo -> StereoMatching, Mono -> neither.
        // snk_frontend_process serves the stereo and the mono branch.  An RGBD build would silently run as "stereo without a right image"
        // (an error) or as mono (depth / right_points left at -1000): refuse it here and name the call that serves it.
        if (g.rgbd && !g.mono)
            throw std::invalid_argument("snake_hip_reference::FrontEnd serves the stereo and the mono branch of Preprocess::Process; an InputType::RGBD build "
                                        "constructs it with Globals::mono = true (left image: extraction, undistortion, grid) and then calls "
                                        "Preprocess::ComputeStereoFromRGBD (snk_rgbd_stereo) -- Snake/Preprocess/Preprocess.cpp:43-46,79-120");
        snk_frontend_params p{};
        p.orb = orb, p.rect_left = g.rect_left, p.rect_right = g.rect_right, p.bounds = g.featureGridBounds;
        p.bf = g.rect_left.bf, p.relaxed_stereo = g.relaxed_stereo ? 1 : 0, p.stereo = g.mono ? 0 : 1;
        return p;
    }
    const Globals& g_;
    Frontend fe_;
    FrontendResult r_;
};

// Snake::SnakeORBMatcher with the reference's signatures (Snake/Tracking/SnakeORBMatcher.h:21-30).  FeatureDistance is an int.
class SnakeORBMatcher
{
   public:
    explicit SnakeORBMatcher(const Globals& g, int device = 0) : g_(g), m_(device) {}

    // Coarse tracking — SnakeORBMatcher.cpp:191-354
    template <class Frame, class LocalMapT>
    int SearchByProjectionFrameFrame2(Frame& CurrentFrame, const LocalMapT& lm, const float th, int featureError, int /*num_threads*/)
    {
        double pose[7];
        detail::pose7(CurrentFrame.Pose(), pose);
        // negative because the motion model lives in world -> camera space (:208-212)
        const double z_diff = -CurrentFrame.local_velocity.translation()(2);
        const int direction = g_.mono ? 0 : (z_diff > g_.baseline ? 1 : (z_diff < g_.baseline ? 2 : 0));
        std::vector<snk_lm_coarse> pts(lm.points.size());
        for (size_t i = 0; i < pts.size(); ++i)
        {
            const auto& p = lm.points[i];
            for (int k = 0; k < 3; ++k) pts[i].pos[k] = p.position(k), pts[i].normal[k] = p.normal(k);
            detail::desc32(p.descriptor, pts[i].desc);
            pts[i].octave = p.octave;
            pts[i].angle  = p.angle;
        }
        const FrameView fv = view_of(CurrentFrame);
        std::vector<int32_t> match;
        const int n = m_.SearchByProjectionFrameFrame2(fv, g_.K, pose, pts, th, featureError, direction, g_.level_scale, match);
        for (size_t i = 0; i < match.size(); ++i)
            if (match[i] >= 0) CurrentFrame.mvpMapPoints[(size_t)match[i]] = lm.points[i].mp;  // :330
        return n;
    }

    // Fine tracking — SnakeORBMatcher.cpp:365-526
    template <class Frame, class LocalMapT>
    int SearchByProjection2(Frame& CurrentFrame, LocalMapT& lm, const float th, float ratio, int /*num_threads*/)
    {
        double pose[7];
        detail::pose7(CurrentFrame.Pose(), pose);
        std::vector<snk_lm_fine> pts(lm.points.size());
        for (size_t i = 0; i < pts.size(); ++i)
        {
            const auto& p = lm.points[i];
            for (int k = 0; k < 3; ++k) pts[i].pos[k] = p.position(k), pts[i].normal[k] = p.normal(k);
            detail::desc32(p.descriptor, pts[i].desc);
            pts[i].reference_depth       = p.reference_depth;
            pts[i].reference_scale_level = p.reference_scale_level;
            pts[i].valid                 = p.valid ? 1 : 0;
        }
        const FrameView fv = view_of(CurrentFrame);
        std::vector<int32_t> match;
        std::vector<uint8_t> visible;
        const int n = m_.SearchByProjection2(fv, g_.K, pose, pts, th, ratio, g_.level_scale, match, visible);
        for (size_t i = 0; i < match.size(); ++i)
        {
            lm.points[i].valid = pts[i].valid != 0;                                              // :397,403,415,428
            if (visible[i]) lm.points[i].mp->IncreaseVisible();                                  // :431
            if (match[i] >= 0) CurrentFrame.mvpMapPoints[(size_t)match[i]] = lm.points[i].mp;   // :521
        }
        return n;
    }

    // BF tracking / loop closing — SnakeORBMatcher.cpp:71-188
    template <class Frame, class KeyframeT>
    int SearchByProjectionFrameToKeyframe(Frame& CurrentFrame, const KeyframeT& kf, float th, int featureError)
    {
        double pose[7];
        detail::pose7(CurrentFrame.Pose(), pose);
        const auto& points = kf.GetMapPointMatches();  // :96
        std::vector<std::array<double, 3>> positions(points.size());
        std::vector<DescriptorORB> descriptors(points.size());
        std::vector<uint8_t> skip(points.size(), 0);
        for (size_t i = 0; i < points.size(); ++i)
        {
            auto mp = points[i];
            bool already = false;  // sAlreadyFound (:74-83): the frame's current map points
            if (mp)
                for (int j = 0; j < CurrentFrame.N && !already; ++j) already = CurrentFrame.mvpMapPoints[(size_t)j] == mp;
            if (!mp || already)
            {
                skip[i] = 1;
                continue;
            }
            const auto wp = mp->getPosition();
            for (int k = 0; k < 3; ++k) positions[i][(size_t)k] = wp(k);
            detail::desc32(mp->GetDescriptor(), descriptors[i].data());
        }
        const FrameView fv = view_of(CurrentFrame);
        std::vector<int32_t> match;
        const int n = m_.SearchByProjectionFrameToKeyframe(fv, g_.K, pose, positions, descriptors, skip, th, featureError, match);
        for (size_t i = 0; i < match.size(); ++i)
            if (match[i] >= 0) CurrentFrame.mvpMapPoints[(size_t)match[i]] = points[i];  // :150
        return n;
    }

   private:
    // the frame as the matchers read it: undistorted keypoints, descriptors, right_points, "has a map point", the grid
    template <class Frame>
    FrameView view_of(const Frame& f) const
    {
        FrameView fv;
        const size_t n = (size_t)f.N;
        fv.undistorted_keypoints.resize(n);
        fv.descriptors.resize(n);
        fv.right_points.assign(f.right_points.begin(), f.right_points.begin() + (long)n);
        fv.taken.resize(n);
        for (size_t i = 0; i < n; ++i)
        {
            fv.undistorted_keypoints[i] = detail::kp64(f.undistorted_keypoints[i]);
            detail::desc32(f.descriptors[i], fv.descriptors[i].data());
            fv.taken[i] = f.mvpMapPoints[i] != nullptr;
        }
        fv.cell_start = f.grid.cell_start;
        fv.cols = f.grid.cols, fv.rows = f.grid.rows, fv.bounds = f.grid.bounds;
        return fv;
    }
    const Globals& g_;
    snake_hip::SnakeORBMatcher m_;
};

// ------------------------------------------------------------------------------------------------
// Local bundle adjustment on the Saiga::Scene that MakeLocalScene filled (LocalBundleAdjustment.cpp:187-346).
// flatten reads: scene.images[i].{se3, constant, stereoPoints[k].{point, depth, wp, weight, outlier}},
// scene.worldPoints[j].{p, constant}, scene.intrinsics[0].{fx, fy, cx, cy}, scene.bf,
// scene.rel_pose_constraints[c].{img1, img2, rel_pose, weight_rotation, weight_translation}.
// Observations are emitted image by image in stereoPoints order; `where[o]` = (image, index in stereoPoints).
// ------------------------------------------------------------------------------------------------
template <class SaigaScene>
inline Scene flatten(const SaigaScene& scene, std::vector<std::pair<int, int>>* where = nullptr)
{
    Scene s;
    s.poses.resize(scene.images.size());
    s.image_constant.resize(scene.images.size());
    for (size_t i = 0; i < scene.images.size(); ++i)
    {
        detail::pose7(scene.images[i].se3, s.poses[i].data());
        s.image_constant[i] = scene.images[i].constant ? 1 : 0;
    }
    s.points.resize(scene.worldPoints.size());
    s.point_constant.resize(scene.worldPoints.size());
    for (size_t j = 0; j < scene.worldPoints.size(); ++j)
    {
        for (int k = 0; k < 3; ++k) s.points[j][(size_t)k] = scene.worldPoints[j].p(k);
        s.point_constant[j] = scene.worldPoints[j].constant ? 1 : 0;
    }
    // sized once, written by index (seven push_backs per observation are a tenth of a millisecond for a 16 000-observation window)
    size_t n_obs = 0;
    for (size_t i = 0; i < scene.images.size(); ++i) n_obs += scene.images[i].stereoPoints.size();
    s.obs_image.resize(n_obs), s.obs_point.resize(n_obs), s.obs_pixel.resize(n_obs), s.obs_depth.resize(n_obs);
    s.obs_weight.resize(n_obs), s.obs_outlier.resize(n_obs);
    if (where) where->resize(n_obs);
    size_t at = 0;
    for (size_t i = 0; i < scene.images.size(); ++i)
        for (size_t k = 0; k < scene.images[i].stereoPoints.size(); ++k, ++at)
        {
            const auto& o    = scene.images[i].stereoPoints[k];
            s.obs_image[at]  = (int32_t)i;
            s.obs_point[at]  = (int32_t)o.wp;
            s.obs_pixel[at]  = {(double)o.point(0), (double)o.point(1)};
            s.obs_depth[at]  = (double)o.depth;
            s.obs_weight[at] = (double)o.weight;
            s.obs_outlier[at] = o.outlier ? 1 : 0;
            if (where) (*where)[at] = {(int)i, (int)k};
        }
    const auto& in = scene.intrinsics[0];
    s.K[0] = in.fx, s.K[1] = in.fy, s.K[2] = in.cx, s.K[3] = in.cy;
    s.bf = scene.bf;
    for (const auto& c : scene.rel_pose_constraints)
    {
        snk_ba_rpc r{};
        r.img1 = c.img1, r.img2 = c.img2;
        detail::pose7(c.rel_pose, r.rel_pose);
        r.weight_rotation = c.weight_rotation, r.weight_translation = c.weight_translation;
        s.rel_pose_constraints.push_back(r);
    }
    return s;
}

// poses / points back into the Saiga::Scene (the reference's solver mutates the scene in place).  `make_se3(q[4] xyzw, t[3])`
// builds the Snake side's SE3 -- with Sophus: [](const double* q, const double* t) { return SE3(Quat(q[3], q[0], q[1], q[2]), Vec3(t[0], t[1], t[2])); }
template <class SaigaScene, class MakeSE3>
inline void unflatten(const Scene& s, SaigaScene& scene, MakeSE3 make_se3)
{
    for (size_t i = 0; i < scene.images.size(); ++i) scene.images[i].se3 = make_se3(s.poses[i].data(), s.poses[i].data() + 4);
    for (size_t j = 0; j < scene.worldPoints.size(); ++j)
        for (int k = 0; k < 3; ++k) scene.worldPoints[j].p(k) = s.points[j][(size_t)k];
}

// LocalBundleAdjustment::SolveLocalScene (LocalBundleAdjustment.cpp:353-413) on the Saiga::Scene: create + initAndSolve, the
// chi-square pass that sets o.outlier (:368-397), one more iteration when anything was marked (:399-410); the scene's poses and
// points are the optimised ones afterwards.  Returns {outlierPoints, cost_initial, cost_final} like the reference.
template <class SaigaScene, class MakeSE3>
inline std::tuple<int, double, double> SolveLocalScene(BARec& cba, SaigaScene& scene, double chi2Mono, double chi2Stereo, MakeSE3 make_se3)
{
    std::vector<std::pair<int, int>> where;
    Scene s = flatten(scene, &where);
    cba.create(s);
    // initAndSolve (:357-365), the chi-square pass (:368-397) and the extra iteration (:399-410) in one library call; the pass
    // runs on the device with the reference's rule: valid, not yet an outlier, chi2 > (depth > 0 ? chi2Stereo : chi2Mono)
    const std::vector<uint8_t> before = s.obs_outlier;
    int outlierPoints                 = 0;
    OptimizationResults res           = cba.solveLocalScene(chi2Mono, chi2Stereo, &outlierPoints);
    for (size_t o = 0; o < s.obs_outlier.size(); ++o)
        if (s.obs_outlier[o] && (before.size() != s.obs_outlier.size() || !before[o]))
            scene.images[(size_t)where[o].first].stereoPoints[(size_t)where[o].second].outlier = true;  // :379, :391
    unflatten(s, scene, make_se3);
    return {outlierPoints, res.cost_initial, res.cost_final};
}
}  // namespace ref
}  // namespace snake_hip
