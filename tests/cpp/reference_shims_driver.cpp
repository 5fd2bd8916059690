// Drives snake_slam_amd/cpp/snake_hip_reference.hpp -- the hot path behind the REFERENCE'S OWN call signatures -- with mock
// structs that carry exactly the member names of the Snake-SLAM types the templates read:
//   Snake/Map/Frame.h:20-46 + Snake/Map/Features.h:18-41   -> Snake::Frame
//   Snake/Map/LocalMap.h:17-80                              -> CoarseTrackingPoint, FineTrackingPoint, LocalMap<T>
//   Snake/Map/MapPoint.h / Keyframe.h                       -> MapPoint (getPosition, GetDescriptor, IncreaseVisible), Keyframe
//   Saiga::Scene as Snake/Optimizer/LocalBundleAdjustment.cpp:187-346 fills it
// (no saiga / Eigen / Sophus here: the mocks stand in for Vec2 / Vec3 / SE3 with the accessors the reference code uses).
// Inputs / outputs: raw little-endian arrays in argv[1], written / checked by tests/test_cpp_reference_shims_gpu.py.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "snake_hip_reference.hpp"

// ---------------------------------------------------------------- mock maths types (Eigen / Sophus accessors)
template <int N>
struct Vec
{
    double v[N] = {};
    double& operator()(int i) { return v[i]; }
    const double& operator()(int i) const { return v[i]; }
    double x() const { return v[0]; }
    double y() const { return v[1]; }
    double z() const { return v[2]; }
};
using Vec2 = Vec<2>;
using Vec3 = Vec<3>;
struct Quat
{
    double qx = 0, qy = 0, qz = 0, qw = 1;
    double x() const { return qx; }
    double y() const { return qy; }
    double z() const { return qz; }
    double w() const { return qw; }
};
struct SE3
{
    Quat q;
    Vec3 t;
    const Quat& unit_quaternion() const { return q; }
    const Vec3& translation() const { return t; }
    Vec3& translation() { return t; }
};

// ---------------------------------------------------------------- mock Snake types (member names = the reference's)
namespace Snake
{
using FeatureDescriptor = std::array<uint64_t, 4>;  // Saiga::DescriptorORB
struct KeyPoint                                      // Saiga::KeyPoint<double>
{
    Vec2 point;
    double size = 0, angle = 0, response = 0;
    int octave = 0;
};
struct MapPoint
{
    int id_ = -1;
    Vec3 position;
    FeatureDescriptor descriptor{};
    int visible = 0;
    Vec3 getPosition() const { return position; }
    const FeatureDescriptor& GetDescriptor() const { return descriptor; }
    void IncreaseVisible() { ++visible; }
};
struct Features  // Snake/Map/Features.h:18-41
{
    int N = 0;
    std::vector<KeyPoint> keypoints, keypoints_right;
    std::vector<FeatureDescriptor> descriptors, descriptors_right;
    std::vector<Vec2> normalized_points;
    std::vector<KeyPoint> undistorted_keypoints;
    std::vector<float> right_points;
    std::vector<float> depth;
    snake_hip::ref::FeatureGrid2 grid;  // the one member whose TYPE a Snake build changes (was Saiga::FeatureGrid2)
};
struct Frame : Features  // Snake/Map/Frame.h:20-46
{
    void allocateTmp()  // Snake/Map/Frame.cpp:21-29
    {
        mvpMapPoints.resize((size_t)N, nullptr);
        mvbOutlier.resize((size_t)N, 0);
        right_points.resize((size_t)N, -1000);
        depth.resize((size_t)N, -1000);
        normalized_points.resize((size_t)N);
    }
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<int> mvbOutlier;
    SE3 local_velocity;
    SE3 tmpPose;
    SE3 Pose() const { return tmpPose; }
};
struct CoarseTrackingPoint  // LocalMap.h:17-30
{
    MapPoint* mp = nullptr;
    FeatureDescriptor descriptor{};
    Vec3 position, normal;
    int octave  = 0;
    float angle = 0;
};
struct FineTrackingPoint  // LocalMap.h:32-55
{
    MapPoint* mp = nullptr;
    FeatureDescriptor descriptor{};
    Vec3 position, normal;
    float reference_depth     = 0;
    int reference_scale_level = 0;
    bool valid                = true;
};
template <typename PointType>
struct LocalMap
{
    std::vector<PointType> points;
};
struct Keyframe
{
    std::vector<MapPoint*> pts;
    const std::vector<MapPoint*>& GetMapPointMatches() const { return pts; }
};
}  // namespace Snake

// ---------------------------------------------------------------- mock Saiga::Scene (what MakeLocalScene touches)
namespace Saiga
{
struct StereoImagePoint
{
    Vec2 point;
    double depth = 0;
    int wp       = -1;
    float weight = 1;
    bool outlier = false;
    explicit operator bool() const { return wp != -1 && !outlier; }
};
struct SceneImage
{
    int intr = 0;
    SE3 se3;
    bool constant   = false;
    int validPoints = 0, rel_constraints = 0;
    std::vector<StereoImagePoint> stereoPoints;
};
struct WorldPoint
{
    Vec3 p;
    bool constant = false, valid = true;
    std::vector<std::pair<int, int>> stereoreferences;
};
struct IntrinsicsPinholed
{
    double fx = 1, fy = 1, cx = 0, cy = 0;
};
struct RelPoseConstraint
{
    int img1 = 0, img2 = 0;
    SE3 rel_pose;
    double weight_rotation = 0, weight_translation = 0;
};
struct Scene
{
    std::vector<SceneImage> images;
    std::vector<WorldPoint> worldPoints;
    std::vector<IntrinsicsPinholed> intrinsics;
    std::vector<RelPoseConstraint> rel_pose_constraints;
    double bf = 0;
};
}  // namespace Saiga

static std::string g_dir;
template <typename T>
static std::vector<T> rd(const std::string& name)
{
    std::ifstream f(g_dir + "/" + name + ".bin", std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("missing input " + name);
    const size_t bytes = (size_t)f.tellg();
    if (bytes % sizeof(T)) throw std::runtime_error("size of " + name);
    std::vector<T> v(bytes / sizeof(T));
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)bytes);
    return v;
}
template <typename T>
static void wr(const std::string& name, const std::vector<T>& v)
{
    std::ofstream f(g_dir + "/" + name + ".bin", std::ios::binary);
    f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}
static SE3 se3_of(const double* p)
{
    SE3 T;
    T.q = Quat{p[0], p[1], p[2], p[3]};
    T.t(0) = p[4], T.t(1) = p[5], T.t(2) = p[6];
    return T;
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    g_dir = argv[1];
    try
    {
        using namespace snake_hip;
        // ---------------- Preprocess::Process order (Preprocess.cpp:35-49): allocateTmp, undistortKeypoints, computeFeatureGrid
        {
            ref::Globals G;
            G.rect_left         = rd<snk_rectification>("rect")[0];
            const auto b        = rd<double>("pp_bounds");
            G.featureGridBounds = snk_grid_bounds{b[0], b[1], b[2], b[3]};
            G.level_scale       = {1.0f, 1.2f, 1.44f, 1.728f};
            ref::Preprocess pre(G);
            Snake::Frame frame;
            const auto kps = rd<snk_keypoint>("rect_kps");
            const auto dsc = rd<Snake::FeatureDescriptor>("pp_desc");
            for (const auto& k : kps)  // frame.keypoints.emplace_back(kp.cast<double>()) (FeatureDetector.cpp:128-131)
            {
                Snake::KeyPoint kp;
                kp.point(0) = k.x, kp.point(1) = k.y, kp.size = k.size, kp.angle = k.angle, kp.response = k.response, kp.octave = k.octave;
                frame.keypoints.push_back(kp);
            }
            frame.descriptors = dsc;
            frame.N           = (int)frame.keypoints.size();
            frame.allocateTmp();
            pre.undistortKeypoints(frame);
            std::vector<double> und, nrm;
            for (int i = 0; i < frame.N; ++i)
            {
                const auto& u = frame.undistorted_keypoints[(size_t)i];
                und.insert(und.end(), {u.point(0), u.point(1), u.angle, (double)u.octave});
                nrm.insert(nrm.end(), {frame.normalized_points[(size_t)i](0), frame.normalized_points[(size_t)i](1)});
            }
            wr("out_pp_undistorted", und);
            wr("out_pp_normalized", nrm);
            pre.computeFeatureGrid(frame);
            std::vector<double> perm_x, perm_ux, perm_nx;
            std::vector<uint64_t> perm_d;
            for (int i = 0; i < frame.N; ++i)
            {
                perm_x.insert(perm_x.end(), {frame.keypoints[(size_t)i].point(0), frame.keypoints[(size_t)i].point(1)});
                perm_ux.insert(perm_ux.end(), {frame.undistorted_keypoints[(size_t)i].point(0), frame.undistorted_keypoints[(size_t)i].point(1)});
                perm_nx.insert(perm_nx.end(), {frame.normalized_points[(size_t)i](0), frame.normalized_points[(size_t)i](1)});
                perm_d.insert(perm_d.end(), frame.descriptors[(size_t)i].begin(), frame.descriptors[(size_t)i].end());
            }
            wr("out_pp_keypoints", perm_x), wr("out_pp_und_grid", perm_ux), wr("out_pp_norm_grid", perm_nx), wr("out_pp_desc_grid", perm_d);
            wr("out_pp_cell_start", frame.grid.cell_start);
        }
        // ---------------- StereoMatching(Frame&) (Preprocess.h:33)
        {
            ref::Globals G;
            const auto r  = rd<snk_rectification>("st_rect");
            G.rect_left = r[0], G.rect_right = r[1];
            G.level_scale = rd<float>("st_ls");
            ref::Preprocess pre(G);
            Snake::Frame frame;
            auto fill = [](const std::vector<snk_kp64>& in, std::vector<Snake::KeyPoint>& out)
            {
                for (const auto& k : in)
                {
                    Snake::KeyPoint kp;
                    kp.point(0) = k.x, kp.point(1) = k.y, kp.angle = k.angle, kp.octave = k.octave;
                    out.push_back(kp);
                }
            };
            fill(rd<snk_kp64>("st_left"), frame.keypoints);
            fill(rd<snk_kp64>("st_right"), frame.keypoints_right);
            frame.descriptors       = rd<Snake::FeatureDescriptor>("st_dl");
            frame.descriptors_right = rd<Snake::FeatureDescriptor>("st_dr");
            frame.N                 = (int)frame.keypoints.size();
            frame.allocateTmp();
            const int n = pre.StereoMatching(frame);
            if (!pre.exact()) throw std::runtime_error("StereoMatching: inputs were not float-representable");
            wr("out_st_rp", frame.right_points), wr("out_st_dp", frame.depth), wr("out_st_n", std::vector<int32_t>{n});
            // ---------------- ComputeStereoFromRGBD(Frame&) (Preprocess.cpp:79-120) on the same frame: undistorted = the left keypoints
            struct DepthView  // the members of Saiga::ImageView<float> the shim reads
            {
                const float* data;
                int width, height;
                size_t pitchBytes;
            };
            const auto dims = rd<int32_t>("rgbd_dims");  // width, height, pitch in floats
            const auto dimg = rd<float>("rgbd_depth");
            const auto mdl  = rd<snk_rgbd_model>("rgbd_model");
            frame.undistorted_keypoints = frame.keypoints;
            const DepthView dv{dimg.data(), dims[0], dims[1], (size_t)dims[2] * sizeof(float)};
            const int nd = pre.ComputeStereoFromRGBD(frame, dv, mdl[0]);
            wr("out_rgbd_rp", frame.right_points), wr("out_rgbd_dp", frame.depth), wr("out_rgbd_n", std::vector<int32_t>{nd});
        }
        // ---------------- the three SnakeORBMatcher searches with the reference's signatures
        {
            ref::Globals G;
            const auto cam      = rd<snk_camera>("tr_cam");
            G.K                 = cam[0];
            G.baseline          = cam[0].bf / cam[0].fx;
            const auto b        = rd<double>("tr_bounds");
            G.featureGridBounds = snk_grid_bounds{b[0], b[1], b[2], b[3]};
            G.level_scale       = rd<float>("tr_ls");
            ref::Preprocess pre(G);
            ref::SnakeORBMatcher matcher(G);
            Snake::Frame frame;
            const auto kps   = rd<snk_kp64>("tr_kps");
            const auto taken = rd<uint8_t>("tr_taken");
            for (const auto& k : kps)
            {
                Snake::KeyPoint kp;
                kp.point(0) = k.x, kp.point(1) = k.y, kp.angle = k.angle, kp.octave = k.octave;
                frame.undistorted_keypoints.push_back(kp);
            }
            frame.keypoints   = frame.undistorted_keypoints;
            frame.descriptors = rd<Snake::FeatureDescriptor>("tr_desc");
            frame.N           = (int)kps.size();
            frame.allocateTmp();
            frame.right_points = rd<float>("tr_rp");
            pre.computeFeatureGrid(frame);  // the fixture is in grid order: the permutation is the identity
            frame.tmpPose = se3_of(rd<double>("tr_pose").data());
            frame.local_velocity.translation()(2) = -G.baseline;  // z_diff == baseline: neither bForward nor bBackward (:210-212)
            Snake::MapPoint earlier;                              // what mvpMapPoints holds for the features the fixture marks taken
            auto reset = [&]()
            {
                for (size_t i = 0; i < taken.size(); ++i) frame.mvpMapPoints[i] = taken[i] ? &earlier : nullptr;
            };
            auto dump = [&](const std::string& name, const std::vector<Snake::MapPoint>& mps)
            {
                std::vector<int32_t> ids;
                for (auto* p : frame.mvpMapPoints) ids.push_back(p == nullptr ? -1 : (p == &earlier ? -2 : (int32_t)(p - mps.data())));
                wr(name, ids);
            };
            // coarse
            const auto lc = rd<snk_lm_coarse>("tr_coarse");
            std::vector<Snake::MapPoint> mpc(lc.size());
            Snake::LocalMap<Snake::CoarseTrackingPoint> lmc;
            for (size_t i = 0; i < lc.size(); ++i)
            {
                Snake::CoarseTrackingPoint p;
                p.mp = &mpc[i];
                for (int k = 0; k < 3; ++k) p.position(k) = lc[i].pos[k], p.normal(k) = lc[i].normal[k];
                std::memcpy(p.descriptor.data(), lc[i].desc, 32);
                p.octave = lc[i].octave, p.angle = lc[i].angle;
                lmc.points.push_back(p);
            }
            reset();
            int n = matcher.SearchByProjectionFrameFrame2(frame, lmc, 15.0f, 75, 4);
            dump("out_tr_coarse_mvp", mpc);
            wr("out_tr_coarse_n", std::vector<int32_t>{n});
            // fine
            const auto lf = rd<snk_lm_fine>("tr_fine");
            std::vector<Snake::MapPoint> mpf(lf.size());
            Snake::LocalMap<Snake::FineTrackingPoint> lmf;
            for (size_t i = 0; i < lf.size(); ++i)
            {
                Snake::FineTrackingPoint p;
                p.mp = &mpf[i];
                for (int k = 0; k < 3; ++k) p.position(k) = lf[i].pos[k], p.normal(k) = lf[i].normal[k];
                std::memcpy(p.descriptor.data(), lf[i].desc, 32);
                p.reference_depth = lf[i].reference_depth, p.reference_scale_level = lf[i].reference_scale_level, p.valid = lf[i].valid != 0;
                lmf.points.push_back(p);
            }
            reset();
            n = matcher.SearchByProjection2(frame, lmf, 5.0f, 0.8f, 4);
            dump("out_tr_fine_mvp", mpf);
            std::vector<int32_t> vis, valid;
            for (size_t i = 0; i < lf.size(); ++i) vis.push_back(mpf[i].visible), valid.push_back(lmf.points[i].valid ? 1 : 0);
            wr("out_tr_fine_n", std::vector<int32_t>{n}), wr("out_tr_fine_vis", vis), wr("out_tr_fine_valid", valid);
            // frame to keyframe
            const auto kpos  = rd<std::array<double, 3>>("tr_kf_pos");
            const auto kdesc = rd<Snake::FeatureDescriptor>("tr_kf_desc");
            const auto kskip = rd<uint8_t>("tr_kf_skip");
            std::vector<Snake::MapPoint> mpk(kpos.size());
            Snake::Keyframe kf;
            for (size_t i = 0; i < kpos.size(); ++i)
            {
                for (int k = 0; k < 3; ++k) mpk[i].position(k) = kpos[i][(size_t)k];
                mpk[i].descriptor = kdesc[i];
                kf.pts.push_back(kskip[i] ? nullptr : &mpk[i]);
            }
            reset();
            n = matcher.SearchByProjectionFrameToKeyframe(frame, kf, 15.0f, 100);
            dump("out_tr_kf_mvp", mpk);
            wr("out_tr_kf_n", std::vector<int32_t>{n});
        }
        // ---------------- FeatureDetector::Detect + Preprocess::Process of a stereo frame in ONE call (ref::FrontEnd)
        {
            struct ImageView  // the members of Saiga::ImageView<unsigned char> the shim reads
            {
                const unsigned char* data;
                int width, height;
                size_t pitchBytes;
            };
            ref::Globals G;
            const auto r = rd<snk_rectification>("fe_rect");
            G.rect_left = r[0], G.rect_right = r[1];
            const auto b        = rd<double>("fe_bounds");
            G.featureGridBounds = snk_grid_bounds{b[0], b[1], b[2], b[3]};
            const auto dims     = rd<int32_t>("fe_dims");  // width, height, pitch
            const auto left = rd<uint8_t>("fe_left"), right = rd<uint8_t>("fe_right");
            ref::FrontEnd fe(G, snk_orb_params{1000, 1.2f, 4, 20, 7, 0});
            const ImageView lv{left.data(), dims[0], dims[1], (size_t)dims[2]}, rv{right.data(), dims[0], dims[1], (size_t)dims[2]};
            std::vector<int32_t> counts;
            Snake::Frame frame;
            for (int rep = 0; rep < 3; ++rep)  // plain launches, the recorded graph, its replay: the last frame is written out
            {
                frame = Snake::Frame();
                const int n = fe.DetectAndProcess(frame, lv, &rv);
                counts.insert(counts.end(), {frame.N, (int32_t)frame.keypoints_right.size(), n});
            }
            std::vector<double> kp, und, nrm, kpr;
            std::vector<uint64_t> dl, dr;
            for (int i = 0; i < frame.N; ++i)
            {
                const auto& k = frame.keypoints[(size_t)i];
                const auto& u = frame.undistorted_keypoints[(size_t)i];
                kp.insert(kp.end(), {k.point(0), k.point(1), k.size, k.angle, k.response, (double)k.octave});
                und.insert(und.end(), {u.point(0), u.point(1), u.angle, (double)u.octave});
                nrm.insert(nrm.end(), {frame.normalized_points[(size_t)i](0), frame.normalized_points[(size_t)i](1)});
                dl.insert(dl.end(), frame.descriptors[(size_t)i].begin(), frame.descriptors[(size_t)i].end());
            }
            for (size_t i = 0; i < frame.keypoints_right.size(); ++i)
            {
                const auto& k = frame.keypoints_right[i];
                kpr.insert(kpr.end(), {k.point(0), k.point(1), k.size, k.angle, k.response, (double)k.octave});
                dr.insert(dr.end(), frame.descriptors_right[i].begin(), frame.descriptors_right[i].end());
            }
            wr("out_fe_counts", counts), wr("out_fe_kp", kp), wr("out_fe_und", und), wr("out_fe_norm", nrm), wr("out_fe_dl", dl);
            wr("out_fe_kpr", kpr), wr("out_fe_dr", dr), wr("out_fe_rp", frame.right_points), wr("out_fe_dp", frame.depth);
            wr("out_fe_cell_start", frame.grid.cell_start);
            if ((int)frame.mvpMapPoints.size() != frame.N) throw std::runtime_error("allocateTmp was not applied");
        }
        // ---------------- SolveLocalScene on a Saiga::Scene (LocalBundleAdjustment.cpp:353-413)
        {
            Saiga::Scene scene;
            const auto pose = rd<std::array<double, 7>>("ba_pose");
            const auto ic   = rd<uint8_t>("ba_img_const");
            const auto pt   = rd<std::array<double, 3>>("ba_pt");
            const auto pc   = rd<uint8_t>("ba_pt_const");
            const auto oi = rd<int32_t>("ba_obs_img"), op = rd<int32_t>("ba_obs_pt");
            const auto uv = rd<std::array<double, 2>>("ba_obs_uv");
            const auto od = rd<double>("ba_obs_depth"), ow = rd<double>("ba_obs_weight");
            const auto K  = rd<double>("ba_K");
            scene.bf = rd<double>("ba_bf")[0];
            scene.intrinsics.push_back(Saiga::IntrinsicsPinholed{K[0], K[1], K[2], K[3]});
            for (size_t i = 0; i < pose.size(); ++i)
            {
                Saiga::SceneImage si;
                si.se3      = se3_of(pose[i].data());
                si.constant = ic[i] != 0;
                scene.images.push_back(si);
            }
            for (size_t j = 0; j < pt.size(); ++j)
            {
                Saiga::WorldPoint wp;
                for (int k = 0; k < 3; ++k) wp.p(k) = pt[j][(size_t)k];
                wp.constant = pc[j] != 0;
                scene.worldPoints.push_back(wp);
            }
            for (size_t o = 0; o < oi.size(); ++o)  // MakeLocalScene's inner loop (:265-290)
            {
                Saiga::StereoImagePoint ip;
                ip.point(0) = uv[o][0], ip.point(1) = uv[o][1];
                ip.depth = od[o], ip.wp = op[o], ip.weight = (float)ow[o];
                auto& img = scene.images[(size_t)oi[o]];
                scene.worldPoints[(size_t)op[o]].stereoreferences.emplace_back(oi[o], (int)img.stereoPoints.size());
                img.stereoPoints.push_back(ip);
                img.validPoints++;
            }
            BARec cba;
            auto make_se3 = [](const double* q, const double* t)
            {
                SE3 T;
                T.q = Quat{q[0], q[1], q[2], q[3]};
                T.t(0) = t[0], T.t(1) = t[1], T.t(2) = t[2];
                return T;
            };
            const auto [outlierPoints, cost_initial, cost_final] = ref::SolveLocalScene(cba, scene, 2.1 * 2.1, 2.3 * 2.3, make_se3);
            std::vector<double> poses, points;
            std::vector<int32_t> outl;  // (image, world point) of every observation marked o.outlier
            for (const auto& im : scene.images)
            {
                double p[7];
                ref::detail::pose7(im.se3, p);
                poses.insert(poses.end(), p, p + 7);
            }
            for (const auto& w : scene.worldPoints) points.insert(points.end(), {w.p(0), w.p(1), w.p(2)});
            for (size_t i = 0; i < scene.images.size(); ++i)
                for (const auto& o : scene.images[i].stereoPoints)
                    if (o.outlier) outl.push_back((int32_t)i), outl.push_back(o.wp);
            wr("out_ba_pose", poses), wr("out_ba_pt", points), wr("out_ba_outliers", outl);
            wr("out_ba_res", std::vector<double>{(double)outlierPoints, cost_initial, cost_final});
        }
    }
    catch (const std::exception& e)
    {
        std::cerr << "reference_shims_driver: " << e.what() << "\n";
        return 1;
    }
    std::puts("reference_shims_driver: ok");
    return 0;
}
