// Runs the C++ adaptor (snake_slam_amd/cpp/snake_hip.hpp -- the header a Snake-SLAM maintainer includes)
// on the golden inputs, the way the reference's threads call the saiga classes it stands in for:
//   ORBExtractor::Detect -> Preprocess::Rectify -> StereoMatching -> matchKnn2 / filterMatches ->
//   SearchByProjectionFrameFrame2 / SearchByProjection2 -> optimizePoseRobust -> BARec create / initAndSolve.
// Inputs and outputs are raw little-endian arrays in the directory argv[1] (written / read by
// tests/test_cpp_adaptor_gpu.py, which compares every output with tests/golden/*.npz).
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "snake_hip.hpp"

static std::string g_dir;

template <typename T>
static std::vector<T> rd(const std::string& name)
{
    std::ifstream f(g_dir + "/" + name + ".bin", std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("missing input " + name);
    const size_t bytes = (size_t)f.tellg();
    if (bytes % sizeof(T)) throw std::runtime_error("size of " + name + " is not a multiple of its element size");
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

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    g_dir = argv[1];
    try
    {
        using namespace snake_hip;
        const std::vector<float> ls3 = {1.0f, 1.2f, 1.2f * 1.2f};
        {  // FeatureDetector::Detect (FeatureDetector.cpp:119-132)
            const auto dims = rd<int32_t>("orb_dims");  // w, h, nfeatures, levels, iniTh, minTh
            const auto img  = rd<uint8_t>("orb_img");
            ORBExtractor ext(dims[2], 1.2f, dims[3], dims[4], dims[5], 2);
            std::vector<KeyPointF> kps;
            std::vector<DescriptorORB> desc;
            ext.Detect(img.data(), dims[0], dims[1], dims[0], kps, desc);
            wr("out_orb_kps", kps);
            wr("out_orb_desc", desc);
            // a second image size through the same object (the extractor reconfigures itself)
            ext.Detect(img.data(), dims[0] / 2, dims[1], dims[0], kps, desc);
            wr("out_orb_kps_half", kps);
        }
        Preprocess pre;
        {  // undistortKeypoints (Preprocess.cpp:55-77)
            const auto rect = rd<snk_rectification>("rect");
            const auto kps  = rd<snk_keypoint>("rect_kps");
            std::vector<snk_kp64> out;
            std::vector<std::array<double, 2>> norm;
            pre.Rectify(rect[0], kps, out, &norm);
            wr("out_rect", out);
            wr("out_rect_norm", norm);
        }
        {  // StereoMatching (Preprocess.cpp:122-242)
            const auto left = rd<snk_kp64>("st_left"), right = rd<snk_kp64>("st_right");
            const auto dl = rd<DescriptorORB>("st_dl"), dr = rd<DescriptorORB>("st_dr");
            const auto bf = rd<double>("st_bf");
            const auto ls = rd<float>("st_ls");
            std::vector<float> rp, dp;
            const int n = pre.StereoMatching(left, dl, right, dr, bf[0], ls, true, rp, dp);
            wr("out_st_rp", rp);
            wr("out_st_dp", dp);
            wr("out_st_n", std::vector<int32_t>{n});
        }
        {  // TrackBruteForce (TrackingCoarse.cpp:350-352)
            const auto q = rd<DescriptorORB>("bf_q"), t = rd<DescriptorORB>("bf_t");
            BruteForceMatcher m;
            m.matchKnn2_omp(q, t, 4);
            const int n = m.filterMatches(120, 0.9f);
            wr("out_bf_knn", m.knn());
            std::vector<int32_t> pairs;
            for (const auto& p : m.matches) pairs.push_back(p.first), pairs.push_back(p.second);
            if ((int)m.matches.size() != n) throw std::runtime_error("filterMatches count");
            wr("out_bf_pairs", pairs);
        }
        {  // SearchByProjectionFrameFrame2 / SearchByProjection2 (TrackingCoarse.cpp:234, TrackingFine.cpp:149)
            FrameView fv;
            fv.undistorted_keypoints = rd<snk_kp64>("tr_kps");
            fv.descriptors           = rd<DescriptorORB>("tr_desc");
            fv.right_points          = rd<float>("tr_rp");
            fv.taken                 = rd<uint8_t>("tr_taken");
            const auto bounds        = rd<double>("tr_bounds");
            SnakeORBMatcher om;
            // the grid of the fixture is rebuilt through the adaptor: the permutation must be the identity (the fixture
            // is already in grid order) and cell_start must equal the fixture's
            const auto perm = om.CreateGrid(fv, snk_grid_bounds{bounds[0], bounds[1], bounds[2], bounds[3]});
            wr("out_tr_perm", perm);
            wr("out_tr_cell_start", fv.cell_start);
            const auto cam  = rd<snk_camera>("tr_cam");
            const auto pose = rd<double>("tr_pose");
            const auto ls   = rd<float>("tr_ls");
            const auto lc   = rd<snk_lm_coarse>("tr_coarse");
            std::vector<int32_t> match;
            int n = om.SearchByProjectionFrameFrame2(fv, cam[0], pose.data(), lc, 15.0f, 75, 0, ls, match);
            match.push_back(n);
            wr("out_tr_coarse", match);
            // the same call on the bound frame (uploaded once) must give the same answer
            const auto stale = om.BindFrame(fv);
            const auto bound = om.BindFrame(fv);  // a second binding: the first token is stale from here on
            std::vector<int32_t> match_b;
            const int nb = om.SearchByProjectionFrameFrame2(bound, cam[0], pose.data(), lc, 15.0f, 75, 0, ls, match_b);
            match_b.push_back(nb);
            if (match_b != match) throw std::runtime_error("bound frame: coarse result differs");
            bool refused = false;
            try
            {
                om.SearchByProjectionFrameFrame2(stale, cam[0], pose.data(), lc, 15.0f, 75, 0, ls, match_b);
            }
            catch (const std::logic_error&)
            {
                refused = true;
            }
            if (!refused) throw std::runtime_error("a stale binding token was accepted");
            // the caller refills the SAME FrameView object for another frame (the usual pattern): the bound copy on the device must
            // not change and the FrameView overload must upload the new contents (round 2 matched against the stale copy here)
            FrameView other = fv;
            std::swap(fv.taken, other.taken);
            for (auto& t : fv.taken) t = 1;  // every feature taken: no match possible on the refilled view
            std::vector<int32_t> match_r;
            if (om.SearchByProjectionFrameFrame2(fv, cam[0], pose.data(), lc, 15.0f, 75, 0, ls, match_r) != 0)
                throw std::runtime_error("refilled FrameView was not uploaded");
            fv.taken = other.taken;
            om.UpdateTaken(bound, fv.taken);
            auto lf = rd<snk_lm_fine>("tr_fine");
            std::vector<uint8_t> vis;
            n = om.SearchByProjection2(bound, cam[0], pose.data(), lf, 5.0f, 0.8f, ls, match, vis);
            match.push_back(n);
            wr("out_tr_fine", match);
            wr("out_tr_fine_vis", vis);
            std::vector<uint8_t> valid;
            for (const auto& p : lf) valid.push_back(p.valid);
            wr("out_tr_fine_valid", valid);
            om.UnbindFrame();
        }
        {  // PoseRefinement::refinePose (PoseRefinement.h:27-66)
            const auto cam = rd<snk_camera>("po_cam");
            auto pose      = rd<double>("po_pose0");
            const auto wps = rd<std::array<double, 3>>("po_wps");
            const auto obs = rd<snk_pose_obs>("po_obs");
            PoseRefinement pr(1.0);
            std::vector<uint8_t> outl;
            const int inl = pr.optimizePoseRobust(wps, obs, outl, pose.data(), cam[0]);
            wr("out_po_pose", pose);
            wr("out_po_outlier", outl);
            wr("out_po_inliers", std::vector<int32_t>{inl});
        }
        {  // SolveLocalScene (LocalBundleAdjustment.cpp:353-413)
            Scene sc;
            sc.poses          = rd<std::array<double, 7>>("ba_pose");
            sc.image_constant = rd<uint8_t>("ba_img_const");
            sc.points         = rd<std::array<double, 3>>("ba_pt");
            sc.point_constant = rd<uint8_t>("ba_pt_const");
            sc.obs_image      = rd<int32_t>("ba_obs_img");
            sc.obs_point      = rd<int32_t>("ba_obs_pt");
            sc.obs_pixel      = rd<std::array<double, 2>>("ba_obs_uv");
            sc.obs_depth      = rd<double>("ba_obs_depth");
            sc.obs_weight     = rd<double>("ba_obs_weight");
            const auto K      = rd<double>("ba_K");
            for (int k = 0; k < 4; ++k) sc.K[k] = K[(size_t)k];
            sc.bf = rd<double>("ba_bf")[0];
            BARec ba;
            ba.create(sc);
            wr("out_ba_chi2", ba.residualsSquared());
            const OptimizationResults r = ba.initAndSolve();
            wr("out_ba_pose", sc.poses);
            wr("out_ba_pt", sc.points);
            wr("out_ba_cost", std::vector<double>{r.cost_initial, r.cost_final});
            // chi-square pass + one more iteration (:368-410)
            const auto chi2 = ba.residualsSquared();
            sc.obs_outlier.assign(chi2.size(), 0);
            int n_out = 0;
            for (size_t i = 0; i < chi2.size(); ++i)
                if (chi2[i] > (sc.obs_depth[i] > 0 ? 5.29 : 4.41)) sc.obs_outlier[i] = 1, ++n_out;
            ba.optimizationOptions.max_iterations = 1;
            const OptimizationResults r2 = ba.solve();
            wr("out_ba_cost2", std::vector<double>{r2.cost_initial, r2.cost_final, (double)n_out});
            wr("out_ba_outlier", sc.obs_outlier);
            wr("out_ba_pose2", sc.poses);
            wr("out_ba_pt2", sc.points);
            for (int round = 0; round < 2; ++round)
            {
                // BARec::solveLocalScene (one library call) from the scene as it was read: with the reference's thresholds (this
                // scene has nothing to mark: the result is the first solve's) and with thresholds that mark a good part of it
                Scene sf = sc;
                sf.poses  = rd<std::array<double, 7>>("ba_pose");
                sf.points = rd<std::array<double, 3>>("ba_pt");
                sf.obs_outlier.clear();
                BARec bf;
                bf.create(sf);
                int marked = -1;
                const OptimizationResults rf = round == 0 ? bf.solveLocalScene(4.41, 5.29, &marked) : bf.solveLocalScene(0.3, 0.4, &marked);
                const std::string tag = round == 0 ? "out_ba_fused" : "out_ba_fused_low";
                wr(tag + "_pose", sf.poses);
                wr(tag + "_pt", sf.points);
                wr(tag + "_outlier", sf.obs_outlier);
                wr(tag + "_cost", std::vector<double>{rf.cost_initial, rf.cost_final, (double)marked});
            }
            // GlobalBundleAdjustment::PointBA (:103-122) and the BAPoseOnly of RealignIntermiediateFrames (:306-316), each from
            // the scene as it was read
            Scene sp = sc;
            sp.poses  = rd<std::array<double, 7>>("ba_pose");
            sp.points = rd<std::array<double, 3>>("ba_pt");
            sp.obs_outlier.clear();
            Scene sq = sp;
            BAPointOnly pba;
            pba.create(sp);
            const OptimizationResults rp = pba.initAndSolve();
            wr("out_pba_pt", sp.points);
            wr("out_pba_pose", sp.poses);
            wr("out_pba_cost", std::vector<double>{rp.cost_initial, rp.cost_final});
            BAPoseOnly qba;
            qba.create(sq);
            const OptimizationResults rq = qba.initAndSolve();
            wr("out_qba_pt", sq.points);
            wr("out_qba_pose", sq.poses);
            wr("out_qba_cost", std::vector<double>{rq.cost_initial, rq.cost_final});
        }
    }
    catch (const std::exception& e)
    {
        std::cerr << "adaptor_driver: " << e.what() << "\n";
        return 1;
    }
    std::puts("adaptor_driver: ok");
    return 0;
}
