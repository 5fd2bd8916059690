// Latency of the reference's per-keyframe local-BA call through the C++ adaptor (snake_slam_amd/cpp/snake_hip.hpp), no Python
// in the timed region: for a NEW scene every call  BARec::create(scene) -> BARec::solveLocalScene(chi2Mono, chi2Stereo)
// (LocalBundleAdjustment.cpp:357-410) and, for comparison, the same work call by call (initAndSolve, residualsSquared, the
// host threshold loop, solve with one iteration).  Scenes come as raw arrays from tools/lba_call_latency_cpp.py:
//   <dir>/scene<k>_{pose,img_const,pt,pt_const,obs_img,obs_pt,obs_uv,obs_depth,obs_weight,K,bf}.bin
// usage: lba_call_latency <dir> <n_scenes> [rounds]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <fstream>
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
    std::vector<T> v(bytes / sizeof(T));
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)bytes);
    return v;
}

static double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.0 : v[v.size() / 2];
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    g_dir             = argv[1];
    const int n       = std::atoi(argv[2]);
    const int rounds  = argc > 3 ? std::atoi(argv[3]) : 3;
    using namespace snake_hip;
    using clk = std::chrono::steady_clock;
    auto ms   = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    try
    {
        std::vector<Scene> scenes((size_t)n);
        for (int k = 0; k < n; ++k)
        {
            const std::string p = "scene" + std::to_string(k) + "_";
            Scene& sc           = scenes[(size_t)k];
            sc.poses            = rd<std::array<double, 7>>(p + "pose");
            sc.image_constant   = rd<uint8_t>(p + "img_const");
            sc.points           = rd<std::array<double, 3>>(p + "pt");
            sc.point_constant   = rd<uint8_t>(p + "pt_const");
            sc.obs_image        = rd<int32_t>(p + "obs_img");
            sc.obs_point        = rd<int32_t>(p + "obs_pt");
            sc.obs_pixel        = rd<std::array<double, 2>>(p + "obs_uv");
            sc.obs_depth        = rd<double>(p + "obs_depth");
            sc.obs_weight       = rd<double>(p + "obs_weight");
            const auto K        = rd<double>(p + "K");
            for (int i = 0; i < 4; ++i) sc.K[i] = K[(size_t)i];
            sc.bf = rd<double>(p + "bf")[0];
        }
        BARec ba;
        std::vector<double> t_create, t_fused, t_total, s_create, s_solve, s_chi, s_extra, s_total;
        long marked_fused = 0, marked_steps = 0;
        for (int r = 0; r < rounds; ++r)  // round 0 = warm-up (buffers grow, kernels load)
            for (int k = 0; k < n; ++k)
            {
                Scene sc = scenes[(size_t)k];  // the call mutates the scene
                const auto t0 = clk::now();
                ba.create(sc);
                const auto t1 = clk::now();
                int marked    = 0;
                ba.solveLocalScene(4.41, 5.29, &marked);
                const auto t2 = clk::now();
                if (r)
                {
                    t_create.push_back(ms(t0, t1)), t_fused.push_back(ms(t1, t2)), t_total.push_back(ms(t0, t2));
                    marked_fused += marked;
                }
            }
        for (int r = 0; r < rounds; ++r)
            for (int k = 0; k < n; ++k)
            {
                Scene sc = scenes[(size_t)k];
                const auto t0 = clk::now();
                ba.create(sc);
                const auto t1 = clk::now();
                ba.initAndSolve();
                const auto t2   = clk::now();
                const auto chi2 = ba.residualsSquared();
                sc.obs_outlier.assign(chi2.size(), 0);
                int n_out = 0;
                for (size_t i = 0; i < chi2.size(); ++i)
                    if (chi2[i] > (sc.obs_depth[i] > 0 ? 5.29 : 4.41)) sc.obs_outlier[i] = 1, ++n_out;
                const auto t3 = clk::now();
                if (n_out > 0)
                {
                    ba.optimizationOptions.max_iterations = 1;
                    ba.solve();
                    ba.optimizationOptions.max_iterations = 3;
                }
                const auto t4 = clk::now();
                if (r)
                {
                    s_create.push_back(ms(t0, t1)), s_solve.push_back(ms(t1, t2)), s_chi.push_back(ms(t2, t3)), s_extra.push_back(ms(t3, t4));
                    s_total.push_back(ms(t0, t4));
                    marked_steps += n_out;
                }
            }
        std::printf("{\"tool\": \"lba_call_latency.cpp\", \"calls\": %zu, \"one_call_ms\": {\"create\": %.4f, \"solveLocalScene\": %.4f, \"total\": %.4f}, "
                    "\"call_by_call_ms\": {\"create\": %.4f, \"initAndSolve\": %.4f, \"chi2_pass\": %.4f, \"solve1\": %.4f, \"total\": %.4f}, "
                    "\"marked_per_call\": {\"one_call\": %.1f, \"call_by_call\": %.1f}}\n",
                    t_total.size(), median(t_create), median(t_fused), median(t_total), median(s_create), median(s_solve), median(s_chi),
                    median(s_extra), median(s_total), (double)marked_fused / (double)std::max<size_t>(t_total.size(), 1),
                    (double)marked_steps / (double)std::max<size_t>(s_total.size(), 1));
    }
    catch (const std::exception& e)
    {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
