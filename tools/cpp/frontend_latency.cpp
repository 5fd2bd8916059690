// Latency of the reference's per-frame front-end through the C++ adaptor (snake_slam_amd/cpp/snake_hip.hpp), no Python in the timed
// region: for a NEW stereo frame every call  Frontend::Process  (= snk_frontend_process: FeatureDetector::Detect left + right,
// Preprocess::Process; Snake/Preprocess/FeatureDetector.cpp:116-156, Snake/Preprocess/Preprocess.cpp:35-53) and, for comparison, the
// same work as the six calls of the per-seam adaptor classes (ORBExtractor::Detect x 2, Preprocess::Rectify x 2, the feature grid,
// Preprocess::StereoMatching).  Images come as raw u8 arrays from tools/frontend_latency_cpp.py: <dir>/pair<k>_{left,right}.bin
// Round 5: the PIPELINED form (Frontend::Submit / Collect, the reference's stage queues: FeatureDetector.h:39, Preprocess.h:36) -- still one
// frame per call: (a) one thread, frame k + 1 submitted before frame k is collected; (b) two threads like the reference's FeatureDetection
// and Preprocess threads, one submitting, one collecting; for depths 2, 3, 4.  Every collected frame's stereo-match count is compared with
// the synchronous call's.
// usage: frontend_latency <dir> <n_pairs> <width> <height> [calls]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "snake_hip.hpp"

static std::vector<uint8_t> rd(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("missing input " + path);
    std::vector<uint8_t> v((size_t)f.tellg());
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)v.size());
    return v;
}
static double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.0 : v[v.size() / 2];
}

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    const std::string dir = argv[1];
    const int n = std::atoi(argv[2]), w = std::atoi(argv[3]), h = std::atoi(argv[4]), calls = argc > 5 ? std::atoi(argv[5]) : 200;
    try
    {
        using namespace snake_hip;
        std::vector<std::vector<uint8_t>> L, R;
        for (int k = 0; k < n; ++k) L.push_back(rd(dir + "/pair" + std::to_string(k) + "_left.bin")), R.push_back(rd(dir + "/pair" + std::to_string(k) + "_right.bin"));
        snk_frontend_params p{};
        p.orb = snk_orb_params{1000, 1.2f, 4, 20, 7, 0};
        const snk_rectification rect{{458.654, 457.296, 367.215, 248.375}, {0, 0, 0, 0, 0, 0, 0, 0}, {1, 0, 0, 0, 1, 0, 0, 0, 1}, {458.654, 457.296, 367.215, 248.375}, 47.9};
        p.rect_left = p.rect_right = rect;
        p.bounds         = snk_grid_bounds{0.0, 0.0, (double)w, (double)h};
        p.bf             = 47.9;
        p.relaxed_stereo = 1;
        p.stereo         = 1;
        Frontend fe(p);
        FrontendResult res;
        ORBExtractor ext(1000, 1.2f, 4, 20, 7);
        Preprocess pre;
        SnakeORBMatcher grid;
        const std::vector<float> ls = {1.0f, 1.2f, 1.2f * 1.2f, 1.2f * 1.2f * 1.2f};
        int n_one = 0, n_six = 0;
        auto one_call = [&](int k) { n_one = fe.Process(L[(size_t)k].data(), w, R[(size_t)k].data(), w, w, h, res); };
        auto six_calls = [&](int k)
        {
            std::vector<KeyPointF> kl, kr;
            std::vector<DescriptorORB> dl, dr;
            ext.Detect(L[(size_t)k].data(), w, h, w, kl, dl);
            ext.Detect(R[(size_t)k].data(), w, h, w, kr, dr);
            std::vector<snk_kp64> ul, ur;
            std::vector<std::array<double, 2>> norm;
            pre.Rectify(rect, kl, ul, &norm);
            pre.Rectify(rect, kr, ur, nullptr);
            FrameView fv;
            fv.undistorted_keypoints = ul;
            const std::vector<int32_t> perm = grid.CreateGrid(fv, p.bounds);
            std::vector<snk_kp64> g(ul.size());
            std::vector<DescriptorORB> gd(dl.size());
            for (size_t i = 0; i < ul.size(); ++i) g[(size_t)perm[i]] = ul[i], gd[(size_t)perm[i]] = dl[i];
            std::vector<float> rp, dp;
            n_six = pre.StereoMatching(g, gd, ur, dr, 47.9, ls, true, rp, dp);
        };
        double med[2];
        for (int which = 0; which < 2; ++which)
        {
            for (int k = 0; k < 8; ++k) which ? six_calls(k % n) : one_call(k % n);
            std::vector<double> ts;
            for (int k = 0; k < calls; ++k)
            {
                const auto t0 = std::chrono::steady_clock::now();
                which ? six_calls(k % n) : one_call(k % n);
                ts.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            }
            med[which] = median(ts);
        }
        // ---- pipelined: frames/s with one frame per call ----
        std::vector<int> want((size_t)n);
        const int n_one_last = n_one;
        for (int k = 0; k < n; ++k) one_call(k), want[(size_t)k] = n_one;
        n_one = n_one_last;
        double submit_us[3] = {0, 0, 0};  // mean time inside Submit, two-thread runs, per depth
        const int frames = calls * 5;
        bool same        = true;
        // round 6: the same frames from caller-owned page-locked buffers (Frontend::SubmitPinned: no staging copy in Submit)
        std::vector<std::unique_ptr<PinnedImage>> P;
        for (int k = 0; k < n; ++k)
        {
            P.emplace_back(new PinnedImage(w, h, 2));
            std::memcpy(P.back()->data(0), L[(size_t)k].data(), (size_t)w * h);
            std::memcpy(P.back()->data(1), R[(size_t)k].data(), (size_t)w * h);
        }
        std::string pipes[2];
        for (int pinned = 0; pinned < 2; ++pinned)
        {
        auto submit = [&](int k)
        {
            if (pinned)
                fe.SubmitPinned(P[(size_t)k]->data(0), w, P[(size_t)k]->data(1), w, w, h);
            else
                fe.Submit(L[(size_t)k].data(), w, R[(size_t)k].data(), w, w, h);
        };
        std::string pipe = "{";
        for (int depth = 2; depth <= 4; ++depth)
        {
            fe.SetDepth(depth);
            double fps[2];
            for (int threads = 1; threads <= 2; ++threads)
            {
                auto run = [&](int count) -> double
                {
                    const auto t0 = std::chrono::steady_clock::now();
                    if (threads == 1)
                    {
                        FrontendResult r;
                        for (int k = 0; k < count + depth - 1; ++k)
                        {
                            if (k < count) submit(k % n);
                            if (k >= depth - 1) { const int got = fe.Collect(r); same = same && got == want[(size_t)((k - depth + 1) % n)]; }
                        }
                    }
                    else
                    {
                        std::thread producer([&]
                        {
                            double in_submit = 0;
                            for (int k = 0; k < count; ++k)
                            {
                                const auto s0 = std::chrono::steady_clock::now();
                                submit(k % n);
                                in_submit += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - s0).count();
                            }
                            submit_us[depth - 2] = in_submit / count;
                        });
                        FrontendResult r;
                        for (int k = 0; k < count; ++k) { const int got = fe.Collect(r); same = same && got == want[(size_t)(k % n)]; }
                        producer.join();
                    }
                    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                };
                run(4 * depth);  // every slot past its captured frame
                fps[threads - 1] = (double)frames / run(frames);
            }
            char buf[240];
            std::snprintf(buf, sizeof(buf), "%s\"depth%d\": {\"one_thread_fps\": %.0f, \"two_threads_fps\": %.0f, \"mean_us_in_submit\": %.1f}", depth > 2 ? ", " : "", depth, fps[0], fps[1], submit_us[depth - 2]);
            pipe += buf;
        }
        pipe += "}";
        pipes[pinned] = pipe;
        }
        std::printf("{\"tool\": \"frontend_latency.cpp\", \"image\": \"%dx%d stereo\", \"calls\": %d, \"one_call_ms\": %.4f, \"six_calls_ms\": %.4f, "
                    "\"stereo_matches_last_frame\": {\"one_call\": %d, \"six_calls\": %d}, \"pipelined_frames\": %d, \"pipelined\": %s, "
                    "\"pipelined_pinned\": %s, \"pipelined_identical_match_counts\": %s}\n",
                    w, h, calls, med[0], med[1], n_one, n_six, frames, pipes[0].c_str(), pipes[1].c_str(), same ? "true" : "false");
        return n_one == n_six && same ? 0 : 3;
    }
    catch (const std::exception& e)
    {
        std::fprintf(stderr, "frontend_latency: %s\n", e.what());
        return 1;
    }
}
