// The matcher / preprocess handle (shared by matcher.hip and preprocess.hip).
#pragma once
#include "common.hpp"

struct snk_matcher : snk::HandleBase
{
    snk::DevBuf q, t, out, aux, aux2, cnt;
    // pinned staging of the per-frame (host pointer) projection matchers and of snk_pose_refine: the inputs of a call are copied here and
    // go up with one DMA, the results come back as one block (copies from / to pageable memory are staged by the runtime chunk by
    // chunk, and every one of them costs a submission: round 4, tools/latency_tracking.py)
    snk::HostBuf h_in, h_res;
    // device copy of the frame the projection matchers were last called with (track.hip): 1-2 coarse calls and one
    // fine call per frame (TrackingCoarse.cpp:234, TrackingFine.cpp:149) look at the same frame, which is uploaded once
    snk::DevBuf view;
    bool view_valid = false;  // snk_match_bind_frame: the arrays below live in `view`
    int view_n = 0, view_cols = 0, view_rows = 0;
    double view_bounds[4] = {0, 0, 0, 0};
};

namespace snk
{
// snk_stereo_match_batch_dev with the option of doing Frame::allocateTmp's -1000 prefill of right_points / depth itself (the one-call front-end)
int stereo_match_batch_dev_impl(snk_matcher* m, const snk_kp64* left_dev, const uint64_t* desc_left_dev, const int32_t* nl_dev, int nl_cap,
                                const snk_kp64* right_dev, const uint64_t* desc_right_dev, const int32_t* nr_dev, int nr_cap, int batch, double bf,
                                const float* level_scale_host, int n_levels, int relaxed, float* right_points_dev, float* depth_dev,
                                int32_t* n_matches_dev, bool prefill);
// both rectifications of a stereo frame in one launch: image 0 of the pair (kps_dev[0 .. cap)) with rect_left (+ normalized points), image 1
// (kps_dev[cap .. 2 cap)) with rect_right; n_dev[2], out_dev[2][cap]
int rectify_pair_dev(snk_matcher* m, const snk_rectification* rect_left, const snk_rectification* rect_right, const snk_keypoint* kps_dev,
                     const int32_t* n_dev, int cap, snk_kp64* out_dev, double* normalized_left_dev);
}  // namespace snk
