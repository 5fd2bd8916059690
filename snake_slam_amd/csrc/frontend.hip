// One stereo frame through the whole front-end in ONE call and ONE synchronisation: the per-frame call shape of the reference
// (FeatureDetector::Detect for the left and the right image, Snake/Preprocess/FeatureDetector.cpp:116-156, then
// Preprocess::Process = allocateTmp, undistortKeypoints, computeFeatureGrid, StereoMatching, Snake/Preprocess/Preprocess.cpp:35-53)
// served the way snk_ba_solve_local_scene serves SolveLocalScene.  Call by call through the host entry points the same work is five
// uploads, five downloads and five synchronisations (Detect x 2, rectify x 2, grid, StereoMatching: ~0.75 ms per 752x480 stereo
// frame, profiles/r03/r03h_host_latency.log); here the two images go up with one copy, the extractor runs them as ONE two-image
// launch chain, rectification / grid / reorder / StereoMatching are enqueued behind it on the same stream, everything the frame
// needs comes back with one copy, and the host waits once.
//
// The kernels are those of the device-resident batch entry points (snk_orb_detect_batch_dev, snk_rectify_batch_dev,
// snk_feature_grid_batch_dev, snk_stereo_match_batch_dev) -- this file adds no arithmetic, so the results are bit for bit those of
// the call-by-call path (tests/test_frontend_gpu.py compares both with each other and with the oracle).
//
// hipGraph: for a fixed image size every launch of the chain has fixed arguments (the handle's own buffers; the counts live on the
// device), so the sequence between the upload and the download is recorded ONCE per configuration with a stream capture in
// hipStreamCaptureModeRelaxed on the handle's own stream and replayed with one hipGraphLaunch.  (The library never touches the
// legacy stream -- DESIGN section 0b -- so a capture on one handle's stream cannot be invalidated by another thread's calls.)
// The first frame of a configuration runs uncaptured (it sizes every scratch buffer: allocation is illegal inside a capture), the
// second is captured, later ones replay.  SNK_FRONTEND_NO_GRAPH=1 keeps plain launches (A/B, tests).  The recorded launches depend on the
// "iround.mode" definition (a kernel argument of StereoMatching): the graph is keyed by it and rebuilt when it changes.
#include <cmath>

#include "common.hpp"

using namespace snk;

// Saiga::FeatureGridBounds2<double, 20>: 20-px cells over the bounds (the same rule as track.hip's grid_dims)
static void grid_dims(const snk_grid_bounds* b, int* cols, int* rows)
{
    const int c = (int)ceil((b->max_x - b->min_x) / 20.0), r = (int)ceil((b->max_y - b->min_y) / 20.0);
    *cols = c < 1 ? 1 : c;
    *rows = r < 1 ? 1 : r;
}

struct snk_frontend
{
    int device         = 0;
    hipStream_t stream = nullptr;
    snk_orb* orb       = nullptr;
    snk_matcher* mat   = nullptr;
    snk_frontend_params par{};
    int width = 0, height = 0, dpitch = 0, cap = 0, cols = 0, rows = 0, n_img = 2;
    float level_scale[8] = {};
    // one device block for everything that goes back to the host, one pinned mirror
    DevBuf d_img, d_out, d_tmp;
    HostBuf h_img, h_out;
    size_t o_n = 0, o_kps = 0, o_desc_r = 0, o_kp64_g = 0, o_desc_g = 0, o_norm = 0, o_perm = 0, o_cs = 0, o_rp = 0, o_dp = 0, out_len = 0;
    size_t t_desc = 0, t_kp64 = 0;  // d_tmp: descriptors in extractor order (2 images) | rectified keypoints (2 images)
    hipGraphExec_t graph = nullptr;
    int graph_key        = -1;  // definition values the recorded launches depend on
    int last_key         = -1;  // ... of the previous frame (a change runs one frame uncaptured: the extractor may size new scratch)
    bool graph_failed    = false;  // a capture / instantiation failed for this configuration: plain launches from then on
    int frames_seen      = 0;   // of the current configuration
};

static void drop_graph(snk_frontend* f)
{
    if (f->graph) (void)hipGraphExecDestroy(f->graph);
    f->graph        = nullptr;
    f->graph_key    = -1;
    f->graph_failed = false;
}

extern "C" int snk_frontend_create(const snk_frontend_params* params, int device, snk_frontend** out)
{
    SNK_REQUIRE(params != nullptr && out != nullptr, "NULL argument");
    *out = nullptr;
    SNK_REQUIRE(params->orb.n_levels >= 1 && params->orb.n_levels <= 8 && params->orb.scale_factor > 1.0f, "orb parameters");
    SNK_REQUIRE(params->bounds.max_x > params->bounds.min_x && params->bounds.max_y > params->bounds.min_y, "empty feature-grid bounds");
    SNK_HIP_CHECK(hipSetDevice(device));
    snk_frontend* f = new snk_frontend();
    f->device       = device;
    f->par          = *params;
    f->n_img        = params->stereo ? 2 : 1;
    if (hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess)
    {
        set_error("hipStreamCreateWithFlags failed");
        delete f;
        return SNK_ERR_HIP;
    }
    int rc = snk_orb_create(&params->orb, device, f->stream, &f->orb);
    if (rc == SNK_OK) rc = snk_matcher_create(device, f->stream, &f->mat);
    if (rc != SNK_OK)
    {
        if (f->orb) (void)snk_orb_destroy(f->orb);
        (void)hipStreamDestroy(f->stream);
        delete f;
        return rc;
    }
    // ScalePyramid::Scale(l) as the extractor defines it: scale[l] = scale[l - 1] * factor in float
    f->level_scale[0] = 1.0f;
    for (int l = 1; l < 8; ++l) f->level_scale[l] = f->level_scale[l - 1] * params->orb.scale_factor;
    grid_dims(&params->bounds, &f->cols, &f->rows);
    *out = f;
    return SNK_OK;
}

extern "C" int snk_frontend_destroy(snk_frontend* f)
{
    if (!f) return SNK_OK;
    (void)hipSetDevice(f->device);
    (void)hipStreamSynchronize(f->stream);
    drop_graph(f);
    (void)snk_orb_destroy(f->orb);
    (void)snk_matcher_destroy(f->mat);
    f->d_img.release();
    f->d_out.release();
    f->d_tmp.release();
    f->h_img.release();
    f->h_out.release();
    (void)hipStreamDestroy(f->stream);
    delete f;
    return SNK_OK;
}

static size_t up64(size_t v) { return (v + 63) & ~(size_t)63; }

static int configure(snk_frontend* f, int w, int h)
{
    int rc;
    SNK_HIP_CHECK(hipStreamSynchronize(f->stream));
    drop_graph(f);
    if ((rc = snk_orb_configure(f->orb, w, h, 2)) != SNK_OK) return rc;
    if ((rc = snk_orb_max_keypoints(f->orb, &f->cap)) != SNK_OK) return rc;
    const size_t cap = (size_t)f->cap, ni = (size_t)f->n_img;
    f->width = w; f->height = h; f->dpitch = (w + 63) & ~63;
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t o = at; at = up64(at + bytes); return o; };
    f->o_n      = take(16 * sizeof(int));                 // n[2], n_stereo
    f->o_kps    = take(ni * cap * sizeof(snk_keypoint));  // both images, extractor order
    f->o_desc_r = take(cap * 32);                         // right descriptors, extractor order
    f->o_kp64_g = take(cap * sizeof(snk_kp64));           // undistorted_keypoints, grid order
    f->o_desc_g = take(cap * 32);                         // left descriptors, grid order
    f->o_norm   = take(cap * 16);                         // normalized_points, extractor order (the host applies the permutation)
    f->o_perm   = take(cap * 4);
    f->o_cs     = take(((size_t)f->cols * f->rows + 1) * 4);
    f->o_rp     = take(cap * 4);
    f->o_dp     = take(cap * 4);                          // directly behind right_points: one fill for both
    f->out_len  = at;
    f->t_desc   = 0;
    f->t_kp64   = up64(ni * cap * 32);
    if ((rc = f->d_out.reserve(f->out_len + 64)) != SNK_OK) return rc;
    if ((rc = f->h_out.reserve(f->out_len + 64)) != SNK_OK) return rc;
    if ((rc = f->d_tmp.reserve(f->t_kp64 + ni * cap * sizeof(snk_kp64) + 64)) != SNK_OK) return rc;
    if ((rc = f->d_img.reserve(ni * (size_t)f->dpitch * h + 64)) != SNK_OK) return rc;
    if ((rc = f->h_img.reserve(ni * (size_t)f->dpitch * h + 64)) != SNK_OK) return rc;
    f->frames_seen = 0;
    return SNK_OK;
}

// everything between the upload and the download, on the handle's stream
static int enqueue_chain(snk_frontend* f)
{
    const size_t cap = (size_t)f->cap;
    char* o  = f->d_out.as<char>();
    char* t  = f->d_tmp.as<char>();
    int* d_n = reinterpret_cast<int*>(o + f->o_n);
    auto* d_kps  = reinterpret_cast<snk_keypoint*>(o + f->o_kps);
    auto* d_desc = reinterpret_cast<uint64_t*>(t + f->t_desc);
    auto* d_kp64 = reinterpret_cast<snk_kp64*>(t + f->t_kp64);
    int rc;
    // FeatureDetector::Detect, left then right (FeatureDetector.cpp:116-156): one two-image launch chain
    if ((rc = snk_orb_detect_batch_dev(f->orb, f->d_img.as<uint8_t>(), f->dpitch, (size_t)f->dpitch * f->height, f->n_img, d_kps, d_desc, d_n,
                                       f->cap)) != SNK_OK)
        return rc;
    // Frame::allocateTmp (Snake/Map/Frame.cpp:25-26): right_points and depth start at -1000
    SNK_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(o + f->o_rp), 0xC47A0000u /* -1000.0f */, (f->o_dp - f->o_rp) / 4 + cap, f->stream));
    // undistortKeypoints (Preprocess.cpp:55-77) with rect_left; Rectification::Forward of the right keypoints (:140-150) with rect_right
    if ((rc = snk_rectify_batch_dev(f->mat, &f->par.rect_left, d_kps, d_n, f->cap, 1, d_kp64, reinterpret_cast<double*>(o + f->o_norm))) != SNK_OK) return rc;
    if (f->n_img == 2 &&
        (rc = snk_rectify_batch_dev(f->mat, &f->par.rect_right, d_kps + cap, d_n + 1, f->cap, 1, d_kp64 + cap, nullptr)) != SNK_OK)
        return rc;
    // computeFeatureGrid (Preprocess.cpp:244-266): permutation, cell starts, undistorted keypoints and descriptors in grid order
    if ((rc = snk_feature_grid_batch_dev(f->mat, &f->par.bounds, d_kp64, d_desc, d_n, f->cap, 1, reinterpret_cast<snk_kp64*>(o + f->o_kp64_g),
                                         reinterpret_cast<uint64_t*>(o + f->o_desc_g), reinterpret_cast<int32_t*>(o + f->o_perm),
                                         reinterpret_cast<int32_t*>(o + f->o_cs))) != SNK_OK)
        return rc;
    if (f->n_img == 2)
    {
        // the right descriptors go back in extractor order (frame.descriptors_right)
        SNK_HIP_CHECK(hipMemcpyAsync(o + f->o_desc_r, d_desc + cap * 4, cap * 32, hipMemcpyDeviceToDevice, f->stream));
        // StereoMatching (Preprocess.cpp:122-242): left in grid order, right in extractor order (:41-49)
        if ((rc = snk_stereo_match_batch_dev(f->mat, reinterpret_cast<const snk_kp64*>(o + f->o_kp64_g), reinterpret_cast<const uint64_t*>(o + f->o_desc_g),
                                             d_n, f->cap, d_kp64 + cap, d_desc + cap * 4, d_n + 1, f->cap, 1, f->par.bf, f->level_scale,
                                             f->par.orb.n_levels, f->par.relaxed_stereo, reinterpret_cast<float*>(o + f->o_rp),
                                             reinterpret_cast<float*>(o + f->o_dp), d_n + 2)) != SNK_OK)
            return rc;
    }
    return SNK_OK;
}

extern "C" int snk_frontend_process(snk_frontend* f, const uint8_t* left, int pitch_left, const uint8_t* right, int pitch_right, int width,
                                    int height, snk_frontend_frame* out)
{
    SNK_REQUIRE(f != nullptr && out != nullptr, "NULL argument");
    out->n = out->n_right = out->n_stereo = 0;
    SNK_REQUIRE(left != nullptr && width >= 1 && height >= 1 && pitch_left >= width, "bad left image");
    SNK_REQUIRE(f->n_img == 1 || (right != nullptr && pitch_right >= width), "bad right image");
    SNK_REQUIRE(out->capacity >= 0, "capacity");
    SNK_HIP_CHECK(hipSetDevice(f->device));
    int rc;
    if (f->width != width || f->height != height)
        if ((rc = configure(f, width, height)) != SNK_OK) return rc;
    // the two images into the pinned staging buffer, ONE upload
    const size_t plane = (size_t)f->dpitch * height;
    uint8_t* hi        = f->h_img.as<uint8_t>();
    for (int r = 0; r < height; ++r) memcpy(hi + (size_t)r * f->dpitch, left + (size_t)r * pitch_left, (size_t)width);
    if (f->n_img == 2)
        for (int r = 0; r < height; ++r) memcpy(hi + plane + (size_t)r * f->dpitch, right + (size_t)r * pitch_right, (size_t)width);
    SNK_HIP_CHECK(hipMemcpyAsync(f->d_img.p, hi, plane * f->n_img, hipMemcpyHostToDevice, f->stream));

    static const bool no_graph = getenv("SNK_FRONTEND_NO_GRAPH") != nullptr;
    const int key              = definition(DEF_IROUND_MODE) | (definition(DEF_ORB_RESPONSE) << 4);
    if (f->last_key != key)
    {
        drop_graph(f);
        f->frames_seen = 0;
        f->last_key    = key;
    }
    bool launched = false;
    if (!no_graph && f->graph)
    {
        SNK_HIP_CHECK(hipGraphLaunch(f->graph, f->stream));
        launched = true;
    }
    else if (!no_graph && !f->graph_failed && f->frames_seen >= 1)
    {
        // second frame of the configuration: record the chain (every scratch buffer has its size from the first frame)
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(f->stream, hipStreamCaptureModeRelaxed) == hipSuccess)
        {
            rc                = enqueue_chain(f);
            const hipError_t e = hipStreamEndCapture(f->stream, &g);
            hipGraphExec_t ex = nullptr;
            if (rc == SNK_OK && e == hipSuccess && g != nullptr && hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess && ex != nullptr)
            {
                f->graph     = ex;
                f->graph_key = key;
            }
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            if (!f->graph)
            {
                // not retried frame after frame (begin capture, the whole chain, end capture, instantiate -- each time); configure() and
                // a changed definition key reset the flag through drop_graph
                f->graph_failed = true;
                if (getenv("SNK_DEBUG")) fprintf(stderr, "snake_hip: front-end graph capture failed (chain rc %d, %s); plain launches\n", rc, hipGetErrorString(e));
            }
            if (f->graph)
            {
                SNK_HIP_CHECK(hipGraphLaunch(f->graph, f->stream));
                launched = true;
            }
        }
        else
        {
            (void)hipGetLastError();
            f->graph_failed = true;
        }
    }
    if (!launched && (rc = enqueue_chain(f)) != SNK_OK) return rc;
    ++f->frames_seen;
    // ONE download, ONE synchronisation
    SNK_HIP_CHECK(hipMemcpyAsync(f->h_out.p, f->d_out.p, f->out_len, hipMemcpyDeviceToHost, f->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(f->stream));

    const char* h  = f->h_out.as<char>();
    const int* hn  = reinterpret_cast<const int*>(h + f->o_n);
    const int n    = hn[0], nr = f->n_img == 2 ? hn[1] : 0;
    out->n         = n;
    out->n_right   = nr;
    out->n_stereo  = f->n_img == 2 ? hn[2] : 0;
    out->cols      = f->cols;
    out->rows      = f->rows;
    if (n > out->capacity || nr > out->capacity)
    {
        set_error("capacity %d too small for %d / %d keypoints (see snk_frontend_max_keypoints)", out->capacity, n, nr);
        return SNK_ERR_CAPACITY;
    }
    const size_t cap = (size_t)f->cap;
    const auto* kps  = reinterpret_cast<const snk_keypoint*>(h + f->o_kps);
    const int* perm  = reinterpret_cast<const int*>(h + f->o_perm);
    // computeFeatureGrid's scatter of the arrays that stayed in extractor order (Preprocess.cpp:254-260)
    if (out->keypoints)
        for (int i = 0; i < n; ++i) out->keypoints[perm[i]] = kps[i];
    if (out->normalized_points)
    {
        const double* nm = reinterpret_cast<const double*>(h + f->o_norm);
        for (int i = 0; i < n; ++i) out->normalized_points[perm[i]][0] = nm[2 * i], out->normalized_points[perm[i]][1] = nm[2 * i + 1];
    }
    if (out->descriptors && n) memcpy(out->descriptors, h + f->o_desc_g, (size_t)n * 32);
    if (out->undistorted_keypoints && n) memcpy(out->undistorted_keypoints, h + f->o_kp64_g, (size_t)n * sizeof(snk_kp64));
    if (out->permutation && n) memcpy(out->permutation, perm, (size_t)n * 4);
    if (out->cell_start) memcpy(out->cell_start, h + f->o_cs, ((size_t)f->cols * f->rows + 1) * 4);
    if (out->right_points && n) memcpy(out->right_points, h + f->o_rp, (size_t)n * 4);
    if (out->depth && n) memcpy(out->depth, h + f->o_dp, (size_t)n * 4);
    if (out->keypoints_right && nr) memcpy(out->keypoints_right, kps + cap, (size_t)nr * sizeof(snk_keypoint));
    if (out->descriptors_right && nr) memcpy(out->descriptors_right, h + f->o_desc_r, (size_t)nr * 32);
    return SNK_OK;
}

extern "C" int snk_frontend_max_keypoints(snk_frontend* f, int width, int height, int* out)
{
    SNK_REQUIRE(f != nullptr && out != nullptr && width >= 1 && height >= 1, "bad arguments");
    SNK_HIP_CHECK(hipSetDevice(f->device));
    int rc;
    if (f->width != width || f->height != height)
        if ((rc = configure(f, width, height)) != SNK_OK) return rc;
    *out = f->cap;
    return SNK_OK;
}

extern "C" int snk_frontend_grid_dims(const snk_frontend* f, int* cols, int* rows)
{
    SNK_REQUIRE(f != nullptr && cols != nullptr && rows != nullptr, "NULL argument");
    *cols = f->cols;
    *rows = f->rows;
    return SNK_OK;
}
