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
//
// Pipelined form (round 5): snk_frontend_submit / snk_frontend_collect.  The reference overlaps its per-frame stages through blocking
// single-slot queues (SynchronizedSlot<FramePtr> output_buffer, Snake/Preprocess/FeatureDetector.h:39, Preprocess.h:36): while
// Preprocess works on frame k, FeatureDetection already extracts frame k + 1.  Here a handle owns `depth` SLOTS (default 3), each a
// complete private context -- stream, extractor, matcher scratch, device blocks, pinned staging, hipGraph, completion event.  submit
// stages the two images, enqueues upload -> chain -> download on the slot's stream, records the event and returns; collect waits for
// the OLDEST frame's event and hands its results out (frames come back in submission order).  Frames in different slots share
// nothing, so their uploads, kernels and downloads overlap on the device: one frame's 14 small launches leave most of the chip
// idle.  submit blocks while all slots are occupied, collect while none is (the semantics of SynchronizedSlot::set / get); the two
// may be called from different threads (one submitting thread, one collecting thread).  Results are bit for bit those of
// snk_frontend_process: same kernels, same order inside a frame.
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "matcher_handle.hpp"

using namespace snk;

// Saiga::FeatureGridBounds2<double, 20>: 20-px cells over the bounds (the same rule as track.hip's grid_dims)
static void grid_dims(const snk_grid_bounds* b, int* cols, int* rows)
{
    const int c = (int)ceil((b->max_x - b->min_x) / 20.0), r = (int)ceil((b->max_y - b->min_y) / 20.0);
    *cols = c < 1 ? 1 : c;
    *rows = r < 1 ? 1 : r;
}

namespace
{
constexpr int MAX_DEPTH = 8;

// Layout of a slot's device block that goes back to the host (d_out / h_out) and of its scratch block (d_tmp) for one image size.
// Every slot keeps ITS OWN copy (written by the submitting thread before the frame is published, read by the collecting thread after):
// the handle's copy is only the size the submitter last configured, read and written under the handle's mutex.
struct FLayout
{
    int width = 0, height = 0, dpitch = 0, cap = 0;
    size_t o_n = 0, o_kps = 0, o_desc_x = 0, o_desc_r = 0, o_kp64_g = 0, o_desc_g = 0, o_norm = 0, o_perm = 0, o_cs = 0, o_rp = 0, o_dp = 0, out_len = 0;
    size_t t_desc = 0, t_kp64 = 0;  // d_tmp: descriptors in extractor order (2 images) | rectified keypoints (2 images)
};

// everything one frame in flight needs; nothing is shared between slots
struct Slot
{
    hipStream_t stream = nullptr;
    snk_orb* orb       = nullptr;
    snk_matcher* mat   = nullptr;
    FLayout lay;                // what this slot's extractor and blocks are sized for (lay.width == 0: nothing yet)
    int img_pitch = 0;          // row pitch of the images in d_img for the frame being enqueued (dpitch, or the caller's for a direct upload)
    int graph_pitch = 0;        // ... of the recorded launches
    DevBuf d_img, d_out, d_tmp;
    HostBuf h_img, h_out;
    hipGraphExec_t graph = nullptr;
    int graph_key        = -1;     // definition values the recorded launches depend on
    int last_key         = -1;     // ... of the previous frame (a change runs one frame uncaptured: the extractor may size new scratch)
    bool graph_failed    = false;  // a capture / instantiation failed for this configuration: plain launches from then on
    int frames_seen      = 0;      // of the current configuration
    hipEvent_t done      = nullptr;  // recorded behind the download of a submitted frame
};
}  // namespace

struct snk_frontend
{
    int device = 0;
    snk_frontend_params par{};
    int cols = 0, rows = 0, n_img = 2;  // constants of the handle
    float level_scale[8] = {};
    FLayout lay;               // the image size last configured (guarded by mu)
    std::vector<Slot*> slots;  // slots[0] also serves snk_frontend_process
    // the ring of submitted frames: frame number q lives in slot q % depth
    std::mutex mu;
    std::condition_variable cv;
    int depth = 3;  // measured on MI355X (profiles/r05): 2 -> 9.2 k, 3 -> 12.5 k, 4 -> 11.3 k stereo frames/s through the C++ adaptor
    unsigned long long submitted = 0, collected = 0;
};

static void drop_graph(Slot* s)
{
    if (s->graph) (void)hipGraphExecDestroy(s->graph);
    s->graph        = nullptr;
    s->graph_key    = -1;
    s->graph_failed = false;
}

static void destroy_slot(Slot* s)
{
    if (!s) return;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    drop_graph(s);
    if (s->orb) (void)snk_orb_destroy(s->orb);
    if (s->mat) (void)snk_matcher_destroy(s->mat);
    s->d_img.release();
    s->d_out.release();
    s->d_tmp.release();
    s->h_img.release();
    s->h_out.release();
    if (s->done) (void)hipEventDestroy(s->done);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

static int create_slot(snk_frontend* f, Slot** out)
{
    Slot* s = new Slot();
    int rc  = SNK_OK;
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&s->done, hipEventDisableTiming) != hipSuccess)
    {
        set_error("front-end slot: stream / event creation failed");
        rc = SNK_ERR_HIP;
    }
    if (rc == SNK_OK) rc = snk_orb_create(&f->par.orb, f->device, s->stream, &s->orb);
    if (rc == SNK_OK) rc = snk_matcher_create(f->device, s->stream, &s->mat);
    // HIP binds a stream to one of its few hardware queues when the stream is first USED, taking the least loaded one: slots that are
    // created and touched back to back get different queues.  Left to their first frame, the slots were bound after every other handle of
    // the process had taken a queue, and two slots could share one -- their frames then run one behind the other (9.8 k instead of 14.4 k
    // frames/s on the same code, profiles/r05/r05s_latencies.log).  One recorded event is enough to bind the queue.
    if (rc == SNK_OK && (hipEventRecord(s->done, s->stream) != hipSuccess || hipEventSynchronize(s->done) != hipSuccess))
    {
        set_error("front-end slot: the stream could not be started");
        rc = SNK_ERR_HIP;
    }
    if (rc != SNK_OK)
    {
        destroy_slot(s);
        return rc;
    }
    *out = s;
    return SNK_OK;
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
    // every slot of the default depth now (see create_slot: consecutive creation = different hardware queues); a larger depth set later
    // adds its slots then
    for (int i = 0; i < f->depth; ++i)
    {
        Slot* s      = nullptr;
        const int rc = create_slot(f, &s);
        if (rc != SNK_OK)
        {
            for (Slot* q : f->slots) destroy_slot(q);
            delete f;
            return rc;
        }
        f->slots.push_back(s);
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
    for (Slot* s : f->slots) destroy_slot(s);
    delete f;
    return SNK_OK;
}

static size_t up64(size_t v) { return (v + 63) & ~(size_t)63; }

// the block layout of an image size (the same for every slot), into the slot's own copy
static int layout(snk_frontend* f, Slot* s, int w, int h)
{
    int rc;
    FLayout y;
    if ((rc = snk_orb_configure(s->orb, w, h, 2)) != SNK_OK) return rc;
    if ((rc = snk_orb_max_keypoints(s->orb, &y.cap)) != SNK_OK) return rc;
    const size_t cap = (size_t)y.cap, ni = (size_t)f->n_img;
    y.width = w; y.height = h; y.dpitch = (w + 63) & ~63;
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t o = at; at = up64(at + bytes); return o; };
    y.o_n      = take(16 * sizeof(int));                 // n[2], n_stereo
    y.o_kps    = take(ni * cap * sizeof(snk_keypoint));  // both images, extractor order
    y.o_desc_x = take(2 * cap * 32);                     // descriptors in extractor order, image-major as the extractor writes them: the left half is
    y.o_desc_r = y.o_desc_x + cap * 32;                  // scratch that rides along in the download, the right half IS frame.descriptors_right
    y.o_kp64_g = take(cap * sizeof(snk_kp64));           // undistorted_keypoints, grid order
    y.o_desc_g = take(cap * 32);                         // left descriptors, grid order
    y.o_norm   = take(cap * 16);                         // normalized_points, extractor order (the host applies the permutation)
    y.o_perm   = take(cap * 4);
    y.o_cs     = take(((size_t)f->cols * f->rows + 1) * 4);
    y.o_rp     = take(cap * 4);
    y.o_dp     = take(cap * 4);                          // directly behind right_points: one fill for both
    y.out_len  = at;
    y.t_desc   = 0;
    y.t_kp64   = up64(ni * cap * 32);
    s->lay     = y;
    return SNK_OK;
}

// size slot s for w x h images (its own extractor, blocks and staging)
static int configure_slot(snk_frontend* f, Slot* s, int w, int h)
{
    int rc;
    SNK_HIP_CHECK(hipStreamSynchronize(s->stream));
    drop_graph(s);
    s->lay.width = 0;  // not configured until everything below has succeeded
    if ((rc = layout(f, s, w, h)) != SNK_OK) return rc;
    const FLayout y  = s->lay;
    s->lay.width     = 0;
    const size_t cap = (size_t)y.cap, ni = (size_t)f->n_img;
    if ((rc = s->d_out.reserve(y.out_len + 64)) != SNK_OK) return rc;
    if ((rc = s->h_out.reserve(y.out_len + 64)) != SNK_OK) return rc;
    if ((rc = s->d_tmp.reserve(y.t_kp64 + ni * cap * sizeof(snk_kp64) + 64)) != SNK_OK) return rc;
    if ((rc = s->d_img.reserve(ni * (size_t)y.dpitch * h + 64)) != SNK_OK) return rc;
    if ((rc = s->h_img.reserve(ni * (size_t)y.dpitch * h + 64)) != SNK_OK) return rc;
    s->lay         = y;
    s->frames_seen = 0;
    s->last_key    = -1;
    return SNK_OK;
}

// everything between the upload and the download, on the slot's stream
static int enqueue_chain(snk_frontend* f, Slot* s)
{
    const FLayout& y = s->lay;
    const size_t cap = (size_t)y.cap;
    char* o  = s->d_out.as<char>();
    char* t  = s->d_tmp.as<char>();
    int* d_n = reinterpret_cast<int*>(o + y.o_n);
    auto* d_kps  = reinterpret_cast<snk_keypoint*>(o + y.o_kps);
    // descriptors of both images in extractor order, image-major with stride cap * 4 words: a stereo handle keeps them inside the output
    // block (o_desc_x, its second half = o_desc_r), so that no device-to-device copy stands between the extractor and the download
    auto* d_desc = f->n_img == 2 ? reinterpret_cast<uint64_t*>(o + y.o_desc_x) : reinterpret_cast<uint64_t*>(t + y.t_desc);
    auto* d_kp64 = reinterpret_cast<snk_kp64*>(t + y.t_kp64);
    int rc;
    // FeatureDetector::Detect, left then right (FeatureDetector.cpp:116-156): one two-image launch chain
    if ((rc = snk_orb_detect_batch_dev(s->orb, s->d_img.as<uint8_t>(), s->img_pitch, (size_t)s->img_pitch * y.height, f->n_img, d_kps, d_desc, d_n,
                                       y.cap)) != SNK_OK)
        return rc;
    // Frame::allocateTmp (Snake/Map/Frame.cpp:25-26): right_points and depth start at -1000 -- a stereo handle leaves that to its
    // StereoMatching call below (whose first kernel does it: one launch less in a chain of short launches), a mono handle fills here
    if (f->n_img != 2)
        SNK_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(o + y.o_rp), 0xC47A0000u /* -1000.0f */, (y.o_dp - y.o_rp) / 4 + cap, s->stream));
    // undistortKeypoints (Preprocess.cpp:55-77) with rect_left; Rectification::Forward of the right keypoints (:140-150) with rect_right
    if (f->n_img == 2)
    {
        if ((rc = rectify_pair_dev(s->mat, &f->par.rect_left, &f->par.rect_right, d_kps, d_n, y.cap, d_kp64, reinterpret_cast<double*>(o + y.o_norm))) != SNK_OK)
            return rc;
    }
    else if ((rc = snk_rectify_batch_dev(s->mat, &f->par.rect_left, d_kps, d_n, y.cap, 1, d_kp64, reinterpret_cast<double*>(o + y.o_norm))) != SNK_OK)
        return rc;
    // computeFeatureGrid (Preprocess.cpp:244-266): permutation, cell starts, undistorted keypoints and descriptors in grid order
    if ((rc = snk_feature_grid_batch_dev(s->mat, &f->par.bounds, d_kp64, d_desc, d_n, y.cap, 1, reinterpret_cast<snk_kp64*>(o + y.o_kp64_g),
                                         reinterpret_cast<uint64_t*>(o + y.o_desc_g), reinterpret_cast<int32_t*>(o + y.o_perm),
                                         reinterpret_cast<int32_t*>(o + y.o_cs))) != SNK_OK)
        return rc;
    if (f->n_img == 2)
    {
        // the right descriptors go back in extractor order (frame.descriptors_right): the extractor wrote them where the download takes
        // them from (d_desc's second half IS o_desc_r, see layout())
        // StereoMatching (Preprocess.cpp:122-242): left in grid order, right in extractor order (:41-49)
        if ((rc = stereo_match_batch_dev_impl(s->mat, reinterpret_cast<const snk_kp64*>(o + y.o_kp64_g), reinterpret_cast<const uint64_t*>(o + y.o_desc_g),
                                             d_n, y.cap, d_kp64 + cap, d_desc + cap * 4, d_n + 1, y.cap, 1, f->par.bf, f->level_scale,
                                             f->par.orb.n_levels, f->par.relaxed_stereo, reinterpret_cast<float*>(o + y.o_rp),
                                             reinterpret_cast<float*>(o + y.o_dp), d_n + 2, /*prefill=*/true)) != SNK_OK)
            return rc;
    }
    return SNK_OK;
}

// The images onto the device, the chain (hipGraph from the slot's second frame on), the download -- all enqueued on the slot's
// stream, no wait.  direct = false: the two images are staged into the slot's pinned buffer (row pitch dpitch) and go up with ONE
// copy; the caller's memory is free again when this returns.  direct = true (snk_frontend_submit_pinned): the caller's buffers are
// the source of the upload (one copy when the right image follows the left one in memory, two otherwise), no staging copy on the
// host -- for callers whose image buffers are pinned and stay untouched until the frame has been collected.
static int enqueue_frame(snk_frontend* f, Slot* s, const uint8_t* left, int pitch_left, const uint8_t* right, int pitch_right, int width, int height,
                         bool direct)
{
    int rc;
    if (s->lay.width != width || s->lay.height != height)
        if ((rc = configure_slot(f, s, width, height)) != SNK_OK) return rc;
    const FLayout& y = s->lay;
    // SNK_FRONTEND_TIMING=1 (diagnostic): mean host microseconds per part of this function, printed every 256 frames
    static const bool timing = getenv("SNK_FRONTEND_TIMING") != nullptr;
    static double t_acc[5]  = {0, 0, 0, 0, 0};
    static int t_n          = 0;
    auto now                = [] { return std::chrono::steady_clock::now(); };
    auto t_prev             = now();
    auto lap                = [&](int i)
    {
        if (!timing) return;
        const auto t = now();
        t_acc[i] += std::chrono::duration<double, std::micro>(t - t_prev).count();
        t_prev = t;
    };
    uint8_t* di = s->d_img.as<uint8_t>();
    if (direct && pitch_left % 4 == 0 && pitch_left <= y.dpitch && (f->n_img == 1 || pitch_right == pitch_left))
    {
        // rows keep the caller's pitch on the device (the extractor takes any pitch; a multiple of four keeps its aligned loads):
        // plain 1-D copies straight out of the caller's memory
        const size_t plane = (size_t)pitch_left * height;
        s->img_pitch       = pitch_left;
        lap(0);
        if (f->n_img == 2 && right == left + plane)
            SNK_HIP_CHECK(hipMemcpyAsync(di, left, 2 * plane, hipMemcpyHostToDevice, s->stream));
        else
        {
            SNK_HIP_CHECK(hipMemcpyAsync(di, left, plane - (size_t)(pitch_left - width), hipMemcpyHostToDevice, s->stream));
            if (f->n_img == 2)
                SNK_HIP_CHECK(hipMemcpyAsync(di + plane, right, plane - (size_t)(pitch_left - width), hipMemcpyHostToDevice, s->stream));
        }
    }
    else if (direct)
    {
        // odd pitches: a 2-D copy per image into the dpitch layout, still without a host-side copy
        const size_t plane = (size_t)y.dpitch * height;
        s->img_pitch       = y.dpitch;
        lap(0);
        SNK_HIP_CHECK(hipMemcpy2DAsync(di, (size_t)y.dpitch, left, (size_t)pitch_left, (size_t)width, (size_t)height, hipMemcpyHostToDevice, s->stream));
        if (f->n_img == 2)
            SNK_HIP_CHECK(hipMemcpy2DAsync(di + plane, (size_t)y.dpitch, right, (size_t)pitch_right, (size_t)width, (size_t)height, hipMemcpyHostToDevice, s->stream));
    }
    else
    {
        // the two images into the pinned staging buffer, ONE upload
        const size_t plane = (size_t)y.dpitch * height;
        uint8_t* hi        = s->h_img.as<uint8_t>();
        s->img_pitch       = y.dpitch;
        if (pitch_left == y.dpitch)
            memcpy(hi, left, plane - (size_t)(y.dpitch - width));
        else
            for (int r = 0; r < height; ++r) memcpy(hi + (size_t)r * y.dpitch, left + (size_t)r * pitch_left, (size_t)width);
        if (f->n_img == 2)
        {
            if (pitch_right == y.dpitch)
                memcpy(hi + plane, right, plane - (size_t)(y.dpitch - width));
            else
                for (int r = 0; r < height; ++r) memcpy(hi + plane + (size_t)r * y.dpitch, right + (size_t)r * pitch_right, (size_t)width);
        }
        lap(0);
        SNK_HIP_CHECK(hipMemcpyAsync(di, hi, plane * f->n_img, hipMemcpyHostToDevice, s->stream));
    }
    lap(1);

    static const bool no_graph = getenv("SNK_FRONTEND_NO_GRAPH") != nullptr;
    const int key              = definition(DEF_IROUND_MODE) | (definition(DEF_ORB_RESPONSE) << 4);
    if (s->last_key != key)
    {
        drop_graph(s);
        s->frames_seen = 0;
        s->last_key    = key;
    }
    if (s->graph && s->graph_pitch != s->img_pitch) drop_graph(s);  // the row pitch is an argument of the recorded launches
    bool launched = false;
    if (!no_graph && s->graph)
    {
        SNK_HIP_CHECK(hipGraphLaunch(s->graph, s->stream));
        launched = true;
    }
    else if (!no_graph && !s->graph_failed && s->frames_seen >= 1)
    {
        // second frame of the configuration: record the chain (every scratch buffer has its size from the first frame)
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(s->stream, hipStreamCaptureModeRelaxed) == hipSuccess)
        {
            rc                = enqueue_chain(f, s);
            const hipError_t e = hipStreamEndCapture(s->stream, &g);
            hipGraphExec_t ex = nullptr;
            if (rc == SNK_OK && e == hipSuccess && g != nullptr && hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess && ex != nullptr)
            {
                s->graph       = ex;
                s->graph_key   = key;
                s->graph_pitch = s->img_pitch;
            }
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            if (!s->graph)
            {
                // not retried frame after frame (begin capture, the whole chain, end capture, instantiate -- each time); configure_slot and
                // a changed definition key reset the flag through drop_graph
                s->graph_failed = true;
                if (getenv("SNK_DEBUG")) fprintf(stderr, "snake_hip: front-end graph capture failed (chain rc %d, %s); plain launches\n", rc, hipGetErrorString(e));
            }
            if (s->graph)
            {
                SNK_HIP_CHECK(hipGraphLaunch(s->graph, s->stream));
                launched = true;
            }
        }
        else
        {
            (void)hipGetLastError();
            s->graph_failed = true;
        }
    }
    if (!launched && (rc = enqueue_chain(f, s)) != SNK_OK) return rc;
    ++s->frames_seen;
    lap(2);
    SNK_HIP_CHECK(hipMemcpyAsync(s->h_out.p, s->d_out.p, y.out_len, hipMemcpyDeviceToHost, s->stream));
    lap(3);
    if (timing && ++t_n % 256 == 0)
    {
        fprintf(stderr, "[frontend timing] per frame, us: stage %.1f | upload enqueue %.1f | chain / graph launch %.1f | download enqueue %.1f\n", t_acc[0] / 256,
                t_acc[1] / 256, t_acc[2] / 256, t_acc[3] / 256);
        t_acc[0] = t_acc[1] = t_acc[2] = t_acc[3] = 0;
    }
    return SNK_OK;
}

// a finished frame's pinned block -> the caller's arrays (the slot's own layout: nothing of the handle that another thread may write)
static int unpack(snk_frontend* f, Slot* s, snk_frontend_frame* out)
{
    const FLayout& y = s->lay;
    const char* h  = s->h_out.as<char>();
    const int* hn  = reinterpret_cast<const int*>(h + y.o_n);
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
    const size_t cap = (size_t)y.cap;
    const auto* kps  = reinterpret_cast<const snk_keypoint*>(h + y.o_kps);
    const int* perm  = reinterpret_cast<const int*>(h + y.o_perm);
    // computeFeatureGrid's scatter of the arrays that stayed in extractor order (Preprocess.cpp:254-260)
    if (out->keypoints)
        for (int i = 0; i < n; ++i) out->keypoints[perm[i]] = kps[i];
    if (out->normalized_points)
    {
        const double* nm = reinterpret_cast<const double*>(h + y.o_norm);
        for (int i = 0; i < n; ++i) out->normalized_points[perm[i]][0] = nm[2 * i], out->normalized_points[perm[i]][1] = nm[2 * i + 1];
    }
    if (out->descriptors && n) memcpy(out->descriptors, h + y.o_desc_g, (size_t)n * 32);
    if (out->undistorted_keypoints && n) memcpy(out->undistorted_keypoints, h + y.o_kp64_g, (size_t)n * sizeof(snk_kp64));
    if (out->permutation && n) memcpy(out->permutation, perm, (size_t)n * 4);
    if (out->cell_start) memcpy(out->cell_start, h + y.o_cs, ((size_t)f->cols * f->rows + 1) * 4);
    if (out->right_points && n) memcpy(out->right_points, h + y.o_rp, (size_t)n * 4);
    if (out->depth && n) memcpy(out->depth, h + y.o_dp, (size_t)n * 4);
    if (out->keypoints_right && nr) memcpy(out->keypoints_right, kps + cap, (size_t)nr * sizeof(snk_keypoint));
    if (out->descriptors_right && nr) memcpy(out->descriptors_right, h + y.o_desc_r, (size_t)nr * 32);
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
    {
        std::lock_guard<std::mutex> lock(f->mu);
        SNK_REQUIRE(f->submitted == f->collected, "frames submitted with snk_frontend_submit are still in flight: collect them first");
    }
    SNK_HIP_CHECK(hipSetDevice(f->device));
    Slot* s = f->slots[0];
    int rc;
    if ((rc = enqueue_frame(f, s, left, pitch_left, right, pitch_right, width, height, false)) != SNK_OK) return rc;
    {
        std::lock_guard<std::mutex> lock(f->mu);
        f->lay = s->lay;
    }
    SNK_HIP_CHECK(hipStreamSynchronize(s->stream));  // ONE synchronisation
    return unpack(f, s, out);
}

extern "C" int snk_frontend_set_depth(snk_frontend* f, int depth)
{
    SNK_REQUIRE(f != nullptr, "frontend is NULL");
    SNK_REQUIRE(depth >= 1 && depth <= MAX_DEPTH, "depth must be 1..8");
    std::lock_guard<std::mutex> lock(f->mu);
    SNK_REQUIRE(f->submitted == f->collected, "frames are in flight");
    SNK_HIP_CHECK(hipSetDevice(f->device));
    while ((int)f->slots.size() < depth)
    {
        Slot* s      = nullptr;
        const int rc = create_slot(f, &s);
        if (rc != SNK_OK) return rc;
        f->slots.push_back(s);
    }
    f->depth     = depth;
    f->submitted = f->collected = 0;  // frame q lives in slot q % depth: restart the numbering with the new modulus
    return SNK_OK;
}

static int submit_impl(snk_frontend* f, const uint8_t* left, int pitch_left, const uint8_t* right, int pitch_right, int width, int height, bool direct)
{
    SNK_REQUIRE(f != nullptr, "frontend is NULL");
    SNK_REQUIRE(left != nullptr && width >= 1 && height >= 1 && pitch_left >= width, "bad left image");
    SNK_REQUIRE(f->n_img == 1 || (right != nullptr && pitch_right >= width), "bad right image");
    SNK_HIP_CHECK(hipSetDevice(f->device));
    int si;
    {
        // SynchronizedSlot::set: wait for a free slot (the collector frees the oldest)
        std::unique_lock<std::mutex> lock(f->mu);
        f->cv.wait(lock, [&] { return f->submitted - f->collected < (unsigned long long)f->depth; });
        // frames of one image size at a time (checked behind the wait: what is in flight NOW decides)
        SNK_REQUIRE(f->submitted == f->collected || (f->lay.width == width && f->lay.height == height),
                    "image size changed while frames are in flight: collect them first");
        si = (int)(f->submitted % (unsigned long long)f->depth);
        while ((int)f->slots.size() <= si)
        {
            Slot* s = nullptr;
            const int rc = create_slot(f, &s);
            if (rc != SNK_OK) return rc;
            f->slots.push_back(s);
        }
    }
    // the slot is this thread's until `submitted` moves: the collector only touches slots of frames < submitted
    Slot* s = f->slots[(size_t)si];
    int rc;
    if ((rc = enqueue_frame(f, s, left, pitch_left, right, pitch_right, width, height, direct)) != SNK_OK)
    {
        (void)hipStreamSynchronize(s->stream);  // whatever was enqueued must not run into the slot's next use
        return rc;
    }
    SNK_HIP_CHECK(hipEventRecord(s->done, s->stream));
    {
        std::lock_guard<std::mutex> lock(f->mu);
        f->lay = s->lay;
        ++f->submitted;
    }
    f->cv.notify_all();
    return SNK_OK;
}

extern "C" int snk_frontend_submit(snk_frontend* f, const uint8_t* left, int pitch_left, const uint8_t* right, int pitch_right, int width,
                                   int height)
{
    return submit_impl(f, left, pitch_left, right, pitch_right, width, height, false);
}

extern "C" int snk_frontend_submit_pinned(snk_frontend* f, const uint8_t* left, int pitch_left, const uint8_t* right, int pitch_right, int width,
                                          int height)
{
    return submit_impl(f, left, pitch_left, right, pitch_right, width, height, true);
}

// caller-owned image memory the upload engine reads directly (Snake/Preprocess/Input.h:48: the Input thread owns its image buffers)
extern "C" int snk_pinned_alloc(size_t bytes, void** out)
{
    SNK_REQUIRE(out != nullptr && bytes > 0, "bad arguments");
    *out = nullptr;
    SNK_HIP_CHECK(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return SNK_OK;
}

extern "C" int snk_pinned_free(void* p)
{
    if (p) SNK_HIP_CHECK(hipHostFree(p));
    return SNK_OK;
}

// wait for the oldest submitted frame to EXIST (not to finish) and report what its arrays need: the collecting thread sizes its
// buffers from this, never from what the submitting thread is doing
extern "C" int snk_frontend_peek(snk_frontend* f, int timeout_ms, int* width, int* height, int* capacity)
{
    SNK_REQUIRE(f != nullptr, "frontend is NULL");
    std::unique_lock<std::mutex> lock(f->mu);
    auto ready = [&] { return f->collected < f->submitted; };
    if (timeout_ms < 0)
        f->cv.wait(lock, ready);
    else if (!f->cv.wait_for(lock, std::chrono::milliseconds(timeout_ms), ready))
    {
        set_error("snk_frontend_peek: no frame was submitted within %d ms", timeout_ms);
        return SNK_ERR_TIMEOUT;
    }
    const Slot* s = f->slots[(size_t)(f->collected % (unsigned long long)f->depth)];
    if (width) *width = s->lay.width;
    if (height) *height = s->lay.height;
    if (capacity) *capacity = s->lay.cap;
    return SNK_OK;
}

extern "C" int snk_frontend_collect(snk_frontend* f, snk_frontend_frame* out, int timeout_ms)
{
    SNK_REQUIRE(f != nullptr && out != nullptr, "NULL argument");
    out->n = out->n_right = out->n_stereo = 0;
    SNK_REQUIRE(out->capacity >= 0, "capacity");
    SNK_HIP_CHECK(hipSetDevice(f->device));
    Slot* s;
    {
        // SynchronizedSlot::get: wait for a submitted frame (timeout_ms < 0: for as long as it takes, 0: do not wait)
        std::unique_lock<std::mutex> lock(f->mu);
        auto ready = [&] { return f->collected < f->submitted; };
        if (timeout_ms < 0)
            f->cv.wait(lock, ready);
        else if (!f->cv.wait_for(lock, std::chrono::milliseconds(timeout_ms), ready))
        {
            set_error("snk_frontend_collect: no frame was submitted within %d ms", timeout_ms);
            return SNK_ERR_TIMEOUT;
        }
        s = f->slots[(size_t)(f->collected % (unsigned long long)f->depth)];
    }
    const hipError_t e = hipEventSynchronize(s->done);  // the frame's download is behind it on the slot's stream
    int rc             = SNK_OK;
    if (e != hipSuccess)
    {
        set_error("hipEventSynchronize failed: %s", hipGetErrorString(e));
        rc = SNK_ERR_HIP;
    }
    else
        rc = unpack(f, s, out);
    {
        std::lock_guard<std::mutex> lock(f->mu);
        ++f->collected;  // the frame is consumed whatever the outcome: its slot is free again
    }
    f->cv.notify_all();
    return rc;
}

extern "C" int snk_frontend_in_flight(snk_frontend* f, int* n)
{
    SNK_REQUIRE(f != nullptr && n != nullptr, "NULL argument");
    std::lock_guard<std::mutex> lock(f->mu);
    *n = (int)(f->submitted - f->collected);
    return SNK_OK;
}

extern "C" int snk_frontend_max_keypoints(snk_frontend* f, int width, int height, int* out)
{
    SNK_REQUIRE(f != nullptr && out != nullptr && width >= 1 && height >= 1, "bad arguments");
    SNK_HIP_CHECK(hipSetDevice(f->device));
    std::lock_guard<std::mutex> lock(f->mu);
    if (f->lay.width != width || f->lay.height != height)
    {
        // another size than the one last configured: slot 0 is re-sized for it, which needs the handle idle
        SNK_REQUIRE(f->submitted == f->collected, "frames of another image size are in flight");
        int rc;
        if ((rc = configure_slot(f, f->slots[0], width, height)) != SNK_OK) return rc;
        f->lay = f->slots[0]->lay;
    }
    *out = f->lay.cap;
    return SNK_OK;
}

extern "C" int snk_frontend_grid_dims(const snk_frontend* f, int* cols, int* rows)
{
    SNK_REQUIRE(f != nullptr && cols != nullptr && rows != nullptr, "NULL argument");
    *cols = f->cols;
    *rows = f->rows;
    return SNK_OK;
}
