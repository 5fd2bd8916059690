// Snake::Preprocess::undistortKeypoints / Rectification::Forward on the device
// (reference Snake/Preprocess/Preprocess.cpp:55-77,140-150): per keypoint, fp64,
// unproject (K_src) -> Gauss-Newton undistort -> rotate (R) -> divide -> project (K_dst).
// One thread per keypoint; the arithmetic is a fixed sequence of IEEE fp64 operations
// (library built with -ffp-contract=off) so it matches the scalar restatement bit for bit.
#include "matcher_handle.hpp"

namespace snk
{
namespace
{
struct RectDev
{
    double K_src[4], D[8], R[9], K_dst[4];
};

__device__ __forceinline__ void distort(const double* D, double x, double y, double& xd, double& yd, double (&J)[4])
{
    const double k1 = D[0], k2 = D[1], k3 = D[2], k4 = D[3], k5 = D[4], k6 = D[5], p1 = D[6], p2 = D[7];
    const double x2 = x * x, y2 = y * y, xy = x * y;
    const double r2 = x2 + y2, r4 = r2 * r2, r6 = r4 * r2;
    const double num = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
    const double den = 1.0 + k4 * r2 + k5 * r4 + k6 * r6;
    const double rad = num / den;
    xd = x * rad + 2.0 * p1 * xy + p2 * (r2 + 2.0 * x2);
    yd = y * rad + p1 * (r2 + 2.0 * y2) + 2.0 * p2 * xy;
    const double dnum = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4;
    const double dden = k4 + 2.0 * k5 * r2 + 3.0 * k6 * r4;
    const double drad = (dnum * den - num * dden) / (den * den);
    J[0] = rad + x * drad * 2.0 * x + 2.0 * p1 * y + p2 * (2.0 * x + 4.0 * x);
    J[1] = x * drad * 2.0 * y + 2.0 * p1 * x + p2 * 2.0 * y;
    J[2] = y * drad * 2.0 * x + p1 * 2.0 * x + 2.0 * p2 * y;
    J[3] = rad + y * drad * 2.0 * y + p1 * (2.0 * y + 4.0 * y) + 2.0 * p2 * x;
}

__device__ __forceinline__ void rectify_body(const RectDev& rc, const snk_keypoint* __restrict__ kps, const int* __restrict__ n_dev, int cap, int n_host,
                                             snk_kp64* __restrict__ out, double2* __restrict__ normalized, int b)
{
    int n       = n_dev ? n_dev[b] : n_host;
    n           = n < cap ? n : cap;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const snk_keypoint kp = kps[(size_t)b * cap + i];
    const double px = ((double)kp.x - rc.K_src[2]) / rc.K_src[0];
    const double py = ((double)kp.y - rc.K_src[3]) / rc.K_src[1];
    double x = px, y = py;
#pragma unroll 1
    for (int it = 0; it < 5; ++it)
    {
        double xd, yd, J[4];
        distort(rc.D, x, y, xd, yd, J);
        const double rx = xd - px, ry = yd - py;
        const double det = J[0] * J[3] - J[1] * J[2];
        const double dx  = (J[3] * rx - J[1] * ry) / det;
        const double dy  = (J[0] * ry - J[2] * rx) / det;
        x = x - dx;
        y = y - dy;
    }
    const double rx = rc.R[0] * x + rc.R[1] * y + rc.R[2];
    const double ry = rc.R[3] * x + rc.R[4] * y + rc.R[5];
    const double rz = rc.R[6] * x + rc.R[7] * y + rc.R[8];
    const double nx = rx / rz, ny = ry / rz;
    if (normalized) normalized[(size_t)b * cap + i] = make_double2(nx, ny);
    snk_kp64 o;
    o.x      = rc.K_dst[0] * nx + rc.K_dst[2];
    o.y      = rc.K_dst[1] * ny + rc.K_dst[3];
    o.angle  = kp.angle;
    o.octave = kp.octave;
    out[(size_t)b * cap + i] = o;
}

__global__ __launch_bounds__(256) void rectify_kernel(RectDev rc, const snk_keypoint* __restrict__ kps,
                                                      const int* __restrict__ n_dev, int cap, int n_host,
                                                      snk_kp64* __restrict__ out, double2* __restrict__ normalized)
{
    rectify_body(rc, kps, n_dev, cap, n_host, out, normalized, blockIdx.y);
}

// The two rectifications of a stereo frame in ONE launch (the one-call front-end): image blockIdx.y of the pair with its own
// rectification -- undistortKeypoints with rect_left (normalized points kept), Rectification::Forward of the right keypoints with
// rect_right (Preprocess.cpp:55-77,140-150).  Same arithmetic per keypoint as two rectify_kernel launches.
__global__ __launch_bounds__(256) void rectify_pair_kernel(RectDev rc_left, RectDev rc_right, const snk_keypoint* __restrict__ kps,
                                                           const int* __restrict__ n_dev, int cap, snk_kp64* __restrict__ out,
                                                           double2* __restrict__ normalized_left)
{
    if (blockIdx.y == 0)
        rectify_body(rc_left, kps, n_dev, cap, 0, out, normalized_left, 0);
    else
        rectify_body(rc_right, kps, n_dev, cap, 0, out, nullptr, 1);
}

// Snake::Preprocess::ComputeStereoFromRGBD (Preprocess.cpp:79-120): one thread per undistorted keypoint; the same fp64 operation
// sequence as the oracle.  status[b]: 0, or (index + 1) of the LOWEST keypoint on which the reference would abort (atomicMin).
struct RgbdDev
{
    double K[4], D[8], Kd[4], bf;
};
__global__ __launch_bounds__(256) void rgbd_stereo_kernel(RgbdDev rc, const snk_kp64* __restrict__ und, const int* __restrict__ n_dev, int cap,
                                                          int n_host, const float* __restrict__ depth_image, int w, int h, int pitch_floats,
                                                          long long image_stride, float* __restrict__ right_points, float* __restrict__ depth,
                                                          int* __restrict__ n_matches, int* __restrict__ status)
{
    const int b = blockIdx.y;
    int n       = n_dev ? n_dev[b] : n_host;
    n           = n < cap ? n : cap;
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool hit    = false;
    if (i < n)
    {
        const snk_kp64 kp = und[(size_t)b * cap + i];
        const double nx = (kp.x - rc.K[2]) / rc.K[0], ny = (kp.y - rc.K[3]) / rc.K[1];
        double dx, dy, J[4];
        distort(rc.D, nx, ny, dx, dy, J);
        const double rx = rc.Kd[0] * dx + rc.Kd[2], ry = rc.Kd[1] * dy + rc.Kd[3];
        const int x = (int)(rx + 0.5), y = (int)(ry + 0.5);
        bool bad    = !(rx + 0.5 > -1.0 && ry + 0.5 > -1.0 && rx < 2.0e9 && ry < 2.0e9) || x < 0 || y < 0 || x >= w || y >= h;
        float d     = 0.0f;
        if (!bad)
        {
            d   = depth_image[(long long)b * image_stride + (long long)y * pitch_floats + x];
            bad = !(d >= 0.0f) || !(d < 20.0f);
        }
        if (bad)
            atomicMin(&status[b], i + 1);
        else if (d > 0.0f)
        {
            depth[(size_t)b * cap + i]        = d;
            const double disparity            = rc.bf / (double)d;
            right_points[(size_t)b * cap + i] = (float)(kp.x - disparity);
            hit                               = true;
        }
        else
        {
            depth[(size_t)b * cap + i]        = -1.0f;
            right_points[(size_t)b * cap + i] = -1.0f;
        }
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&n_matches[b], __popcll(m));
}
__global__ void rgbd_init_kernel(int* n_matches, int* status, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) n_matches[b] = 0, status[b] = 0x7FFFFFFF;
}

int to_dev(const snk_rectification* r, RectDev* d)
{
    SNK_REQUIRE(r != nullptr, "rectification is NULL");
    SNK_REQUIRE(r->K_src[0] != 0.0 && r->K_src[1] != 0.0, "K_src focal length is zero");
    memcpy(d->K_src, r->K_src, sizeof(d->K_src));
    memcpy(d->D, r->D_src, sizeof(d->D));
    memcpy(d->R, r->R, sizeof(d->R));
    memcpy(d->K_dst, r->K_dst, sizeof(d->K_dst));
    return SNK_OK;
}
}  // namespace
}  // namespace snk

using namespace snk;

extern "C" {

int snk_rectify_batch_dev(snk_matcher* m, const snk_rectification* rect, const snk_keypoint* kps_dev,
                          const int32_t* n_dev, int cap, int batch, snk_kp64* out_dev, double* normalized_dev)
{
    SNK_REQUIRE(m != nullptr, "matcher is NULL");
    SNK_REQUIRE(batch >= 0 && cap >= 0, "bad sizes");
    SNK_REQUIRE(kps_dev && n_dev && out_dev, "NULL device buffer");
    RectDev rc;
    int st = to_dev(rect, &rc);
    if (st != SNK_OK) return st;
    if (batch == 0 || cap == 0) return SNK_OK;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    hipLaunchKernelGGL(rectify_kernel, dim3(ceil_div(cap, 256), batch), dim3(256), 0, m->stream, rc, kps_dev, n_dev, cap,
                       0, out_dev, (double2*)normalized_dev);
    SNK_LAUNCH_CHECK();
    return SNK_OK;
}

}  // extern "C"

int snk::rectify_pair_dev(snk_matcher* m, const snk_rectification* rect_left, const snk_rectification* rect_right, const snk_keypoint* kps_dev,
                          const int32_t* n_dev, int cap, snk_kp64* out_dev, double* normalized_left_dev)
{
    SNK_REQUIRE(m != nullptr && kps_dev && n_dev && out_dev, "NULL argument");
    RectDev rl, rr;
    int st = to_dev(rect_left, &rl);
    if (st == SNK_OK) st = to_dev(rect_right, &rr);
    if (st != SNK_OK) return st;
    if (cap == 0) return SNK_OK;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    hipLaunchKernelGGL(rectify_pair_kernel, dim3(ceil_div(cap, 256), 2), dim3(256), 0, m->stream, rl, rr, kps_dev, n_dev, cap, out_dev,
                       (double2*)normalized_left_dev);
    SNK_LAUNCH_CHECK();
    return SNK_OK;
}

extern "C" {

int snk_rectify(snk_matcher* m, const snk_rectification* rect, const snk_keypoint* kps, int n, snk_kp64* out,
                double (*normalized)[2])
{
    SNK_REQUIRE(m != nullptr, "matcher is NULL");
    SNK_REQUIRE(n >= 0, "negative count");
    RectDev rc;
    int st = to_dev(rect, &rc);
    if (st != SNK_OK) return st;
    if (n == 0) return SNK_OK;
    SNK_REQUIRE(kps && out, "NULL buffer");
    SNK_HIP_CHECK(hipSetDevice(m->device));
    // aux: keypoints in | undistorted out | normalized out (16-byte aligned pieces); through the pinned staging buffers, one copy each way
    const size_t kin = ((size_t)n * sizeof(snk_keypoint) + 15) & ~(size_t)15, kout = (size_t)n * sizeof(snk_kp64), nb = (size_t)n * 16;
    const size_t res_b = kout + (normalized ? nb : 0);
    if ((st = m->aux.reserve(kin + kout + nb)) != SNK_OK) return st;
    if ((st = m->h_in.reserve(kin)) != SNK_OK) return st;
    if ((st = m->h_res.reserve(res_b)) != SNK_OK) return st;
    char* ab = m->aux.as<char>();
    memcpy(m->h_in.p, kps, (size_t)n * sizeof(snk_keypoint));
    SNK_HIP_CHECK(hipMemcpyAsync(ab, m->h_in.p, (size_t)n * sizeof(snk_keypoint), hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(rectify_kernel, dim3(ceil_div(n, 256), 1), dim3(256), 0, m->stream, rc, (const snk_keypoint*)ab,
                       (const int*)nullptr, n, n, (snk_kp64*)(ab + kin), normalized ? reinterpret_cast<double2*>(ab + kin + kout) : nullptr);
    SNK_LAUNCH_CHECK();
    SNK_HIP_CHECK(hipMemcpyAsync(m->h_res.p, ab + kin, res_b, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(m->stream));
    memcpy(out, m->h_res.p, kout);
    if (normalized) memcpy(normalized, m->h_res.as<char>() + kout, nb);
    return SNK_OK;
}

static int rgbd_params(const snk_rgbd_model* mdl, RgbdDev* rc)
{
    SNK_REQUIRE(mdl != nullptr, "rgbd model is NULL");
    SNK_REQUIRE(mdl->K[0] != 0.0 && mdl->K[1] != 0.0, "K focal length is zero");
    memcpy(rc->K, mdl->K, sizeof(rc->K));
    memcpy(rc->D, mdl->D_depth, sizeof(rc->D));
    memcpy(rc->Kd, mdl->K_depth, sizeof(rc->Kd));
    rc->bf = mdl->bf;
    return SNK_OK;
}

int snk_rgbd_stereo_batch_dev(snk_matcher* m, const snk_rgbd_model* model, const snk_kp64* undistorted_dev, const int32_t* n_dev, int cap,
                              int batch, const float* depth_images_dev, int width, int height, int pitch_floats, size_t image_stride_floats,
                              float* right_points_dev, float* depth_dev, int32_t* n_matches_dev, int32_t* status_dev)
{
    SNK_REQUIRE(m != nullptr, "matcher is NULL");
    SNK_REQUIRE(batch >= 0 && cap >= 0 && width >= 1 && height >= 1 && pitch_floats >= width, "bad sizes");
    SNK_REQUIRE(undistorted_dev && n_dev && depth_images_dev && right_points_dev && depth_dev && n_matches_dev && status_dev, "NULL device buffer");
    RgbdDev rc;
    int st = rgbd_params(model, &rc);
    if (st != SNK_OK) return st;
    if (batch == 0) return SNK_OK;
    SNK_HIP_CHECK(hipSetDevice(m->device));
    hipLaunchKernelGGL(rgbd_init_kernel, dim3(ceil_div(batch, 256)), dim3(256), 0, m->stream, n_matches_dev, status_dev, batch);
    if (cap > 0)
        hipLaunchKernelGGL(rgbd_stereo_kernel, dim3(ceil_div(cap, 256), batch), dim3(256), 0, m->stream, rc, undistorted_dev, n_dev, cap, 0,
                           depth_images_dev, width, height, pitch_floats, (long long)image_stride_floats, right_points_dev, depth_dev,
                           n_matches_dev, status_dev);
    SNK_LAUNCH_CHECK();
    return SNK_OK;
}

int snk_rgbd_stereo(snk_matcher* m, const snk_rgbd_model* model, const snk_kp64* undistorted, int n, const float* depth_image, int width,
                    int height, int pitch_floats, float* right_points, float* depth, int* n_matches)
{
    SNK_REQUIRE(m != nullptr && n_matches != nullptr, "NULL argument");
    *n_matches = 0;
    SNK_REQUIRE(n >= 0 && width >= 1 && height >= 1 && pitch_floats >= width, "bad sizes");
    RgbdDev rc;
    int st = rgbd_params(model, &rc);
    if (st != SNK_OK) return st;
    if (n == 0) return SNK_OK;
    SNK_REQUIRE(undistorted && depth_image && right_points && depth, "NULL buffer");
    SNK_HIP_CHECK(hipSetDevice(m->device));
    // aux: keypoints | right_points | depth | n_matches, status;   aux2: the depth image
    const size_t kb = ((size_t)n * sizeof(snk_kp64) + 15) & ~(size_t)15, fb = ((size_t)n * 4 + 15) & ~(size_t)15;
    const size_t ib = (size_t)width * height * 4;
    if ((st = m->aux.reserve(kb + 2 * fb + 16)) != SNK_OK) return st;
    if ((st = m->aux2.reserve(ib)) != SNK_OK) return st;
    char* ab = m->aux.as<char>();
    SNK_HIP_CHECK(hipMemcpyAsync(ab, undistorted, (size_t)n * sizeof(snk_kp64), hipMemcpyHostToDevice, m->stream));
    SNK_HIP_CHECK(hipMemcpy2DAsync(m->aux2.p, (size_t)width * 4, depth_image, (size_t)pitch_floats * 4, (size_t)width * 4, (size_t)height,
                                   hipMemcpyHostToDevice, m->stream));
    int* d_cnt = reinterpret_cast<int*>(ab + kb + 2 * fb);
    hipLaunchKernelGGL(rgbd_init_kernel, dim3(1), dim3(256), 0, m->stream, d_cnt, d_cnt + 1, 1);
    hipLaunchKernelGGL(rgbd_stereo_kernel, dim3(ceil_div(n, 256), 1), dim3(256), 0, m->stream, rc, (const snk_kp64*)ab, (const int*)nullptr, n, n,
                       m->aux2.as<float>(), width, height, width, (long long)0, reinterpret_cast<float*>(ab + kb),
                       reinterpret_cast<float*>(ab + kb + fb), d_cnt, d_cnt + 1);
    SNK_LAUNCH_CHECK();
    int res[2] = {0, 0};
    SNK_HIP_CHECK(hipMemcpyAsync(res, d_cnt, 8, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(m->stream));
    if (res[1] != 0x7FFFFFFF)
    {
        // the reference aborts here (SAIGA_ASSERT, Preprocess.cpp:100,103-104); outputs untouched
        set_error("ComputeStereoFromRGBD: keypoint %d reprojects outside the depth image or its depth is not in [0, 20)", res[1] - 1);
        *n_matches = -res[1];
        return SNK_ERR_INVALID_ARG;
    }
    SNK_HIP_CHECK(hipMemcpyAsync(right_points, ab + kb, (size_t)n * 4, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipMemcpyAsync(depth, ab + kb + fb, (size_t)n * 4, hipMemcpyDeviceToHost, m->stream));
    SNK_HIP_CHECK(hipStreamSynchronize(m->stream));
    *n_matches = res[0];
    return SNK_OK;
}
}
