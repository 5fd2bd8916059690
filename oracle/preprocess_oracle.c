/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement of Snake::Preprocess::undistortKeypoints (reference
 * Snake/Preprocess/Preprocess.cpp:55-77) and of Rectification::Forward as used by
 * StereoMatching (Preprocess.cpp:140-150): unproject with K_src, undistort (Gauss-Newton),
 * rotate by R, perspective divide, project with K_dst.
 *
 * PARITY UNPINNED: Rectification / undistortPointGN / Distortion live in the absent saiga
 * submodule.  [DEFINED] here: the OpenCV rational radial-tangential model with coefficients
 * (k1..k6, p1, p2) and 5 Gauss-Newton iterations started at the distorted point, all in fp64
 * with a fixed operation order (no FMA contraction).
 */
#include <math.h>
#include <stdint.h>

#include "snk_oracle.h"

static void distort(const double* D, double x, double y, double* xd, double* yd, double J[4])
{
    const double k1 = D[0], k2 = D[1], k3 = D[2], k4 = D[3], k5 = D[4], k6 = D[5], p1 = D[6], p2 = D[7];
    const double x2 = x * x, y2 = y * y, xy = x * y;
    const double r2 = x2 + y2, r4 = r2 * r2, r6 = r4 * r2;
    const double num = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
    const double den = 1.0 + k4 * r2 + k5 * r4 + k6 * r6;
    const double rad = num / den;
    *xd = x * rad + 2.0 * p1 * xy + p2 * (r2 + 2.0 * x2);
    *yd = y * rad + p1 * (r2 + 2.0 * y2) + 2.0 * p2 * xy;
    /* d rad / d r2 */
    const double dnum = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4;
    const double dden = k4 + 2.0 * k5 * r2 + 3.0 * k6 * r4;
    const double drad = (dnum * den - num * dden) / (den * den);
    J[0] = rad + x * drad * 2.0 * x + 2.0 * p1 * y + p2 * (2.0 * x + 4.0 * x); /* dxd/dx */
    J[1] = x * drad * 2.0 * y + 2.0 * p1 * x + p2 * 2.0 * y;                   /* dxd/dy */
    J[2] = y * drad * 2.0 * x + p1 * 2.0 * x + 2.0 * p2 * y;                   /* dyd/dx */
    J[3] = rad + y * drad * 2.0 * y + p1 * (2.0 * y + 4.0 * y) + 2.0 * p2 * x; /* dyd/dy */
}

void orc_undistort_gn(const double* D, double px, double py, double* ox, double* oy)
{
    double x = px, y = py;
    for (int it = 0; it < 5; ++it)
    {
        double xd, yd, J[4];
        distort(D, x, y, &xd, &yd, J);
        const double rx = xd - px, ry = yd - py;
        const double det = J[0] * J[3] - J[1] * J[2];
        const double dx  = (J[3] * rx - J[1] * ry) / det;
        const double dy  = (J[0] * ry - J[2] * rx) / det;
        x = x - dx;
        y = y - dy;
    }
    *ox = x;
    *oy = y;
}

/* Preprocess.cpp:62-75 for every keypoint: out = rectified pixel, normalized = the point stored
 * in frame.normalized_points (may be NULL).  octave / angle are copied (Preprocess.cpp:60,129-130). */
void orc_rectify(const orc_rectification* R, const orc_keypoint* kps, int n, orc_kp64* out, double (*normalized)[2])
{
    for (int i = 0; i < n; ++i)
    {
        double x = (double)kps[i].x, y = (double)kps[i].y;
        x = (x - R->K_src[2]) / R->K_src[0]; /* K_src.unproject2 */
        y = (y - R->K_src[3]) / R->K_src[1];
        double ux, uy;
        orc_undistort_gn(R->D_src, x, y, &ux, &uy); /* undistortPointGN(p, p, D_src) */
        const double rx = R->R[0] * ux + R->R[1] * uy + R->R[2];
        const double ry = R->R[3] * ux + R->R[4] * uy + R->R[5];
        const double rz = R->R[6] * ux + R->R[7] * uy + R->R[8];
        const double nx = rx / rz, ny = ry / rz;
        if (normalized)
        {
            normalized[i][0] = nx;
            normalized[i][1] = ny;
        }
        out[i].x      = R->K_dst[0] * nx + R->K_dst[2]; /* K_dst.normalizedToImage */
        out[i].y      = R->K_dst[1] * ny + R->K_dst[3];
        out[i].angle  = kps[i].angle;
        out[i].octave = kps[i].octave;
    }
}

/* Snake::Preprocess::ComputeStereoFromRGBD (reference Snake/Preprocess/Preprocess.cpp:79-120), the RGB-D branch of Preprocess::Process:
 * per undistorted keypoint, K.unproject2 -> distortNormalizedPoint(depthModel.dis) -> depthModel.K.normalizedToImage (:93-95), the
 * nearest depth pixel by `int x = reprojected.x() + 0.5` (:97-98: truncation towards zero of the double), and
 *   depth > 0 : depth[i] = depth, right_points[i] = kpun.x - bf / depth (:106-109, float stores)      else: both -1 (:113-114).
 * Returns the number of keypoints with depth, or the NEGATED (index + 1) of the first keypoint on which the reference would
 * abort: outside the depth image (:100), depth < 0 or >= 20 (:103-104).  [DEFINED]: distortNormalizedPoint = the forward rational
 * radial-tangential model of `distort` above (what undistortPointGN inverts). */
int orc_rgbd_stereo(const orc_kp64* und, int n, const double* K, const double* D_depth, const double* K_depth, double bf,
                    const float* depth_image, int w, int h, int pitch_floats, float* right_points, float* depth)
{
    int matches = 0;
    for (int i = 0; i < n; ++i)
    {
        const double nx = (und[i].x - K[2]) / K[0], ny = (und[i].y - K[3]) / K[1];
        double dx, dy, J[4];
        distort(D_depth, nx, ny, &dx, &dy, J);
        const double rx = K_depth[0] * dx + K_depth[2], ry = K_depth[1] * dy + K_depth[3];
        const int ok_range = rx + 0.5 > -1.0 && ry + 0.5 > -1.0 && rx < 2.0e9 && ry < 2.0e9; /* (int) of anything else is undefined */
        const int x = ok_range ? (int)(rx + 0.5) : -1, y = ok_range ? (int)(ry + 0.5) : -1;
        if (!ok_range || x < 0 || y < 0 || x >= w || y >= h) return -(i + 1);
        const float d = depth_image[(long long)y * pitch_floats + x];
        if (!(d >= 0.0f) || !(d < 20.0f)) return -(i + 1);
        if (d > 0.0f)
        {
            depth[i]               = d;
            const double disparity = bf / (double)d;
            right_points[i]        = (float)(und[i].x - disparity);
            ++matches;
        }
        else
        {
            depth[i]        = -1.0f;
            right_points[i] = -1.0f;
        }
    }
    return matches;
}
