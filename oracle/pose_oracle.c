/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement ("oracle") of the robust pose-only optimisation the reference runs right after
 * every projection matcher: PoseRefinement::refinePose (Snake/Tracking/PoseRefinement.h:27-87, call
 * sites TrackingCoarse.cpp:270, TrackingFine.cpp:158) and PoseRefinement::RefinePoseWithMatches
 * (Snake/Tracking/PoseRefinement.cpp:25-79, call sites TrackingCoarse.cpp:452,477,
 * LoopDetector.cpp:284).
 *
 * PARITY UNPINNED.  The optimiser itself (Saiga::RobustPoseOptimization::optimizePoseRobust and
 * RobustSmoothPoseOptimization) lives in the absent, unpinned submodule darglein/saiga.  What the
 * reference fixes is restated as is: the observation of a match = (undistorted keypoint, weight =
 * sqrt(InverseSquaredScale(octave)), depth; PoseRefinement.h:47-55), world point = local-map /
 * map-point position, chi thresholds reprojectionErrorThreshold{Mono,Stereo} * errorFactor
 * (PoseRefinement.cpp:13-15, SnakeGlobal.h:145-146), outlier flags written back per match and the
 * inlier count returned (PoseRefinement.h:78-85), optional pose prior with rotation / translation
 * weights (PoseRefinement.h:68-73).  Everything else is [DEFINED] here ("snk-pose v1", ORB-SLAM2
 * PoseOptimization lineage):
 *   residual  as snk-ba v1: r = weight * (projection - observation), stereo adds
 *             u_r = u - bf / z against (u_obs - bf / depth); z <= 0 => no contribution, outlier;
 *   rounds    `outer` rounds of `inner` damped Gauss-Newton iterations over the current inliers;
 *             Huber (delta = threshold) in rounds 0 .. robust_rounds-1, plain least squares after;
 *             after each round every match is re-classified: outlier <=> |r|^2 > threshold^2
 *             (outliers can come back);
 *   step      (H + lambda * clamp(diag(H), 1e-6, 1e32)) delta = -b, Cholesky, T <- exp(delta) * T;
 *             a failed factorisation leaves the pose unchanged for that iteration;
 *   prior     e = log(T * T_pred^-1) = (rho, omega); H += diag(wt^2 I3, wr^2 I3), b += that * e
 *             (first-order Jacobian I).
 * Double precision throughout.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "snk_oracle.h"

static void quat_to_R(const double* q, double* R)
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

/* residual r (dim 2 / 3) and d r / d(delta) rows; returns dim or 0 (behind the camera) */
static int linearize(const double* R, const double* t, const double* p, const orc_camera* cam, const orc_pose_obs* o,
                     double* r, double* J)
{
    const double X = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + t[0];
    const double Y = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + t[1];
    const double Z = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + t[2];
    if (Z <= 0.0) return 0;
    const double iz = 1.0 / Z, iz2 = iz * iz, w = o->weight;
    const int dim = o->depth > 0.0 ? 3 : 2;
    double P[9];
    r[0] = w * (cam->fx * X * iz + cam->cx - o->x);
    r[1] = w * (cam->fy * Y * iz + cam->cy - o->y);
    P[0] = cam->fx * iz; P[1] = 0.0;          P[2] = -cam->fx * X * iz2;
    P[3] = 0.0;          P[4] = cam->fy * iz; P[5] = -cam->fy * Y * iz2;
    if (dim == 3)
    {
        r[2] = w * ((cam->fx * X * iz + cam->cx - cam->bf * iz) - (o->x - cam->bf / o->depth));
        P[6] = cam->fx * iz; P[7] = 0.0; P[8] = -cam->fx * X * iz2 + cam->bf * iz2;
    }
    for (int k = 0; k < dim; ++k)
    {
        const double a = w * P[3 * k], b = w * P[3 * k + 1], c = w * P[3 * k + 2];
        J[6 * k + 0] = a;
        J[6 * k + 1] = b;
        J[6 * k + 2] = c;
        J[6 * k + 3] = -b * Z + c * Y;
        J[6 * k + 4] = a * Z - c * X;
        J[6 * k + 5] = -a * Y + b * X;
    }
    return dim;
}

/* e = log(T * T_pred^-1) as (rho, omega) */
void orc_se3_log_rel(const double* pose, const double* pred, double* e)
{
    /* q_e = q * conj(q_pred) */
    const double ax = pose[0], ay = pose[1], az = pose[2], aw = pose[3];
    const double bx = -pred[0], by = -pred[1], bz = -pred[2], bw = pred[3];
    double q[4];
    q[0] = aw * bx + ax * bw + ay * bz - az * by;
    q[1] = aw * by - ax * bz + ay * bw + az * bx;
    q[2] = aw * bz + ax * by - ay * bx + az * bw;
    q[3] = aw * bw - ax * bx - ay * by - az * bz;
    if (q[3] < 0.0)
        for (int i = 0; i < 4; ++i) q[i] = -q[i];
    double Re[9];
    quat_to_R(q, Re);
    /* t_e = t - R_e t_pred */
    const double tx = pose[4] - (Re[0] * pred[4] + Re[1] * pred[5] + Re[2] * pred[6]);
    const double ty = pose[5] - (Re[3] * pred[4] + Re[4] * pred[5] + Re[5] * pred[6]);
    const double tz = pose[6] - (Re[6] * pred[4] + Re[7] * pred[5] + Re[8] * pred[6]);
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2], n = sqrt(n2);
    double wx, wy, wz, th, cc; /* omega, |omega|, coefficient of [w]x^2 in V^-1 */
    if (n < 1e-10)
    {
        const double k = 2.0 / q[3];
        wx = k * q[0]; wy = k * q[1]; wz = k * q[2];
        th = 0.0;
        cc = 1.0 / 12.0;
    }
    else
    {
        th = 2.0 * atan2(n, q[3]);
        const double k = th / n;
        wx = k * q[0]; wy = k * q[1]; wz = k * q[2];
        if (th < 1e-4)
            cc = 1.0 / 12.0 + th * th / 720.0;
        else
            cc = (1.0 - (th * sin(th)) / (2.0 * (1.0 - cos(th)))) / (th * th);
    }
    /* rho = V^-1 t = t - 0.5 w x t + cc w x (w x t) */
    const double c1x = wy * tz - wz * ty, c1y = wz * tx - wx * tz, c1z = wx * ty - wy * tx;
    const double c2x = wy * c1z - wz * c1y, c2y = wz * c1x - wx * c1z, c2z = wx * c1y - wy * c1x;
    e[0] = tx - 0.5 * c1x + cc * c2x;
    e[1] = ty - 0.5 * c1y + cc * c2y;
    e[2] = tz - 0.5 * c1z + cc * c2z;
    e[3] = wx; e[4] = wy; e[5] = wz;
}

static int chol_solve6(const double* A, const double* b, double* x)
{
    double L[36];
    memset(L, 0, sizeof(L));
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j)
        {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j)
            {
                if (!(s > 0.0)) return -1;
                L[i * 6 + i] = sqrt(s);
            }
            else
                L[i * 6 + j] = s / L[j * 6 + j];
        }
    double y[6];
    for (int i = 0; i < 6; ++i)
    {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * y[k];
        y[i] = s / L[i * 6 + i];
    }
    for (int i = 5; i >= 0; --i)
    {
        double s = y[i];
        for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k];
        x[i] = s / L[i * 6 + i];
    }
    return 0;
}

static double clampd(double v)
{
    return v < 1e-6 ? 1e-6 : (v > 1e32 ? 1e32 : v);
}

/* chi2[i] = |r_i|^2 at `pose` (HUGE_VAL behind the camera) */
void orc_pose_chi2(const double* pose, const orc_camera* cam, const double (*wps)[3], const orc_pose_obs* obs, int n,
                   double* chi2)
{
    double R[9];
    quat_to_R(pose, R);
    for (int i = 0; i < n; ++i)
    {
        double r[3], J[18];
        const int dim = linearize(R, pose + 4, wps[i], cam, &obs[i], r, J);
        double s = 0.0;
        for (int k = 0; k < dim; ++k) s += r[k] * r[k];
        chi2[i] = dim ? s : HUGE_VAL;
    }
}

/* pose: in/out; prediction may be NULL (no prior); outlier: out [n]; returns the inlier count. */
int orc_pose_refine(double* pose, const orc_camera* cam, const orc_pose_options* opt, const double (*wps)[3],
                    const orc_pose_obs* obs, int n, const double* prediction, double w_rot, double w_trans,
                    uint8_t* outlier)
{
    for (int i = 0; i < n; ++i) outlier[i] = 0;
    int inliers = 0;
    for (int round = 0; round < opt->outer_iterations; ++round)
    {
        const int robust = round < opt->robust_rounds;
        for (int it = 0; it < opt->inner_iterations; ++it)
        {
            double H[36], b[6], R[9];
            memset(H, 0, sizeof(H));
            memset(b, 0, sizeof(b));
            quat_to_R(pose, R);
            for (int i = 0; i < n; ++i)
            {
                if (outlier[i]) continue;
                double r[3], J[18];
                const int dim = linearize(R, pose + 4, wps[i], cam, &obs[i], r, J);
                if (!dim) continue;
                double s = 0.0;
                for (int k = 0; k < dim; ++k) s += r[k] * r[k];
                double wgt = 1.0; /* IRLS weight rho'(s) */
                if (robust)
                {
                    const double d = dim == 3 ? opt->th_stereo : opt->th_mono;
                    if (s > d * d) wgt = d / sqrt(s);
                }
                for (int k = 0; k < dim; ++k)
                {
                    for (int a = 0; a < 6; ++a)
                    {
                        const double ja = wgt * J[6 * k + a];
                        b[a] += ja * r[k];
                        for (int c = a; c < 6; ++c) H[a * 6 + c] += ja * J[6 * k + c];
                    }
                }
            }
            if (prediction)
            {
                double e[6];
                orc_se3_log_rel(pose, prediction, e);
                for (int a = 0; a < 6; ++a)
                {
                    const double w2 = a < 3 ? w_trans * w_trans : w_rot * w_rot;
                    H[a * 6 + a] += w2;
                    b[a] += w2 * e[a];
                }
            }
            for (int a = 0; a < 6; ++a)
            {
                H[a * 6 + a] += opt->lambda * clampd(H[a * 6 + a]);
                for (int c = 0; c < a; ++c) H[a * 6 + c] = H[c * 6 + a];
            }
            double nb[6], d[6];
            for (int a = 0; a < 6; ++a) nb[a] = -b[a];
            if (chol_solve6(H, nb, d) == 0)
            {
                double np[7];
                orc_se3_update(pose, d, np);
                memcpy(pose, np, sizeof(np));
            }
        }
        /* re-classify every match */
        double R[9];
        quat_to_R(pose, R);
        inliers = 0;
        for (int i = 0; i < n; ++i)
        {
            double r[3], J[18];
            const int dim = linearize(R, pose + 4, wps[i], cam, &obs[i], r, J);
            double s = 0.0;
            for (int k = 0; k < dim; ++k) s += r[k] * r[k];
            const double d = dim == 3 ? opt->th_stereo : opt->th_mono;
            outlier[i]     = (!dim || s > d * d) ? 1 : 0;
            inliers += outlier[i] ? 0 : 1;
        }
    }
    return inliers;
}
