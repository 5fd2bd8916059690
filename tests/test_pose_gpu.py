"""GPU: HIP pose refinement (snk-pose v1) vs the CPU oracle through the C ABI.  Floating point:
poses within 1e-9 (observed ~1e-13), outlier flags identical except for matches whose chi-square
sits within 1e-6 of the threshold, inlier counts consistent with the flags."""
import numpy as np
import pytest

import pose_helpers as PH

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-9


def compare(orc, got, pr, opt, **prior):
    cam = orc.Camera(*PH.CAM)
    pose_w, out_w, inl_w = orc.pose_refine(pr["pose0"], cam, pr["wps"], pr["obs"], opt, **prior)
    pose_g, out_g, inl_g = got
    assert np.allclose(pose_g, pose_w, rtol=0, atol=POSE_TOL), np.abs(pose_g - pose_w).max()
    diff = np.nonzero(out_g != out_w)[0]
    if len(diff):
        chi2 = orc.pose_chi2(pose_w, cam, pr["wps"], pr["obs"])
        th2 = np.where(pr["obs"]["depth"] > 0, opt.th_stereo ** 2, opt.th_mono ** 2)
        assert (np.abs(chi2[diff] - th2[diff]) < 1e-6).all(), "outlier flags differ away from the threshold"
    assert inl_g == int((out_g == 0).sum())
    assert abs(inl_g - inl_w) <= len(diff)


@pytest.mark.parametrize("seed,n", [(11, 300), (12, 64), (13, 65), (14, 1000), (15, 7)])
def test_refine_matches_oracle(orc, seed, n):
    from snake_slam_amd.tracking import PoseRefinement

    pr = PH.make_problem(seed, n, outlier_frac=0.2)
    ref = PoseRefinement()
    try:
        got = ref.refinePose(PH.CAM, pr["pose0"], pr["wps"], pr["obs"])
        compare(orc, got, pr, orc.pose_options())
        rot, tr = PH.pose_error(got[0], pr["pose_gt"])
        if n >= 64:
            assert rot < 5e-3 and tr < 0.05
    finally:
        ref.close()


def test_more_than_32_matches_per_thread(orc):
    """The kernel keeps the outlier flags of a thread's first 32 matches in a register and reads the rest from the `outlier` array: 9000
    matches through the four-wavefront kernel (36 per thread) and, in a batch whose average is small (one wavefront per problem),
    a problem of 2300 matches."""
    from snake_slam_amd.tracking import PoseRefinement

    ref = PoseRefinement()
    try:
        big = PH.make_problem(31, 9000, outlier_frac=0.25)
        compare(orc, ref.refinePose(PH.CAM, big["pose0"], big["wps"], big["obs"]), big, orc.pose_options())
        prs = [PH.make_problem(40 + i, n, outlier_frac=0.2) for i, n in enumerate([2300] + [20] * 15)]  # 2600 matches / 16 problems < 192
        res = ref.refine_batch(PH.CAM, [dict(pose=p["pose0"], wps=p["wps"], obs=p["obs"]) for p in prs])
        for got, pr in zip(res, prs):
            compare(orc, got, pr, orc.pose_options())
    finally:
        ref.close()


def test_batch_of_frames_one_launch(orc):
    from snake_slam_amd.tracking import PoseRefinement

    prs = [PH.make_problem(20 + i, n, outlier_frac=0.15, behind=b) for i, (n, b) in enumerate([(250, 0), (0, 0), (3, 0), (500, 5), (129, 0)])]
    ref = PoseRefinement(errorFactor=1.5, outer=3, inner=5, robust_rounds=2, lam=1e-3)
    try:
        res = ref.refine_batch(PH.CAM, [dict(pose=p["pose0"], wps=p["wps"], obs=p["obs"]) for p in prs])
        opt = orc.pose_options(2.1 * 1.5, 2.3 * 1.5, 3, 5, 2, 1e-3)
        for got, pr in zip(res, prs):
            compare(orc, got, pr, opt)
        assert res[1][2] == 0 and len(res[1][1]) == 0
        assert res[3][1][:5].all()  # points behind the camera
    finally:
        ref.close()


def test_smooth_variant_with_prediction_prior(orc):
    from snake_slam_amd.tracking import PoseRefinement

    pr = PH.make_problem(31, 80, outlier_frac=0.1, noise=1.0)
    rng = np.random.default_rng(31)
    pred = PH.perturb(rng, pr["pose_gt"], 0.02, 0.05)
    ref = PoseRefinement()
    try:
        for wr, wt in [(50.0, 20.0), (1e4, 1e4), (1e-3, 0.0)]:
            got = ref.refinePose(PH.CAM, pr["pose0"], pr["wps"], pr["obs"], prediction=pred, prediction_weight_rotation=wr,
                                 prediction_weight_translation=wt)
            compare(orc, got, pr, orc.pose_options(), prediction=pred, w_rot=wr, w_trans=wt)
        tied = ref.refinePose(PH.CAM, pr["pose0"], pr["wps"], pr["obs"], prediction=pred, prediction_weight_rotation=1e4,
                              prediction_weight_translation=1e4)
        assert sum(PH.pose_error(tied[0], pred)) < 1e-3
    finally:
        ref.close()


def test_refine_with_matches_needs_three(orc):
    from snake_slam_amd.tracking import PoseRefinement

    pr = PH.make_problem(41, 2, outlier_frac=0.0)
    ref = PoseRefinement()
    try:
        pose, outl, inl = ref.RefinePoseWithMatches(PH.CAM, pr["pose0"], pr["wps"], pr["obs"])
        assert inl == 0 and np.array_equal(pose, pr["pose0"])
    finally:
        ref.close()


def test_bad_arguments_are_rejected(orc):
    from snake_slam_amd import SnakeHipError
    from snake_slam_amd.tracking import PoseRefinement

    pr = PH.make_problem(42, 10)
    ref = PoseRefinement()
    try:
        ref.options.th_mono = -1.0
        with pytest.raises(SnakeHipError):
            ref.refinePose(PH.CAM, pr["pose0"], pr["wps"], pr["obs"])
    finally:
        ref.close()
