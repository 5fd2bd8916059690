"""CPU: the pose-refinement oracle (snk-pose v1) against independent numpy checks."""
import numpy as np
import pytest

import pose_helpers as PH


def cam_of(orc):
    return orc.Camera(*PH.CAM)


def test_chi2_matches_numpy_projection(orc):
    pr = PH.make_problem(1, 200)
    fx, fy, cx, cy, bf = PH.CAM
    R, t = PH.quat_to_R(pr["pose0"][:4]), pr["pose0"][4:]
    pc = pr["wps"] @ R.T + t
    u = fx * pc[:, 0] / pc[:, 2] + cx
    v = fy * pc[:, 1] / pc[:, 2] + cy
    o = pr["obs"]
    r0, r1 = o["weight"] * (u - o["x"]), o["weight"] * (v - o["y"])
    r2 = np.where(o["depth"] > 0, o["weight"] * ((u - bf / pc[:, 2]) - (o["x"] - bf / np.where(o["depth"] > 0, o["depth"], 1))), 0)
    want = r0 ** 2 + r1 ** 2 + r2 ** 2
    got = orc.pose_chi2(pr["pose0"], cam_of(orc), pr["wps"], o)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("seed", [2, 3, 4])
def test_refine_recovers_pose_and_flags_outliers(orc, seed):
    pr = PH.make_problem(seed, 300, outlier_frac=0.25)
    pose, outl, inl = orc.pose_refine(pr["pose0"], cam_of(orc), pr["wps"], pr["obs"])
    rot0, tr0 = PH.pose_error(pr["pose0"], pr["pose_gt"])
    rot, tr = PH.pose_error(pose, pr["pose_gt"])
    assert rot < 2e-3 and tr < 0.02 and rot < rot0 and tr < tr0
    assert outl[pr["is_outlier"]].all()                 # every gross outlier is flagged
    assert (~outl[~pr["is_outlier"]].astype(bool)).mean() > 0.9  # nearly all true matches are kept
    assert inl == int((outl == 0).sum())
    assert abs(np.linalg.norm(pose[:4]) - 1) < 1e-12


def test_points_behind_the_camera_are_outliers(orc):
    pr = PH.make_problem(5, 120, outlier_frac=0.0, behind=7)
    pose, outl, inl = orc.pose_refine(pr["pose0"], cam_of(orc), pr["wps"], pr["obs"])
    assert outl[:7].all() and inl == int((outl == 0).sum())
    assert np.isinf(orc.pose_chi2(pose, cam_of(orc), pr["wps"], pr["obs"])[:7]).all()


def test_degenerate_inputs_leave_a_finite_pose(orc):
    pr = PH.make_problem(6, 2, outlier_frac=0.0)
    pose, outl, inl = orc.pose_refine(pr["pose0"], cam_of(orc), pr["wps"], pr["obs"])
    assert np.isfinite(pose).all()
    pose, outl, inl = orc.pose_refine(pr["pose0"], cam_of(orc), pr["wps"][:0], pr["obs"][:0])
    assert inl == 0 and np.allclose(pose, pr["pose0"], rtol=0, atol=1e-15)


def test_se3_log_rel(orc):
    from scipy.spatial.transform import Rotation

    rng = np.random.default_rng(7)
    for _ in range(20):
        a, b = PH.random_pose(rng, 1.0), PH.random_pose(rng, 1.0)
        e = orc.se3_log_rel(a, b)
        Ra, Rb = PH.quat_to_R(a[:4]), PH.quat_to_R(b[:4])
        Re = Ra @ Rb.T
        w = Rotation.from_matrix(Re).as_rotvec()
        assert np.allclose(e[3:], w, atol=1e-10)
        # exp(e) must reproduce the relative translation: t_e = V(w) rho
        th = np.linalg.norm(w)
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * K @ K
        assert np.allclose(V @ e[:3], a[4:] - Re @ b[4:], atol=1e-10)
    assert np.allclose(orc.se3_log_rel(a, a), 0, atol=1e-14)


def test_prior_pulls_towards_the_prediction(orc):
    pr = PH.make_problem(8, 60, outlier_frac=0.0, noise=1.5)
    free, _, _ = orc.pose_refine(pr["pose0"], cam_of(orc), pr["wps"], pr["obs"])
    tied, _, _ = orc.pose_refine(pr["pose0"], cam_of(orc), pr["wps"], pr["obs"], prediction=pr["pose0"], w_rot=1e4, w_trans=1e4)
    assert sum(PH.pose_error(tied, pr["pose0"])) < 1e-3 < sum(PH.pose_error(free, pr["pose0"]))
    weak, _, _ = orc.pose_refine(pr["pose0"], cam_of(orc), pr["wps"], pr["obs"], prediction=pr["pose0"], w_rot=1e-6, w_trans=1e-6)
    assert sum(PH.pose_error(weak, free)) < 1e-6
