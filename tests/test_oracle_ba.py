"""CPU: pin the BA oracle ("snk-ba v1").  PARITY UNPINNED vs saiga's BARecRel (absent); pinned
here against an independent numpy restatement of the cost, exact recovery of a noise-free scene,
and first-order optimality of the converged solution."""
import numpy as np
import pytest

import ba_numpy


@pytest.fixture(scope="module")
def small():
    from snake_slam_amd import synth

    return synth.ba_scene(n_kf=6, n_pt=120, obs_per_pt=4, seed=11)


def test_se3_update_matches_matrix_exponential(orc):
    import ctypes as C

    rng = np.random.default_rng(0)
    for scale in (1e-10, 1e-3, 0.3, 2.0):
        pose = np.concatenate([rng.normal(size=4), rng.normal(size=3)])
        pose[:4] /= np.linalg.norm(pose[:4])
        d = rng.normal(size=6) * scale
        out = np.zeros(7)
        orc.lib().orc_se3_update(C.c_void_p(pose.ctypes.data), C.c_void_p(d.ctypes.data), C.c_void_p(out.ctypes.data))
        T = ba_numpy.se3_exp_update(pose, d)
        assert np.abs(ba_numpy.quat_R(out[:4]) - T[:3, :3]).max() < 1e-12
        assert np.abs(out[4:] - T[:3, 3]).max() < 1e-12


def test_costs_and_chi2_match_numpy(orc, small):
    sc, _ = small
    chi = orc.ba_chi2(sc)
    res = ba_numpy.residuals(sc)
    for o, r in enumerate(res):
        want = 0.0 if r is None else float(r @ r)
        assert abs(chi[o] - want) <= 1e-9 * max(1.0, want)
    _, _, c0, c1, _ = orc.ba_solve(sc, orc.ba_options())
    assert abs(c0 - ba_numpy.robust_cost(sc)) <= 1e-9 * c0
    assert c1 < 0.2 * c0


def test_noise_free_scene_is_recovered_exactly(orc):
    from snake_slam_amd import synth

    sc, gt = synth.ba_scene(n_kf=8, n_pt=300, obs_per_pt=5, pixel_noise=0.0, seed=5)
    pose, pt, c0, c1, _ = orc.ba_solve(sc, orc.ba_options(max_iterations=12))
    assert c0 > 100 and c1 < 1e-9
    assert np.sqrt(((pt - gt["pt"]) ** 2).sum(1).mean()) < 1e-6
    assert np.abs(pose[:, 4:] - gt["pose"][:, 4:]).max() < 1e-6
    assert np.array_equal(pose[0], sc["pose"][0])  # the constant camera never moves


def test_converged_solution_is_a_stationary_point(orc, small):
    sc, _ = small
    pose, pt, c0, c1, _ = orc.ba_solve(sc, orc.ba_options(max_iterations=25, max_pcg_iterations=200))
    assert abs(ba_numpy.robust_cost(sc, pose, pt) - c1) <= 1e-9 * c1
    # finite-difference gradient wrt a few point coordinates and one camera translation
    h = 1e-6
    g0 = []
    for (p, a) in [(3, 0), (40, 1), (77, 2)]:
        for (P, T, store) in ((sc["pt"], sc["pose"], g0),):
            pass
    def grad_pt(pose_, pt_, p, a):
        q = pt_.copy(); q[p, a] += h
        m = pt_.copy(); m[p, a] -= h
        return (ba_numpy.robust_cost(sc, pose_, q) - ba_numpy.robust_cost(sc, pose_, m)) / (2 * h)
    def grad_t(pose_, pt_, i, a):
        q = pose_.copy(); q[i, 4 + a] += h
        m = pose_.copy(); m[i, 4 + a] -= h
        return (ba_numpy.robust_cost(sc, q, pt_) - ba_numpy.robust_cost(sc, m, pt_)) / (2 * h)
    for (p, a) in [(3, 0), (40, 1), (77, 2)]:
        assert abs(grad_pt(pose, pt, p, a)) < 1e-3 * max(1.0, abs(grad_pt(sc["pose"], sc["pt"], p, a)))
    for (i, a) in [(2, 0), (5, 2)]:
        assert abs(grad_t(pose, pt, i, a)) < 1e-3 * max(1.0, abs(grad_t(sc["pose"], sc["pt"], i, a)))


def test_constant_points_and_outlier_mask(orc, small):
    sc, _ = small
    sc2 = dict(sc)
    sc2["pt_const"] = np.zeros_like(sc["pt_const"])
    sc2["pt_const"][:30] = 1
    pose, pt, c0, c1, _ = orc.ba_solve(sc2, orc.ba_options())
    assert np.array_equal(pt[:30], sc["pt"][:30]) and not np.array_equal(pt[30:], sc["pt"][30:])
    out = np.zeros(len(sc["obs_img"]), np.uint8)
    out[::7] = 1
    chi = orc.ba_chi2(sc, out)
    assert (chi[::7] == 0).all()
    _, _, c0m, _, _ = orc.ba_solve(sc, orc.ba_options(), outlier=out)
    assert abs(c0m - ba_numpy.robust_cost(sc, outlier=out)) <= 1e-9 * c0m


def test_benchmark_scene_shape(orc):
    from snake_slam_amd import synth

    sc, gt = synth.ba_scene()
    assert len(sc["obs_img"]) == 16000 and sc["pose"].shape == (20, 7) and sc["pt"].shape == (2000, 3)
    assert 0.45 < (sc["obs_depth"] > 0).mean() < 0.55
    _, pt, c0, c1, _ = orc.ba_solve(sc, orc.ba_options())
    assert c1 < 0.05 * c0
    assert np.sqrt(((pt - gt["pt"]) ** 2).sum(1).mean()) < np.sqrt(((sc["pt"] - gt["pt"]) ** 2).sum(1).mean())
