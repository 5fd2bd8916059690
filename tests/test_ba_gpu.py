"""GPU: HIP local BA vs the CPU oracle through the C ABI.  Tolerance (north_star): pose / point
RMSE <= 1e-5 against the CPU restatement (fp64 on both sides; only summation orders differ)."""
import numpy as np
import os

import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-5


def rmse(a, b):
    return float(np.sqrt(((np.asarray(a) - np.asarray(b)) ** 2).sum(axis=-1).mean())) if len(a) else 0.0


def compare(orc, scene, options_kw=None, outlier=None, iterations=None):
    from snake_slam_amd.ba import BARec, lba_options

    kw = options_kw or {}
    ba = BARec(lba_options(**kw))
    ba.create(scene)
    if outlier is not None:
        ba.set_outliers(0, outlier)
    chi = ba.residuals(0)
    want_chi = orc.ba_chi2(scene, outlier)
    assert np.allclose(chi, want_chi, rtol=1e-12, atol=1e-12)
    ci, cf = ba.solve(iterations)
    pose, pt, pcg = ba.state(0)
    wpose, wpt, wci, wcf, wpcg = orc.ba_solve(scene, orc.ba_options(**kw), iterations=iterations, outlier=outlier)
    assert abs(ci[0] - wci) <= 1e-9 * max(1.0, wci)
    assert abs(cf[0] - wcf) <= 1e-7 * max(1.0, wcf)
    assert rmse(pt, wpt) <= TOL, rmse(pt, wpt)
    assert rmse(pose[:, 4:], wpose[:, 4:]) <= TOL
    assert rmse(pose[:, :4], wpose[:, :4]) <= TOL
    ba.close()
    return ci[0], cf[0], pose, pt


def test_benchmark_scene_parity(orc):
    from snake_slam_amd import synth

    sc, gt = synth.ba_scene()
    ci, cf, pose, pt = compare(orc, sc)
    assert cf < 0.05 * ci
    assert np.array_equal(pose[0], sc["pose"][0])


def test_converged_optimum_matches_scipy(orc):
    """The HIP solver run to convergence against an INDEPENDENT minimiser of the same robust cost (tests/ba_scipy.py:
    scipy.optimize.least_squares with a complex-step Jacobian; the oracle is not involved): cost to 1e-12 relative, poses / points
    to 1e-8.  The CPU suite holds the same pin for the oracle (tests/test_oracle_pin.py::test_ba_optimum_against_scipy)."""
    import ba_scipy
    from scipy.spatial.transform import Rotation
    from snake_slam_amd import synth
    from snake_slam_amd.ba import BARec, lba_options

    sc, _ = synth.ba_scene(n_kf=6, n_pt=120, obs_per_pt=4, seed=61, outlier_frac=0.05)
    R, t, pt, cost, _ = ba_scipy.optimum(sc)
    ba = BARec(lba_options(max_iterations=60, max_pcg_iterations=2000, pcg_tol=1e-14))
    ba.create(sc)
    ci, cf = ba.initAndSolve()
    pose, got_pt, _ = ba.state(0)
    ba.close()
    assert abs(cf[0] - cost) <= 1e-12 * cost, (cf[0], cost)
    assert np.abs(Rotation.from_quat(pose[:, :4]).as_matrix() - R).max() <= 1e-8
    # poses to 1e-8; points to 1e-8 RMSE (the specification's measure) and 1e-7 each: a point seen under a narrow angle sits in a
    # flat valley of the cost, where the converged LM of another summation order stops a few 1e-8 away (measured 3.4e-8 on one of 120)
    assert np.abs(pose[:, 4:] - t).max() <= 1e-8 and rmse(got_pt, pt) <= 1e-8 and np.abs(got_pt - pt).max() <= 1e-7


def test_small_and_degenerate_scenes(orc):
    from snake_slam_amd import synth

    compare(orc, synth.ba_scene(n_kf=6, n_pt=120, obs_per_pt=4, seed=11)[0])
    compare(orc, synth.ba_scene(n_kf=3, n_pt=7, obs_per_pt=3, seed=2)[0])
    compare(orc, synth.ba_scene(n_kf=36, n_pt=900, obs_per_pt=10, seed=3, n_fixed=6)[0])  # full-size LBA window
    compare(orc, synth.ba_scene(n_kf=10, n_pt=400, obs_per_pt=6, seed=4, stereo_frac=0.0, n_fixed=2)[0])  # mono only
    compare(orc, synth.ba_scene(n_kf=10, n_pt=400, obs_per_pt=6, seed=5, stereo_frac=1.0)[0])


def test_noise_free_recovery(orc):
    from snake_slam_amd import synth

    sc, gt = synth.ba_scene(n_kf=8, n_pt=300, obs_per_pt=5, pixel_noise=0.0, seed=5)
    ci, cf, pose, pt = compare(orc, sc, dict(max_iterations=12))
    assert cf < 1e-9 and rmse(pt, gt["pt"]) < 1e-6


def test_constant_points_outliers_and_invalid_indices(orc):
    from snake_slam_amd import synth

    sc, _ = synth.ba_scene(n_kf=8, n_pt=200, obs_per_pt=5, seed=8, outlier_frac=0.05)
    sc["pt_const"][:40] = 1
    sc["obs_img"] = sc["obs_img"].copy()
    sc["obs_img"][5] = -1
    sc["obs_pt"] = sc["obs_pt"].copy()
    sc["obs_pt"][9] = 10**6
    ci, cf, pose, pt = compare(orc, sc)
    assert np.array_equal(pt[:40], sc["pt"][:40])
    # the reference's outlier round: chi-square test, flag, one more iteration (LocalBundleAdjustment.cpp:368-410)
    from snake_slam_amd.ba import BARec, lba_options

    ba = BARec(lba_options())
    ba.create(sc)
    ba.initAndSolve()
    chi = ba.residuals(0)
    stereo = sc["obs_depth"] > 0
    out = np.where(stereo, chi > 2.3**2, chi > 2.1**2).astype(np.uint8)
    assert 0 < out.sum() < len(out) // 4
    ba.close()
    compare(orc, sc, outlier=out, iterations=1)


def test_batched_windows_match_single(orc):
    from snake_slam_amd import synth
    from snake_slam_amd.ba import BARec, lba_options

    scenes = [synth.ba_scene(n_kf=5 + k, n_pt=100 + 37 * k, obs_per_pt=4, seed=20 + k)[0] for k in range(5)]
    ba = BARec(lba_options())
    ba.create(scenes)
    ci, cf = ba.initAndSolve()
    for k, sc in enumerate(scenes):
        wpose, wpt, wci, wcf, _ = orc.ba_solve(sc, orc.ba_options())
        pose, pt, _ = ba.state(k)
        assert abs(ci[k] - wci) <= 1e-9 * wci and abs(cf[k] - wcf) <= 1e-7 * wcf
        assert rmse(pt, wpt) <= TOL and rmse(pose, wpose) <= TOL
    # reset restores the initial state; a second solve is bit-reproducible
    p1 = [ba.state(k) for k in range(5)]
    ba.reset()
    ci2, cf2 = ba.initAndSolve()
    assert np.array_equal(ci, ci2) and np.array_equal(cf, cf2)
    for k in range(5):
        pose, pt, _ = ba.state(k)
        assert np.array_equal(pose, p1[k][0]) and np.array_equal(pt, p1[k][1])
    ba.close()


def test_bench_shape_batch_of_256_windows(orc):
    """bench.py's BA leg: 256 windows of the 20 x 2000 x 8 scene in ONE launch sequence (this shape takes the point-major
    schur_set<4,2> path because max_set_items * B >= 256).  Windows 0, 127 and 255 are distinct scenes and are compared
    with the oracle; the windows that share a scene must agree bit for bit (same arithmetic wherever they sit in the batch)."""
    from snake_slam_amd import synth
    from snake_slam_amd.ba import BARec, lba_options

    distinct = [synth.ba_scene(seed=synth.SEED + k)[0] for k in range(4)]
    special = {0: synth.ba_scene(seed=9100)[0], 127: synth.ba_scene(seed=9227)[0], 255: synth.ba_scene(seed=9355)[0]}
    scenes = [special.get(k, distinct[k % 4]) for k in range(256)]
    ba = BARec(lba_options())
    ba.create(scenes)
    ci, cf = ba.initAndSolve()
    for k, sc in special.items():
        wpose, wpt, wci, wcf, _ = orc.ba_solve(sc, orc.ba_options())
        pose, pt, _ = ba.state(k)
        assert abs(ci[k] - wci) <= 1e-9 * wci and abs(cf[k] - wcf) <= 1e-7 * wcf, k
        assert rmse(pt, wpt) <= TOL and rmse(pose, wpose) <= TOL, k
    for k, k2 in ((1, 5), (2, 6), (3, 251), (126, 130), (250, 254)):
        assert ci[k] == ci[k2] and cf[k] == cf[k2]
        a, b = ba.state(k), ba.state(k2)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    ba.close()


def test_global_ba_scale_and_pose_only(orc):
    """GlobalBundleAdjustment uses the same solver seam with a bigger scene
    (reference Snake/Optimizer/GlobalBundleAdjustment.cpp:32-43: PCG 40), and PoseRefinement is the
    same residual with every point constant."""
    from snake_slam_amd import synth

    # "global" scene: 120 keyframes, 6000 points, reduced system 714 x 714 (S stays in global memory)
    sc, gt = synth.ba_scene(n_kf=120, n_pt=6000, obs_per_pt=10, seed=31, n_fixed=1)
    ci, cf, pose, pt = compare(orc, sc, dict(max_iterations=4, max_pcg_iterations=40))
    assert cf < 0.05 * ci
    # pose-only: all points constant, cameras free except the first
    sc2, _ = synth.ba_scene(n_kf=6, n_pt=400, obs_per_pt=6, seed=32)
    sc2["pt_const"][:] = 1
    ci, cf, pose, pt = compare(orc, sc2)
    assert np.array_equal(pt, sc2["pt"]) and cf < ci


@pytest.mark.parametrize("n_kf", [26, 86, 87, 150, 172, 173, 342, 343])
def test_global_ba_register_pcg_boundary_sizes(orc, n_kf):
    """The one-launch PCG with the rows of S in registers (pcgl_persist_reg) at the sizes where its form changes:
    just above one workgroup's PCG (25 free cameras, 19 workgroups of 8 rows), n6 = 510 / 516 (the last size with 8 and the first with 16
    rows per workgroup), 894, 1026, 1032, 2046 (the largest system it takes, last workgroup 14 rows) and 2052 (back to the
    streaming pcgl_persist1).  Same tolerance against the oracle as every global scene.  Match: GlobalBundleAdjustment.cpp:32-43."""
    from snake_slam_amd import synth

    sc, gt = synth.ba_scene(n_kf=n_kf, n_pt=12 * n_kf, obs_per_pt=6, seed=900 + n_kf, n_fixed=1)
    ci, cf, pose, pt = compare(orc, sc, dict(max_iterations=2, max_pcg_iterations=40))
    assert cf < ci


def test_global_ba_multi_workgroup_pcg(orc):
    """Reduced systems beyond one workgroup's LDS (here 299 free cameras, S = 1794 x 1794) take the
    multi-workgroup PCG (vectors in HBM, fixed-order partial sums); same tolerance vs the oracle."""
    from snake_slam_amd import synth

    sc, gt = synth.ba_scene(n_kf=300, n_pt=9000, obs_per_pt=8, seed=41, n_fixed=1)
    ci, cf, pose, pt = compare(orc, sc, dict(max_iterations=3, max_pcg_iterations=40))
    assert cf < 0.05 * ci
    # a batch of two mid-size problems on the same path (per-problem partial sums must not mix)
    from snake_slam_amd.ba import BARec, lba_options

    scenes = [synth.ba_scene(n_kf=60, n_pt=1500, obs_per_pt=6, seed=42 + k)[0] for k in range(2)]
    ba = BARec(lba_options(max_iterations=2, max_pcg_iterations=30))
    ba.create(scenes)
    ci, cf = ba.initAndSolve()
    for k in range(2):
        wpose, wpt, wci, wcf, _ = orc.ba_solve(scenes[k], orc.ba_options(2, 30))
        pose, pt, _ = ba.state(k)
        assert abs(cf[k] - wcf) <= 1e-7 * wcf and rmse(pose, wpose) <= TOL and rmse(pt, wpt) <= TOL
    ba.close()


def test_point_only_ba(orc):
    """GlobalBundleAdjustment::PointBA (reference Snake/Optimizer/GlobalBundleAdjustment.cpp:103-122):
    every image constant, only the points move (no reduced camera system at all)."""
    from snake_slam_amd import synth

    sc, _ = synth.ba_scene(n_kf=8, n_pt=500, obs_per_pt=5, seed=51)
    sc["img_const"][:] = 1
    ci, cf, pose, pt = compare(orc, sc, dict(max_iterations=4))
    assert np.array_equal(pose, sc["pose"]) and cf < ci


def test_relative_pose_constraints(orc):
    """IMU scenes: Saiga RelPoseConstraint terms between consecutive keyframes (reference
    LocalBundleAdjustment.cpp:294-346) -- camera-camera blocks of the reduced system, cost, trial cost."""
    from snake_slam_amd import synth
    from snake_slam_amd.ba import BARec, lba_options

    sc, gt = synth.ba_scene(n_kf=10, n_pt=400, obs_per_pt=5, seed=61, n_fixed=1)
    base = compare(orc, sc)
    synth.ba_add_rpcs(sc, gt, seed=62)
    ci, cf, pose, pt = compare(orc, sc)
    assert ci > base[0] and cf < ci
    # stiff constraints change the solution visibly and still match the oracle; two fixed images, one
    # constraint between them (ignored), one reversed pair, one duplicate pair (chained on the same block)
    sc2, gt2 = synth.ba_scene(n_kf=8, n_pt=300, obs_per_pt=4, seed=63, n_fixed=2)
    synth.ba_add_rpcs(sc2, gt2, seed=64, weight_rotation=400.0, weight_translation=150.0)
    r = sc2["rpc"].copy()
    rev = r[4:5].copy()
    rev["img1"], rev["img2"] = r[4]["img2"], r[4]["img1"]
    R = synth.quat_to_R(r[4]["rel_pose"][:4])
    rev["rel_pose"][0, :3] = -r[4]["rel_pose"][:3]
    rev["rel_pose"][0, 3] = r[4]["rel_pose"][3]
    rev["rel_pose"][0, 4:] = -R.T @ r[4]["rel_pose"][4:]
    sc2["rpc"] = np.concatenate([r, rev, r[5:6]])
    compare(orc, sc2, dict(max_iterations=4))
    # batched windows with and without constraints side by side; reset reproduces bit for bit
    scenes = []
    for k in range(3):
        s, g = synth.ba_scene(n_kf=6, n_pt=150, obs_per_pt=4, seed=70 + k)
        if k != 1:
            synth.ba_add_rpcs(s, g, seed=80 + k)
        scenes.append(s)
    ba = BARec(lba_options())
    ba.create(scenes)
    ci, cf = ba.initAndSolve()
    for k in range(3):
        wpose, wpt, wci, wcf, _ = orc.ba_solve(scenes[k], orc.ba_options())
        pose, pt, _ = ba.state(k)
        assert abs(ci[k] - wci) <= 1e-9 * wci and abs(cf[k] - wcf) <= 1e-7 * wcf
        assert rmse(pose, wpose) <= TOL and rmse(pt, wpt) <= TOL
    ba.reset()
    ci2, cf2 = ba.initAndSolve()
    assert np.array_equal(ci, ci2) and np.array_equal(cf, cf2)
    ba.close()


def test_irregular_camera_sets_and_point_major_limits(orc):
    """Shapes around the limits of the point-major Schur pass (<= 10 free-camera observations out of <= 14 per point, a
    camera at most once per point): random camera subsets (hundreds of different camera sets, most of them with a
    single point), points with 12 observations, one camera twice on a point.  Whatever pass is chosen must agree
    with the oracle (tests/test_zy_ba_variants_gpu.py runs this file with each pass forced)."""
    from snake_slam_amd import synth

    rng = np.random.default_rng(77)
    # random subsets: every point picks 3..9 of 12 cameras
    sc, _ = synth.ba_scene(n_kf=12, n_pt=300, obs_per_pt=9, seed=81, n_fixed=1)
    keep = np.ones(len(sc["obs_img"]), bool)
    for p in range(300):
        idx = np.nonzero(sc["obs_pt"] == p)[0]
        drop = rng.permutation(idx)[: int(rng.integers(0, 7))]
        keep[drop] = False
    sub = dict(sc)
    for k in ("obs_img", "obs_pt", "obs_uv", "obs_depth", "obs_weight"):
        sub[k] = sc[k][keep]
    compare(orc, sub)
    # 12 observations per point: beyond the pair-slot limit
    compare(orc, synth.ba_scene(n_kf=14, n_pt=150, obs_per_pt=12, seed=82, n_fixed=1)[0])
    # one camera observes a point twice (duplicated observations)
    sc2, _ = synth.ba_scene(n_kf=6, n_pt=80, obs_per_pt=4, seed=83)
    dup = dict(sc2)
    for k in ("obs_img", "obs_pt", "obs_uv", "obs_depth", "obs_weight"):
        dup[k] = np.concatenate([sc2[k], sc2[k][:5]])
    compare(orc, dup)


def test_handles_release_their_device_memory(orc):
    """create / set_problem / solve / destroy in a loop must not leak device memory (every buffer of the handle is
    released by snk_ba_destroy)."""
    import torch

    from snake_slam_amd import synth
    from snake_slam_amd.ba import BARec, lba_options

    sc, gt = synth.ba_scene(n_kf=12, n_pt=800, obs_per_pt=6, seed=91)
    sc_rpc = synth.ba_add_rpcs(dict(sc), gt, seed=3)

    def cycle(scene):
        ba = BARec(lba_options())
        ba.create([scene] * 8)
        ba.solve(2)
        ba.close()

    cycle(sc), cycle(sc_rpc)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(25):
        cycle(sc), cycle(sc_rpc)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, (free0 - free1) >> 20  # a leak of the small tables alone was ~1.5 MB per cycle


def test_point_only_and_pose_only_adaptors_at_global_scale(orc):
    """`BAPointOnly` / `BAPoseOnly` (reference Snake/Optimizer/GlobalBundleAdjustment.cpp:103-122, 306-316) on a global-BA
    sized scene (200 keyframes, 8000 points): the adaptors hold every image / every point of the scene handed to create()
    and leave the caller's scene untouched; result = the oracle on the scene with those flags set."""
    from snake_slam_amd import synth
    from snake_slam_amd.ba import BAPointOnly, BAPoseOnly, gba_options

    sc, _ = synth.ba_scene(n_kf=200, n_pt=8000, obs_per_pt=8, seed=71, n_fixed=1)
    for cls, hold in ((BAPointOnly, "img_const"), (BAPoseOnly, "pt_const")):
        ba = cls(gba_options())
        ba.create(sc)
        ci, cf = ba.initAndSolve()
        pose, pt, _ = ba.state(0)
        ba.close()
        assert not sc["pt_const"].any() and int(sc["img_const"].sum()) == 1  # caller's flags untouched
        held = dict(sc)
        held[hold] = np.ones_like(sc[hold])
        wpose, wpt, wci, wcf, _ = orc.ba_solve(held, orc.ba_options(4, 40))
        assert abs(ci[0] - wci) <= 1e-9 * wci and abs(cf[0] - wcf) <= 1e-7 * wcf and cf[0] < 0.6 * ci[0]
        assert rmse(pose, wpose) <= TOL and rmse(pt, wpt) <= TOL
        if hold == "img_const":
            assert np.array_equal(pose, sc["pose"])
        else:
            assert np.array_equal(pt, sc["pt"])


def _local_scene_by_steps(ba, sc, problem, pre):
    """The reference's sequence call by call (LocalBundleAdjustment.cpp:357-410), the thresholding on the host."""
    if pre is not None:
        ba.set_outliers(problem, pre)
    ci, cf = ba.initAndSolve()
    chi = ba.residuals(problem)
    thr = np.where(np.asarray(sc["obs_depth"]) > 0, 2.3**2, 2.1**2)
    mark = chi > thr  # chi is 0 for invalid observations and for those already flagged
    flags = mark.astype(np.uint8) if pre is None else (mark | (pre != 0)).astype(np.uint8)
    if mark.any():
        ba.set_outliers(problem, flags)
        ba.solve(1)
    pose, pt, _ = ba.state(problem)
    return int(mark.sum()), ci[problem], cf[problem], pose, pt, flags


@pytest.mark.parametrize("case", ["outliers", "clean", "preset", "invalid", "multi_workgroup_pcg"])
def test_solve_local_scene_equals_the_call_sequence(orc, case):
    """snk_ba_solve_local_scene == solve -> residuals -> host threshold -> set_outliers -> solve(1) -> get_state, bit for bit
    (same kernels in the same order; only the thresholding moved to the device)."""
    from snake_slam_amd import synth
    from snake_slam_amd.ba import BARec, lba_options

    # 30 keyframes: the reduced system (174 unknowns) exceeds one workgroup's LDS -> multi-workgroup PCG, whose launch sequence
    # sizes itself on the host: the extra iteration is decided after reading the count back
    n_kf, n_pt = (30, 900) if case == "multi_workgroup_pcg" else (8, 300)
    sc, _ = synth.ba_scene(n_kf=n_kf, n_pt=n_pt, obs_per_pt=5, seed=21, outlier_frac=0.0 if case == "clean" else 0.05,
                           pixel_noise=0.05 if case == "clean" else 0.5)
    pre = None
    if case == "preset":
        pre = (np.random.default_rng(3).random(len(sc["obs_img"])) < 0.1).astype(np.uint8)
    if case == "invalid":
        sc["obs_img"] = sc["obs_img"].copy()
        sc["obs_pt"] = sc["obs_pt"].copy()
        sc["obs_img"][7] = -1
        sc["obs_pt"][11] = 10**6
        sc["obs_uv"] = sc["obs_uv"].copy()
        sc["obs_uv"][7] += 500.0   # would be far above the threshold if it were looked at
        sc["obs_uv"][11] += 500.0
    a, b = BARec(lba_options()), BARec(lba_options())
    a.create(sc)
    b.create(sc)
    want = _local_scene_by_steps(a, sc, 0, pre)
    if pre is not None:
        b.set_outliers(0, pre)
    got = b.solve_local_scene(2.1**2, 2.3**2)
    assert got[0] == want[0] and (got[0] == 0) == (case == "clean")
    assert got[1] == want[1] and got[2] == want[2]
    assert np.array_equal(got[3], want[3]) and np.array_equal(got[4], want[4])
    assert np.array_equal(got[5], want[5])
    if case == "invalid":
        assert got[5][7] == 0 and got[5][11] == 0
    # and the oracle agrees with the whole sequence
    wpose, wpt, _, _, _ = orc.ba_solve(sc, orc.ba_options(), iterations=3, outlier=pre)
    if got[0]:
        sc2 = dict(sc, pose=wpose, pt=wpt)
        wpose, wpt, _, _, _ = orc.ba_solve(sc2, orc.ba_options(), iterations=1, outlier=got[5])
    assert rmse(got[4], wpt) <= TOL and rmse(got[3][:, 4:], wpose[:, 4:]) <= TOL and rmse(got[3][:, :4], wpose[:, :4]) <= TOL
    # a second scene through the same handles: counts and staging are reset
    sc3, _ = synth.ba_scene(n_kf=6, n_pt=150, obs_per_pt=4, seed=22, outlier_frac=0.1)
    a.create(sc3)
    b.create(sc3)
    want = _local_scene_by_steps(a, sc3, 0, None)
    got = b.solve_local_scene(2.1**2, 2.3**2)
    assert got[0] == want[0] > 0 and np.array_equal(got[3], want[3]) and np.array_equal(got[4], want[4]) and np.array_equal(got[5], want[5])
    a.close()
    b.close()


def test_solve_local_scene_in_a_batch(orc):
    """Several windows loaded: every window is solved and marked; the outputs are those of the window asked for."""
    from snake_slam_amd import synth
    from snake_slam_amd.ba import BARec, lba_options

    scs = [synth.ba_scene(n_kf=6, n_pt=120 + 30 * i, obs_per_pt=4, seed=40 + i, outlier_frac=0.05 * i)[0] for i in range(3)]
    for problem in (0, 2):
        one, many = BARec(lba_options()), BARec(lba_options())
        one.create(scs[problem])
        many.create(scs)
        want = one.solve_local_scene(2.1**2, 2.3**2)
        got = many.solve_local_scene(2.1**2, 2.3**2, problem=problem)
        assert got[0] == want[0]
        assert rmse(got[3], want[3]) <= 1e-9 and rmse(got[4], want[4]) <= 1e-9 and np.array_equal(got[5], want[5])
        one.close()
        many.close()
    import ctypes as C

    ba = BARec(lba_options())
    n = C.c_int(0)
    assert ba._lib.snk_ba_solve_local_scene(ba._h, 0, 4.0, 5.0, 1, None, C.byref(n), None, None, None, None) != 0  # no scene loaded
    ba.create(scs[0])
    assert ba._lib.snk_ba_solve_local_scene(ba._h, 1, 4.0, 5.0, 1, None, C.byref(n), None, None, None, None) != 0  # index out of range
    assert ba._lib.snk_ba_solve_local_scene(ba._h, 0, 4.0, 5.0, 1, None, None, None, None, None, None) != 0        # n_marked is required
    assert ba._lib.snk_ba_solve_local_scene(ba._h, 0, 4.0, 5.0, 1, None, C.byref(n), None, None, None, None) == 0  # every output optional
    ba.close()


@pytest.mark.parametrize("large", [False, True])
def test_solve_local_scene_decides_per_problem_on_every_path(orc, large):
    """A batch in which one window has marks and another has none: the extra iteration (LocalBundleAdjustment.cpp:399-410) runs for
    the marked windows ONLY -- whichever window the caller asks about, on the device path (default), on the host-decided path
    (SNK_BA_LOCAL_SYNC=1 in the variants matrix) and for scenes on the multi-workgroup PCG (large: 30 keyframes, always host-decided).
    ADVICE round 3: the host path used to look at the asked-for window alone and then re-solve ALL windows or none."""
    from snake_slam_amd import synth
    from snake_slam_amd.ba import BARec, lba_options

    dirty, _ = synth.ba_scene(n_kf=30 if large else 8, n_pt=900 if large else 300, obs_per_pt=5, seed=31, outlier_frac=0.05)
    clean, _ = synth.ba_scene(n_kf=7, n_pt=200, obs_per_pt=4, seed=32, outlier_frac=0.0, pixel_noise=0.05)
    dirty2, _ = synth.ba_scene(n_kf=6, n_pt=150, obs_per_pt=4, seed=33, outlier_frac=0.1)
    scs = [clean, dirty, dirty2]
    singles = []
    for sc in scs:
        one = BARec(lba_options())
        one.create(sc)
        singles.append(one.solve_local_scene(2.1**2, 2.3**2))
        one.close()
    assert singles[0][0] == 0 and singles[1][0] > 0 and singles[2][0] > 0
    for ask in (0, 1):  # asking about the clean window must not skip the others; asking about a marked one must not touch the clean one
        many = BARec(lba_options())
        many.create(scs)
        got = many.solve_local_scene(2.1**2, 2.3**2, problem=ask)
        assert got[0] == singles[ask][0]
        for k in range(3):
            pose, pt, _ = many.state(k)
            assert rmse(pose, singles[k][3]) <= 1e-9 and rmse(pt, singles[k][4]) <= 1e-9, (ask, k)
        many.close()


def test_solve_local_scene_degenerate_scenes(orc):
    """Nothing to optimise (every camera constant), nothing to observe (no observations), a single point: the call returns the
    scene as it was handed over (or the step-by-step result) and marks nothing it should not."""
    from snake_slam_amd import synth
    from snake_slam_amd.ba import BARec, lba_options

    base, _ = synth.ba_scene(n_kf=4, n_pt=40, obs_per_pt=3, seed=5)
    allc = dict(base, img_const=np.ones_like(base["img_const"]), pt_const=np.ones_like(base["pt_const"]))
    empty = dict(base, obs_img=base["obs_img"][:0], obs_pt=base["obs_pt"][:0], obs_uv=base["obs_uv"][:0], obs_depth=base["obs_depth"][:0],
                 obs_weight=base["obs_weight"][:0])
    one, _ = synth.ba_scene(n_kf=3, n_pt=1, obs_per_pt=3, seed=6)
    for sc in (allc, empty, one):
        a, b = BARec(lba_options()), BARec(lba_options())
        a.create(sc)
        b.create(sc)
        want = _local_scene_by_steps(a, sc, 0, None)
        got = b.solve_local_scene(2.1**2, 2.3**2)
        assert got[0] == want[0]
        assert np.array_equal(got[3], want[3]) and np.array_equal(got[4], want[4]) and np.array_equal(got[5], want[5])
        if sc is allc or sc is empty:
            assert np.array_equal(got[3], sc["pose"]) and np.array_equal(got[4], sc["pt"]) and got[0] == 0
        a.close()
        b.close()


def test_big_batch_of_unequal_scenes(orc):
    """Batches of >= 256 problems cut their camera sets into work items of up to 128 points (two list registers in schur_fused /
    update_cost): 300 scenes of different sizes -- sets of 1 .. ~190 points, runs of 2 .. 8 -- in one batch, EVERY scene against
    the oracle under the parity rule of tests/ba_parity.py: strict tolerances, or -- for scenes whose PCG ran into the reference's
    30-iteration limit on both sides -- strict tolerances once both sides may converge.  Round 3 ended red on this test: it sampled 24
    scenes at strict tolerance, and under SNK_BA_NO_SCHUR_SET=1 sample 195 (8 keyframes x 344 points x 3 observations, PCG 90 vs 90)
    was 1.3e-7 apart in cost; profiles/r04/r04a_diag_big_batch.log shows 3-6 such scenes of the 300 on EVERY path including the default
    one, all of them at 1e-11 RMSE with a converged PCG -- truncation sensitivity, not a path defect."""
    from ba_parity import check_scene
    from snake_slam_amd import synth
    from snake_slam_amd.ba import BARec, lba_options

    rng = np.random.default_rng(17)
    scenes = []
    for k in range(300):
        n_kf = int(rng.integers(3, 9))
        opp = int(rng.integers(2, n_kf + 1))
        n_min = max(24, -(-8 * n_kf // opp))
        n_pt = int(rng.choice([n_min, max(n_min, 65 * n_kf // 2), max(n_min, 129 * n_kf // 3), int(rng.integers(n_min, n_min + 400))]))
        sc, _ = synth.ba_scene(n_kf=n_kf, n_pt=n_pt, obs_per_pt=opp, seed=1000 + k, n_fixed=int(rng.integers(1, 3)), outlier_frac=0.02)
        if k % 7 == 0:
            sc["pt_const"] = (rng.random(n_pt) < 0.15).astype(np.uint8)
        scenes.append(sc)
    ba = BARec(lba_options())
    ba.create(scenes)
    ci, cf = ba.initAndSolve()
    states = [ba.state(k) for k in range(len(scenes))]
    ba.close()
    truncated = []
    for k, sc in enumerate(scenes):
        pose, pt, pcg = states[k]
        kind, text, _ = check_scene(orc, sc, (ci[k], cf[k], pose, pt, pcg))
        assert kind != "fail", f"scene {k}: {text}"
        if kind == "truncated":
            truncated.append(k)
    assert len(truncated) <= 12, truncated  # 3 .. 6 of 300 measured, depending on the path


def test_hand_over_threads_end_with_the_handle(orc):
    """The host threads that build the lists of a batch hand-over are parked in the handle between hand-overs (HostPool in
    csrc/ba.hip) -- they must not outlive it, and repeated hand-overs must not add more.  Linux: /proc/self/status."""
    from snake_slam_amd import synth
    from snake_slam_amd.ba import BARec, lba_options

    def n_threads():
        for line in open("/proc/self/status"):
            if line.startswith("Threads:"):
                return int(line.split()[1])
        return -1

    scenes = [synth.ba_scene(n_kf=5 + k % 3, n_pt=90 + 11 * k, obs_per_pt=4, seed=300 + k)[0] for k in range(32)]
    warm = BARec(lba_options(max_iterations=1, max_pcg_iterations=10))  # runtime threads of the first use of the device
    warm.create(scenes[:1])
    warm.initAndSolve()
    warm.close()
    before = n_threads()
    ba = BARec(lba_options(max_iterations=2, max_pcg_iterations=30))
    ba.create(scenes)
    during = n_threads()
    for _ in range(3):
        ba.create(scenes)
    assert n_threads() == during
    ci, cf = ba.initAndSolve()
    k = 17
    wpose, wpt, wci, wcf, _ = orc.ba_solve(scenes[k], orc.ba_options(2, 30))
    pose, pt, _ = ba.state(k)
    assert rmse(pose, wpose) <= TOL and rmse(pt, wpt) <= TOL
    ba.close()
    if "SNK_BA_NO_HOST_POOL" not in os.environ:  # (the variants matrix runs this file with threads created and joined per pass as well)
        assert during > before, "a batch of 32 scenes is expected to use the threaded list builder"
    assert n_threads() <= before
