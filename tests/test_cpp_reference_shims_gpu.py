"""GPU: the hot path through the REFERENCE'S OWN five signatures (snake_slam_amd/cpp/snake_hip_reference.hpp), driven from C++
with mock Snake / Saiga structs that have the reference's member names (tests/cpp/reference_shims_driver.cpp):
undistortKeypoints(Frame&), computeFeatureGrid(Frame&), StereoMatching(Frame&), SearchByProjectionFrameFrame2(Frame&, const
LocalMap<CoarseTrackingPoint>&, th, featureError, num_threads), SearchByProjection2(Frame&, LocalMap<FineTrackingPoint>&, ...),
SearchByProjectionFrameToKeyframe(Frame&, const Keyframe&, ...), SolveLocalScene on a Saiga::Scene.  Checked: the side effects
the reference functions have (mvpMapPoints[idx] = lmp.mp, lmp.valid, IncreaseVisible, right_points / depth, the permuted
feature arrays, o.outlier, the scene's poses / points) against the golden fixtures and the oracle."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
G = ROOT / "tests" / "golden"


def test_reference_signatures_match_golden(tmp_path, orc):
    from snake_slam_amd.matcher import KP64_DTYPE, Rectification

    lib = ROOT / "snake_slam_amd" / "lib"
    exe = tmp_path / "ref_driver"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", f"-I{ROOT / 'include'}", f"-I{ROOT / 'snake_slam_amd' / 'cpp'}",
                        str(ROOT / "tests" / "cpp" / "reference_shims_driver.cpp"), f"-L{lib}", "-lsnake_hip", "-L/opt/rocm/lib",
                        "-lamdhip64", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = tmp_path

    def put(name, a):
        np.ascontiguousarray(a).tofile(d / f"{name}.bin")

    def get(name, dtype):
        return np.fromfile(d / f"out_{name}.bin", dtype)

    rng = np.random.default_rng(2024)
    # --- Preprocess: undistortKeypoints + computeFeatureGrid on the rectification fixture
    g = np.load(G / "rectify_small.npz")
    rect = Rectification.make(g["K"], g["D"], g["R"], g["Kd"])
    (d / "rect.bin").write_bytes(bytes(rect))
    put("rect_kps", g["kps"])
    pp_desc = rng.integers(0, 2**64, (len(g["kps"]), 4), dtype=np.uint64)
    put("pp_desc", pp_desc)
    out = g["out"]
    bounds = (float(np.floor(out["x"].min()) - 5), float(np.floor(out["y"].min()) - 5), float(np.ceil(out["x"].max()) + 5), float(np.ceil(out["y"].max()) + 5))
    put("pp_bounds", np.array(bounds, np.float64))
    # --- StereoMatching(Frame&): the matcher fixture's keypoints rounded to float (what kp.cast<double>() of the extractor's
    # KeyPoint<float> holds), rectification = identity; expectation from the oracle
    m = np.load(G / "match_small.npz")
    left, right = m["left"].copy(), m["right"].copy()
    for a in (left, right):
        a["x"], a["y"] = a["x"].astype(np.float32).astype(np.float64), a["y"].astype(np.float32).astype(np.float64)
    put("st_left", left), put("st_right", right), put("st_dl", m["dl"]), put("st_dr", m["dr"]), put("st_ls", m["ls"].astype(np.float32))
    ident = Rectification.make((1.0, 1.0, 0.0, 0.0), bf=float(m["bf"]))
    (d / "st_rect.bin").write_bytes(bytes(ident) + bytes(ident))
    want_st = orc.stereo_match(left, m["dl"], right, m["dr"], float(m["bf"]), m["ls"], True)
    # --- ComputeStereoFromRGBD(Frame&): the same left keypoints as undistorted_keypoints, a depth image with a padded row pitch
    from snake_slam_amd.matcher import RgbdModel

    rg_K, rg_Kd, rg_D = (525.0, 525.0, 319.5, 239.5), (570.3, 570.3, 320.0, 240.0), (0.05, -0.1, 0.0, 0.0, 0.0, 0.0, 1e-3, -5e-4)
    dw, dh, dpitch = 800, 520, 832
    dimg = np.where(rng.random((dh, dw)) < 0.3, 0.0, rng.uniform(0.3, 19.9, (dh, dw))).astype(np.float32)
    padded = np.full((dh, dpitch), 99.0, np.float32)
    padded[:, :dw] = dimg
    put("rgbd_depth", padded), put("rgbd_dims", np.array([dw, dh, dpitch], np.int32))
    (d / "rgbd_model.bin").write_bytes(bytes(RgbdModel.make(rg_K, rg_D, rg_Kd, 40.0)))
    want_rgbd = orc.rgbd_stereo(left, rg_K, rg_D, rg_Kd, 40.0, dimg)
    # --- tracking matchers
    t = np.load(G / "track_small.npz")
    put("tr_kps", t["f_kps"]), put("tr_desc", t["f_desc"]), put("tr_rp", t["f_right_points"]), put("tr_taken", t["f_taken"])
    put("tr_bounds", np.asarray(t["f_bounds"], np.float64)), put("tr_cam", np.asarray(t["cam"], np.float64))
    put("tr_pose", np.asarray(t["pose"], np.float64)), put("tr_ls", t["ls"].astype(np.float32))
    put("tr_coarse", t["coarse_pts"]), put("tr_fine", t["fine_pts"])
    put("tr_kf_pos", t["kf_pos"]), put("tr_kf_desc", t["kf_desc"]), put("tr_kf_skip", t["kf_skip"])
    # --- local BA
    b = np.load(G / "ba_small.npz")
    for k, dt in (("pose", np.float64), ("img_const", np.uint8), ("pt", np.float64), ("pt_const", np.uint8), ("obs_img", np.int32),
                  ("obs_pt", np.int32), ("obs_uv", np.float64), ("obs_depth", np.float64), ("obs_weight", np.float64), ("K", np.float64)):
        put(f"ba_{k}", np.asarray(b[f"in_{k}"], dt))
    put("ba_bf", np.array([float(b["in_bf"])], np.float64))

    # --- the whole front-end of a stereo frame in one call: FeatureDetector::Detect + Preprocess::Process (ref::FrontEnd)
    from snake_slam_amd import synth

    fe_k1, fe_d1 = (458.654, 457.296, 367.215, 248.375), (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0, 0.0, 0.0, 0.0)
    fe_k2, fe_d2 = (457.587, 456.134, 379.999, 255.238), (-0.28368365, 0.07451284, -0.00010473, -3.55590700e-05, 0.0, 0.0, 0.0, 0.0)
    fe_bounds, fe_bf = (-120.0, -60.0, 880.0, 540.0), 47.9
    fe_left, fe_right = synth.stereo_frame(77, 752, 480)
    pitch = 768
    for name, im in (("fe_left", fe_left), ("fe_right", fe_right)):
        padded = np.full((480, pitch), 255, np.uint8)  # a row pitch larger than the width, padding that must not be read as image
        padded[:, :752] = im
        put(name, padded)
    put("fe_dims", np.array([752, 480, pitch], np.int32)), put("fe_bounds", np.array(fe_bounds, np.float64))
    (d / "fe_rect.bin").write_bytes(bytes(Rectification.make(fe_k1, fe_d1, bf=fe_bf)) + bytes(Rectification.make(fe_k2, fe_d2, bf=fe_bf)))

    r = subprocess.run([str(exe), str(d)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)

    # ref::FrontEnd::DetectAndProcess: every member FeatureDetector::Detect and Preprocess::Process fill, against the oracle chain
    ls = np.cumprod(np.array([1.0, 1.2, 1.2, 1.2], np.float32), dtype=np.float32)
    par = orc.orb_params(1000, 1.2, 4, 20, 7)
    kl, dl = orc.orb_detect(par, fe_left)
    kr, dr = orc.orb_detect(par, fe_right)
    ul, nl = orc.rectify(orc.rectification(fe_k1, fe_d1), kl)
    ur, _ = orc.rectify(orc.rectification(fe_k2, fe_d2), kr)
    perm, cell_start, _, _ = orc.feature_grid(ul, fe_bounds)
    gu, gd, gk, gn = np.zeros_like(ul), np.zeros_like(dl), np.zeros_like(kl), np.zeros_like(nl)
    gu[perm], gd[perm], gk[perm], gn[perm] = ul, dl, kl, nl
    n_st, rp, dp = orc.stereo_match(gu, gd, ur, dr, fe_bf, ls, True)
    counts = get("fe_counts", np.int32).reshape(3, 3)
    assert (counts == np.array([len(kl), len(kr), n_st])).all() and n_st > 100
    kp = get("fe_kp", np.float64).reshape(-1, 6)
    for c, f in enumerate(("x", "y", "size", "angle", "response", "octave")):
        assert np.array_equal(kp[:, c], gk[f].astype(np.float64)), f
        assert np.array_equal(get("fe_kpr", np.float64).reshape(-1, 6)[:, c], kr[f].astype(np.float64)), f
    und = get("fe_und", np.float64).reshape(-1, 4)
    assert np.array_equal(und[:, 0], gu["x"]) and np.array_equal(und[:, 1], gu["y"])
    assert np.array_equal(und[:, 2].astype(np.float32), gu["angle"]) and np.array_equal(und[:, 3].astype(np.int32), gu["octave"])
    assert np.array_equal(get("fe_norm", np.float64).reshape(-1, 2), gn)
    assert np.array_equal(get("fe_dl", np.uint64).reshape(-1, 4), gd) and np.array_equal(get("fe_dr", np.uint64).reshape(-1, 4), dr)
    assert np.array_equal(get("fe_rp", np.float32), rp) and np.array_equal(get("fe_dp", np.float32), dp)
    assert np.array_equal(get("fe_cell_start", np.int32), cell_start)

    # undistortKeypoints: undistorted_keypoints[i] = keypoints[i] with the rectified point; normalized_points
    und = get("pp_undistorted", np.float64).reshape(-1, 4)
    assert np.array_equal(und[:, 0], out["x"]) and np.array_equal(und[:, 1], out["y"])
    assert np.array_equal(und[:, 2].astype(np.float32), out["angle"]) and np.array_equal(und[:, 3].astype(np.int32), out["octave"])
    assert np.array_equal(get("pp_normalized", np.float64).reshape(-1, 2), g["norm"])
    # computeFeatureGrid: all four arrays scattered by the oracle's permutation
    perm, cell_start, _, _ = orc.feature_grid(out, bounds)
    n = len(out)
    want = np.zeros((n, 2))
    want[perm] = np.stack([g["kps"]["x"].astype(np.float64), g["kps"]["y"].astype(np.float64)], 1)
    assert np.array_equal(get("pp_keypoints", np.float64).reshape(-1, 2), want)
    want[perm] = np.stack([out["x"], out["y"]], 1)
    assert np.array_equal(get("pp_und_grid", np.float64).reshape(-1, 2), want)
    want[perm] = g["norm"]
    assert np.array_equal(get("pp_norm_grid", np.float64).reshape(-1, 2), want)
    wd = np.zeros_like(pp_desc)
    wd[perm] = pp_desc
    assert np.array_equal(get("pp_desc_grid", np.uint64).reshape(-1, 4), wd)
    assert np.array_equal(get("pp_cell_start", np.int32), cell_start)
    assert not np.array_equal(perm, np.arange(n))  # a real permutation
    # StereoMatching(Frame&)
    assert int(get("st_n", np.int32)[0]) == want_st[0] > 10
    assert np.array_equal(get("st_rp", np.float32), want_st[1]) and np.array_equal(get("st_dp", np.float32), want_st[2])
    # ComputeStereoFromRGBD(Frame&)
    assert int(get("rgbd_n", np.int32)[0]) == want_rgbd[0] > 10
    assert np.array_equal(get("rgbd_rp", np.float32), want_rgbd[1]) and np.array_equal(get("rgbd_dp", np.float32), want_rgbd[2])

    # the matchers: mvpMapPoints[idx] = lm.points[i].mp (-2 = a map point the frame already had, -1 = none)
    def check_mvp(name, idx, count, n_name):
        mvp = get(name, np.int32)
        exp = np.where(t["f_taken"] != 0, -2, -1).astype(np.int32)
        for i, f in enumerate(idx):
            if f >= 0:
                assert exp[f] == -1
                exp[f] = i
        assert np.array_equal(mvp, exp) and int(get(n_name, np.int32)[0]) == int(count) == int((idx >= 0).sum())

    check_mvp("tr_coarse_mvp", t["coarse_idx"], t["coarse_n"], "tr_coarse_n")
    check_mvp("tr_fine_mvp", t["fine_idx"], t["fine_n"], "tr_fine_n")
    assert np.array_equal(get("tr_fine_vis", np.int32), t["fine_vis"].astype(np.int32))      # IncreaseVisible once per visible point
    assert np.array_equal(get("tr_fine_valid", np.int32), t["fine_valid"].astype(np.int32))  # lmp.valid cleared by the culls
    check_mvp("tr_kf_mvp", t["kf_idx"], t["kf_n"], "tr_kf_n")
    assert int(t["coarse_n"]) > 20 and int(t["fine_n"]) > 20 and int(t["kf_n"]) > 20

    # SolveLocalScene: the oracle with the reference's sequence (3 iterations, chi-square pass, one more iteration with the mask)
    sc = {k[3:]: b[k] for k in b.files if k.startswith("in_")}
    p1, q1, c0, c1, _ = orc.ba_solve(sc, orc.ba_options())
    s1 = dict(sc, pose=p1, pt=q1)
    outl = (orc.ba_chi2(s1) > np.where(sc["obs_depth"] > 0, 2.3 * 2.3, 2.1 * 2.1)).astype(np.uint8)
    res = get("ba_res", np.float64)
    assert int(res[0]) == int(outl.sum()) and np.allclose(res[1:], [c0, c1], rtol=1e-7)
    got_pairs = set(map(tuple, get("ba_outliers", np.int32).reshape(-1, 2).tolist()))
    assert got_pairs == {(int(i), int(p)) for i, p, o in zip(sc["obs_img"], sc["obs_pt"], outl) if o}
    if outl.any():
        p2, q2, _, _, _ = orc.ba_solve(s1, orc.ba_options(max_iterations=1), outlier=outl)
    else:
        p2, q2 = p1, q1
    rm = lambda a, c: float(np.sqrt(((a - c) ** 2).sum(-1).mean()))
    assert rm(get("ba_pose", np.float64).reshape(-1, 7), p2) <= 1e-5 and rm(get("ba_pt", np.float64).reshape(-1, 3), q2) <= 1e-5
