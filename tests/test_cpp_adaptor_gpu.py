"""GPU: the C++ adaptor header (snake_slam_amd/cpp/snake_hip.hpp), which is what a Snake-SLAM maintainer includes, is
built into a small driver (tests/cpp/adaptor_driver.cpp, plain g++ -- no HIP in the translation unit) and EXECUTED on
the golden inputs: ORBExtractor::Detect -> Rectify -> StereoMatching -> matchKnn2 / filterMatches -> the projection
matchers -> optimizePoseRobust -> BARec.  Every dumped result must equal tests/golden/*.npz."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
G = ROOT / "tests" / "golden"


def build_driver(out_dir: Path) -> Path:
    lib = ROOT / "snake_slam_amd" / "lib"
    exe = out_dir / "adaptor_driver"
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", f"-I{ROOT / 'include'}", f"-I{ROOT / 'snake_slam_amd' / 'cpp'}",
           str(ROOT / "tests" / "cpp" / "adaptor_driver.cpp"), f"-L{lib}", "-lsnake_hip", "-L/opt/rocm/lib", "-lamdhip64",
           f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def write_inputs(d: Path):
    from snake_slam_amd.matcher import Rectification

    def put(name, a):
        np.ascontiguousarray(a).tofile(d / f"{name}.bin")

    g = np.load(G / "orb_small.npz")
    h, w = g["img"].shape
    nfeat, _, nl, ini, mn = g["params"]
    put("orb_dims", np.array([w, h, nfeat, nl, ini, mn], np.int32))
    put("orb_img", g["img"])
    g = np.load(G / "rectify_small.npz")
    rect = Rectification.make(g["K"], g["D"], g["R"], g["Kd"])
    (d / "rect.bin").write_bytes(bytes(rect))
    put("rect_kps", g["kps"])
    g = np.load(G / "match_small.npz")
    for k, n in (("left", "st_left"), ("right", "st_right"), ("dl", "st_dl"), ("dr", "st_dr"), ("q", "bf_q"), ("t", "bf_t")):
        put(n, g[k])
    put("st_bf", np.array([float(g["bf"])], np.float64))
    put("st_ls", g["ls"].astype(np.float32))
    g = np.load(G / "track_small.npz")
    put("tr_kps", g["f_kps"]), put("tr_desc", g["f_desc"]), put("tr_rp", g["f_right_points"]), put("tr_taken", g["f_taken"])
    put("tr_bounds", np.asarray(g["f_bounds"], np.float64))
    put("tr_cam", np.asarray(g["cam"], np.float64)), put("tr_pose", np.asarray(g["pose"], np.float64))
    put("tr_ls", g["ls"].astype(np.float32)), put("tr_coarse", g["coarse_pts"]), put("tr_fine", g["fine_pts"])
    g = np.load(G / "pose_small.npz")
    put("po_cam", np.asarray(g["cam"], np.float64)), put("po_pose0", g["pose0"]), put("po_wps", g["wps"]), put("po_obs", g["obs"])
    g = np.load(G / "ba_small.npz")
    for k, dt in (("pose", np.float64), ("img_const", np.uint8), ("pt", np.float64), ("pt_const", np.uint8), ("obs_img", np.int32),
                  ("obs_pt", np.int32), ("obs_uv", np.float64), ("obs_depth", np.float64), ("obs_weight", np.float64),
                  ("K", np.float64)):
        put(f"ba_{k}", np.asarray(g[f"in_{k}"], dt))
    put("ba_bf", np.array([float(g["in_bf"])], np.float64))


def test_cpp_adaptor_runs_the_pipeline_and_matches_golden(tmp_path, orc):
    exe = build_driver(tmp_path)
    write_inputs(tmp_path)
    r = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)

    def get(name, dtype):
        return np.fromfile(tmp_path / f"out_{name}.bin", dtype)

    from snake_slam_amd.orb import KEYPOINT_DTYPE

    g = np.load(G / "orb_small.npz")
    assert np.array_equal(get("orb_kps", KEYPOINT_DTYPE), g["kps"])
    assert np.array_equal(get("orb_desc", np.uint64).reshape(-1, 4), g["desc"])
    # the half-width view of the same buffer (pitch > width) against the oracle
    hk, _ = orc.orb_detect(orc.orb_params(*[int(v) if i != 1 else float(v) for i, v in enumerate(g["params"])]),
                           np.ascontiguousarray(g["img"][:, : g["img"].shape[1] // 2]))
    assert np.array_equal(get("orb_kps_half", KEYPOINT_DTYPE), hk)
    g = np.load(G / "rectify_small.npz")
    assert get("rect", g["out"].dtype).tobytes() == g["out"].tobytes()
    assert np.array_equal(get("rect_norm", np.float64).reshape(-1, 2), g["norm"])
    g = np.load(G / "match_small.npz")
    assert int(get("st_n", np.int32)[0]) == int(g["n"])
    assert np.array_equal(get("st_rp", np.float32), g["rp"]) and np.array_equal(get("st_dp", np.float32), g["dp"])
    assert np.array_equal(get("bf_knn", np.int32).reshape(-1, 4), g["knn"])
    assert np.array_equal(get("bf_pairs", np.int32).reshape(-1, 2), g["pairs"])
    g = np.load(G / "track_small.npz")
    assert np.array_equal(get("tr_perm", np.int32), np.arange(len(g["f_kps"])))
    assert np.array_equal(get("tr_cell_start", np.int32), g["f_cell_start"])
    c = get("tr_coarse", np.int32)
    assert c[-1] == int(g["coarse_n"]) and np.array_equal(c[:-1], g["coarse_idx"])
    f = get("tr_fine", np.int32)
    assert f[-1] == int(g["fine_n"]) and np.array_equal(f[:-1], g["fine_idx"])
    assert np.array_equal(get("tr_fine_vis", np.uint8), g["fine_vis"]) and np.array_equal(get("tr_fine_valid", np.uint8), g["fine_valid"])
    g = np.load(G / "pose_small.npz")
    assert np.allclose(get("po_pose", np.float64), g["pose"], rtol=0, atol=1e-9)
    assert np.array_equal(get("po_outlier", np.uint8), g["outlier"]) and int(get("po_inliers", np.int32)[0]) == int(g["inliers"])
    g = np.load(G / "ba_small.npz")
    rm = lambda a, b: float(np.sqrt(((a - b) ** 2).sum(-1).mean()))
    assert np.allclose(get("ba_chi2", np.float64), g["chi2"], rtol=1e-12, atol=1e-15)
    assert rm(get("ba_pose", np.float64).reshape(-1, 7), g["pose"]) <= 1e-5 and rm(get("ba_pt", np.float64).reshape(-1, 3), g["pt"]) <= 1e-5
    assert np.allclose(get("ba_cost", np.float64), g["cost"], rtol=1e-7)
    # second solve after the chi-square pass: the oracle with the same outlier mask, starting from the first result
    sc = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    sc["pose"], sc["pt"] = g["pose"], g["pt"]
    outl = get("ba_outlier", np.uint8)
    chi = orc.ba_chi2(sc)
    assert np.array_equal(outl, (chi > np.where(sc["obs_depth"] > 0, 5.29, 4.41)).astype(np.uint8))
    wpose, _, wc0, wc1, _ = orc.ba_solve(sc, orc.ba_options(max_iterations=1), outlier=outl)
    c2 = get("ba_cost2", np.float64)
    assert int(c2[2]) == int(outl.sum()) and np.allclose(c2[:2], [wc0, wc1], rtol=1e-7)
    assert rm(get("ba_pose2", np.float64).reshape(-1, 7), wpose) <= 1e-5
    # BARec::solveLocalScene (snk_ba_solve_local_scene), bit for bit the call-by-call sequence of the Python mirror: with the
    # reference's thresholds this scene has nothing to mark (no extra iteration: the first solve's result), with low ones it has
    from snake_slam_amd.ba import BARec, lba_options

    s_in = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    for tag, (t_mono, t_stereo) in (("ba_fused", (4.41, 5.29)), ("ba_fused_low", (0.3, 0.4))):
        ba = BARec(lba_options())
        ba.create(s_in)
        ci, cf_ = ba.initAndSolve()
        mask = (ba.residuals(0) > np.where(s_in["obs_depth"] > 0, t_stereo, t_mono)).astype(np.uint8)
        if mask.any():
            ba.set_outliers(0, mask)
            ba.solve(1)
        wp, wx, _ = ba.state(0)
        ba.close()
        assert (mask.sum() > 0) == (tag == "ba_fused_low")
        assert np.array_equal(get(tag + "_pose", np.float64).reshape(-1, 7), wp) and np.array_equal(get(tag + "_pt", np.float64).reshape(-1, 3), wx)
        assert np.array_equal(get(tag + "_outlier", np.uint8), mask)
        cfu = get(tag + "_cost", np.float64)
        assert int(cfu[2]) == int(mask.sum()) and cfu[0] == ci[0] and cfu[1] == cf_[0]
    assert np.array_equal(get("ba_fused_pose", np.float64), get("ba_pose", np.float64))
    # BAPointOnly / BAPoseOnly (GlobalBundleAdjustment.cpp:103-122, 306-316): the oracle on the scene with every image /
    # every point held, global options (4 iterations, PCG <= 40)
    s0 = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    for tag, hold in (("pba", "img_const"), ("qba", "pt_const")):
        sh = dict(s0)
        sh[hold] = np.ones_like(s0[hold])
        wpose, wpt, wc0, wc1, _ = orc.ba_solve(sh, orc.ba_options(4, 40, 1e-10, 2.1, 2.3, 0.0))
        assert np.allclose(get(f"{tag}_cost", np.float64), [wc0, wc1], rtol=1e-7) and wc1 < wc0
        gp, gq = get(f"{tag}_pose", np.float64).reshape(-1, 7), get(f"{tag}_pt", np.float64).reshape(-1, 3)
        assert rm(gp, wpose) <= 1e-5 and rm(gq, wpt) <= 1e-5
        if hold == "img_const":
            assert np.array_equal(gp, s0["pose"]) and not np.array_equal(gq, s0["pt"])
        else:
            assert np.array_equal(gq, s0["pt"]) and not np.array_equal(gp, s0["pose"])
