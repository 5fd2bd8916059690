"""GPU: the front-end's alternative kernel forms, chosen by launch size in normal use, forced through their switches: the parity suites
run again in child processes so that every form sees every test input (the pattern of test_zy_ba_variants_gpu.py; sorts after the
default-path suites).
* SNK_ORB_LEVEL_BH = 64 / 22 / 8: rows per band of level_kernel (64 for big batches, 22 / 8 for the per-frame calls);
* SNK_GRID_NETWORK=1, SNK_STEREO_SORT_NETWORK=1: the bitonic networks instead of the counting forms of the feature grid / the stereo
  row index;
* SNK_TRACK_FRAME_WGS = 1 / 3: workgroups per frame of the frame-resident projection matchers;
* SNK_TRACK_PPW=64 (every batched matcher call takes the frame-resident kernels) with one workgroup per frame -- the matcher resolves its
  matches itself -- and with SNK_TRACK_NO_FUSED_RESOLVE=1 (separate resolve_batch_kernel);
* SNK_POSE_WAVES=2: two wavefronts per frame in pose_kernel (what batches of more than two frames per CU take);
* SNK_ORB_DESC_NO_DMA=1: describe_kernel's register path for the blurred patch (round 6 made the LDS-DMA form the default);
* SNK_ORB_BLUR_TILED=1: the blurred planes in 32 x 4 pixel tiles (round 6 experiment, measured slower end to end: not the default);
* SNK_ORB_HARRIS_PER_CELL=1: the Harris response with one wavefront per FAST cell (round 5) instead of the candidates of 16 cells packed
  into the lanes (round 6 default)."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("env,files", [
    ({"SNK_ORB_LEVEL_BH": "64"}, ["test_orb_gpu.py"]),
    ({"SNK_ORB_LEVEL_BH": "22"}, ["test_orb_gpu.py"]),
    ({"SNK_ORB_LEVEL_BH": "8"}, ["test_orb_gpu.py"]),
    ({"SNK_ORB_DESC_NO_DMA": "1"}, ["test_orb_gpu.py"]),
    ({"SNK_ORB_BLUR_TILED": "1"}, ["test_orb_gpu.py"]),
    ({"SNK_ORB_HARRIS_PER_CELL": "1"}, ["test_orb_gpu.py"]),
    ({"SNK_GRID_NETWORK": "1", "SNK_STEREO_SORT_NETWORK": "1"}, ["test_match_gpu.py", "test_track_gpu.py", "test_frontend_gpu.py"]),
    ({"SNK_TRACK_FRAME_WGS": "1"}, ["test_track_gpu.py", "test_tracking_chain_gpu.py"]),
    ({"SNK_TRACK_FRAME_WGS": "3"}, ["test_track_gpu.py", "test_tracking_chain_gpu.py"]),
    ({"SNK_TRACK_PPW": "64", "SNK_TRACK_FRAME_WGS": "1", "SNK_TRACK_NO_RECURSE": "1", "SNK_POSE_NO_RECURSE": "1"}, ["test_track_gpu.py", "test_tracking_chain_gpu.py"]),
    ({"SNK_TRACK_PPW": "64", "SNK_TRACK_FRAME_WGS": "1", "SNK_TRACK_NO_FUSED_RESOLVE": "1", "SNK_TRACK_NO_RECURSE": "1", "SNK_POSE_NO_RECURSE": "1"}, ["test_track_gpu.py", "test_tracking_chain_gpu.py"]),
    ({"SNK_POSE_WAVES": "2", "SNK_TRACK_NO_RECURSE": "1", "SNK_POSE_NO_RECURSE": "1"}, ["test_pose_gpu.py", "test_tracking_chain_gpu.py"]),
])
def test_parity_suites_with_forced_form(env, files):
    r = subprocess.run([sys.executable, "-m", "pytest"] + [str(ROOT / "tests" / f) for f in files] +
                       ["-m", "gpu", "-x", "-q", "-rf", "--tb=short", "-p", "no:cacheprovider"], env=dict(os.environ, **env), capture_output=True,
                       text=True, cwd=str(ROOT), timeout=900)
    assert r.returncode == 0, f"{env}\n--- child stdout (tail) ---\n{r.stdout[-6000:]}\n--- child stderr (tail) ---\n{r.stderr[-1500:]}"
    assert " passed" in r.stdout


def test_stereo_index_of_rows_far_outside_the_image(orc):
    """The counting form of the stereo row index holds 4096 rows; right keypoints that rectification pushed thousands of rows apart run
    the network inside the same launch.  Host call and batched call (B = 3: the sort + 16-lane path), against the oracle."""
    import torch

    from helpers import SEED, make_stereo_case
    from snake_slam_amd.matcher import Preprocess

    rng = np.random.default_rng(SEED + 777)
    left, dl, right, dr, bf, ls = make_stereo_case(rng, 400, 380)
    # spread a third of the pairs far apart (the same shift on both sides keeps them matchable)
    sh = np.where(np.arange(380) % 3 == 0, rng.integers(-3000, 3000, 380) * 1.0, 0.0)
    right["y"] += sh
    k = min(len(left), len(right))
    left["y"][:k] += sh[:k]
    pre = Preprocess(0)
    try:
        n, rp, dp = pre.StereoMatching(left, dl, right, dr, bf, ls, True)
        wn, wrp, wdp = orc.stereo_match(left, dl, right, dr, bf, ls, True)
        assert n == wn and np.array_equal(rp, wrp) and np.array_equal(dp, wdp)
        B, capl, capr = 3, 420, 400
        from oracle.oracle import KP64

        L, R = np.zeros((B, capl), KP64), np.zeros((B, capr), KP64)
        DL, DR = np.zeros((B, capl, 4), np.uint64), np.zeros((B, capr, 4), np.uint64)
        for b in range(B):
            L[b, :400], DL[b, :400], R[b, :380], DR[b, :380] = left, dl, right, dr
        R["y"][1] -= 5000.0  # one frame entirely above the image
        dev = torch.device("cuda:0")
        t = lambda a: torch.from_numpy(a).to(dev)
        nl, nr = t(np.full(B, 400, np.int32)), t(np.full(B, 380, np.int32))
        rpd = torch.full((B, capl), -1000.0, dtype=torch.float32, device=dev)
        dpd = torch.full((B, capl), -1000.0, dtype=torch.float32, device=dev)
        nm = torch.zeros(B, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        pre.match_batch_dev(t(L.view(np.uint8).reshape(B, capl, 24)), t(DL.view(np.int64)), nl, t(R.view(np.uint8).reshape(B, capr, 24)),
                            t(DR.view(np.int64)), nr, bf, ls, True, rpd, dpd, nm)
        pre.sync()
        for b in range(B):
            wn, wrp, wdp = orc.stereo_match(L[b, :400], DL[b, :400], R[b, :380], DR[b, :380], bf, ls, True)
            assert int(nm[b]) == wn and np.array_equal(rpd[b, :400].cpu().numpy(), wrp) and np.array_equal(dpd[b, :400].cpu().numpy(), wdp), b
    finally:
        pre.close()


def test_blur_in_describe_experiment_is_exact_in_the_interior(orc):
    """SNK_ORB_BLUR_IN_DESCRIBE=1 (round 5, the review's item 1c as an experiment switch: no blurred level in HBM, the descriptor wavefront
    blurs each keypoint's raw 43 x 48 window in LDS).  Its claim: the same keypoints, and bit-identical descriptors for every keypoint at
    least 21 pixels inside its level (the level-wide blur reflects at the border, the window blur does not).  Child process: the switch is
    read once; compared with the ORACLE's descriptors."""
    import json

    code = r"""
import json, sys
import numpy as np
from snake_slam_amd import synth
from snake_slam_amd.orb import ORBExtractor
out = []
for seed, (w, h, nf, nl) in enumerate([(752, 480, 1000, 4), (1241, 376, 2000, 7), (320, 240, 300, 3)]):
    img, _ = synth.stereo_frame(20 + seed, w, h)
    ext = ORBExtractor(nf, 1.2, nl, 20, 7)
    k, d = ext.Detect(img)
    ext.close()
    np.save(sys.argv[1] + f"_{seed}.npy", np.concatenate([k["x"][:, None].astype(np.float64), k["y"][:, None], k["octave"][:, None], k["angle"][:, None],
                                                          d.view(np.uint8).reshape(len(k), 32)], 1))
"""
    import tempfile

    from snake_slam_amd import synth

    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run([sys.executable, "-c", code, os.path.join(td, "bid")], env=dict(os.environ, SNK_ORB_BLUR_IN_DESCRIBE="1"), capture_output=True,
                           text=True, cwd=str(ROOT), timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        for seed, (w, h, nf, nl) in enumerate([(752, 480, 1000, 4), (1241, 376, 2000, 7), (320, 240, 300, 3)]):
            got = np.load(os.path.join(td, f"bid_{seed}.npy"))
            img, _ = synth.stereo_frame(20 + seed, w, h)
            wk, wd = orc.orb_detect(orc.orb_params(nf, 1.2, nl, 20, 7), img)
            assert len(got) == len(wk) and np.array_equal(got[:, 0], wk["x"].astype(np.float64)) and np.array_equal(got[:, 1], wk["y"].astype(np.float64))
            assert np.array_equal(got[:, 3], wk["angle"].astype(np.float64))
            scale = np.float32(1.2) ** wk["octave"].astype(np.float32)
            lw, lh = np.rint(w / scale), np.rint(h / scale)
            x, y = np.rint(wk["x"] / scale), np.rint(wk["y"] / scale)
            interior = (x >= 21) & (y >= 21) & (x <= lw - 22) & (y <= lh - 22)
            same = (got[:, 4:].astype(np.uint8) == wd.view(np.uint8).reshape(len(wk), 32)).all(1)
            assert interior.sum() > 0.8 * len(wk) and same[interior].all(), (seed, int(interior.sum()), int(same[interior].sum()))
