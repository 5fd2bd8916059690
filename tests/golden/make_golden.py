#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the CPU oracle (oracle/).

The reference (darglein/Snake-SLAM) ships no golden vectors for this path and its arithmetic lives
in the absent saiga submodule, so these fixtures pin THIS repository's
definition ("snk-orb v1", "snk-ba v1", matcher rules): they freeze the oracle so that neither
it nor the kernels can drift silently.  Inputs are seeded; run from the repo root:

    python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from helpers import SEED, make_stereo_case, rand_desc  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from snake_slam_amd import synth  # noqa: E402

OUT = Path(__file__).resolve().parent


def main():
    orc.build()
    rng = np.random.default_rng(SEED)
    # --- matchers ---
    q, t = rand_desc(rng, 40), rand_desc(rng, 50)
    t[7] = t[3]
    q[0] = t[3]
    knn = orc.bf_knn2(q, t)
    pairs = orc.bf_filter(knn, 120, 0.9)
    left, dl, right, dr, bf, ls = make_stereo_case(rng, 120, 110)
    n, rp, dp = orc.stereo_match(left, dl, right, dr, bf, ls, True)
    np.savez_compressed(OUT / "match_small.npz", q=q, t=t, knn=np.stack([knn[f] for f in ("idx1", "dist1", "idx2", "dist2")], 1),
                        pairs=pairs, filter_args=np.array([120, 0.9]), left=left, dl=dl, right=right, dr=dr, bf=bf, ls=ls,
                        n=n, rp=rp, dp=dp)
    # --- ORB ---
    img, _ = synth.stereo_frame(3, 160, 120, n_rects=40, texture=0.0)  # the image the committed fixture was made from (flat rectangles)
    p = orc.orb_params(200, 1.2, 3, 20, 7)
    kps, desc = orc.orb_detect(p, img)
    np.savez_compressed(OUT / "orb_small.npz", img=img, params=np.array([200, 1.2, 3, 20, 7]), kps=kps, desc=desc)
    # --- rectify ---
    K = (458.654, 457.296, 367.215, 248.375)
    D = (-0.28340811, 0.07395907, 0.0, 0.0, 0.0, 0.0, 0.00019359, 1.76187114e-05)
    a = 0.01
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    Kd = (435.2, 435.2, 367.4, 252.2)
    k = np.zeros(64, orc.KEYPOINT)
    k["x"] = rng.uniform(0, 752, 64).astype(np.float32)
    k["y"] = rng.uniform(0, 480, 64).astype(np.float32)
    k["angle"] = rng.uniform(0, 360, 64).astype(np.float32)
    k["octave"] = rng.integers(0, 4, 64)
    out, norm = orc.rectify(orc.rectification(K, D, R, Kd), k)
    np.savez_compressed(OUT / "rectify_small.npz", K=K, D=D, R=R, Kd=Kd, kps=k, out=out, norm=norm)
    # --- BA ---
    sc, _ = synth.ba_scene(n_kf=6, n_pt=60, obs_per_pt=4, seed=77)
    pose, pt, c0, c1, it = orc.ba_solve(sc, orc.ba_options())
    chi = orc.ba_chi2(sc)
    np.savez_compressed(OUT / "ba_small.npz", **{f"in_{k}": np.asarray(v) for k, v in sc.items()}, pose=pose, pt=pt,
                        cost=np.array([c0, c1]), chi2=chi)
    make_pose()
    make_track()
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size, "bytes")


def make_pose():
    """Pose refinement (snk-pose v1): one frame with outliers, one with a prediction prior."""
    import pose_helpers as PH

    orc.build()
    cam = orc.Camera(*PH.CAM)
    pr = PH.make_problem(501, 120, outlier_frac=0.2)
    pose, outl, inl = orc.pose_refine(pr["pose0"], cam, pr["wps"], pr["obs"])
    pred = PH.perturb(np.random.default_rng(502), pr["pose_gt"], 0.02, 0.05)
    pose_s, outl_s, inl_s = orc.pose_refine(pr["pose0"], cam, pr["wps"], pr["obs"], prediction=pred, w_rot=50.0, w_trans=20.0)
    np.savez_compressed(OUT / "pose_small.npz", cam=np.array(PH.CAM), pose0=pr["pose0"], wps=pr["wps"], obs=pr["obs"],
                        pose=pose, outlier=outl, inliers=inl, prediction=pred, prior=np.array([50.0, 20.0]), pose_s=pose_s,
                        outlier_s=outl_s, inliers_s=inl_s, chi2=orc.pose_chi2(pose, cam, pr["wps"], pr["obs"]))


def _flat_frame(prefix, frame):
    return {f"{prefix}_{k}": np.asarray(v) for k, v in frame.items()}


def make_track():
    """Grid + projection matchers + keyframe-rate matchers: small seeded cases, inputs and oracle outputs."""
    import track_helpers as T

    orc.build()
    out = {}
    rng = np.random.default_rng(SEED + 900)
    frame, cam, pose, ls, world, _ = T.make_tracking_case(orc, rng, n_clutter=150, m_pts=120)
    out.update(_flat_frame("f", frame), cam=np.array(cam), pose=pose, ls=ls)
    pc = T.lm_coarse(orc, world)
    n, idx = orc.match_coarse(frame, cam, pose, pc, 15.0, 75, 0, ls)
    out.update(coarse_pts=pc, coarse_idx=idx, coarse_n=n)
    pf = T.lm_fine(orc, rng, world, pose, ls)
    n, idx, vis, valid = orc.match_fine(frame, cam, pose, pf, 5.0, 0.8, ls)
    out.update(fine_pts=pf, fine_idx=idx, fine_vis=vis, fine_valid=valid, fine_n=n)
    skip = (rng.random(len(world["pos"])) < 0.1).astype(np.uint8)
    n, idx = orc.match_keyframe(frame, cam, pose, world["pos"], world["desc"], skip, 15.0, 100)
    out.update(kf_pos=world["pos"], kf_desc=world["desc"], kf_skip=skip, kf_idx=idx, kf_n=n)
    fp = T.fusion_points(orc, rng, world, pose, ls)
    mask = (rng.random(len(fp)) > 0.2).astype(np.uint8)
    n, idx = orc.match_fuse(frame, cam, pose, fp, mask, 4.0, 2.0, 50, ls)
    out.update(fuse_pts=fp, fuse_mask=mask, fuse_idx=idx, fuse_n=n)
    c = T.make_triangulation_case(orc, rng, m_pts=150, n_clutter=80)
    n, idx = orc.match_triangulation_project(c["grid"], c["pose1"], c["pose2"], c["cam"], c["kps1"], c["np1"], c["desc1"], c["has1"],
                                             c["frame2"], c["np2"], c["E"], 4.0, 50)
    out.update(_flat_frame("t2", c["frame2"]), t_grid=c["grid"], t_pose1=c["pose1"], t_pose2=c["pose2"], t_cam=np.array(c["cam"]),
               t_kps1=c["kps1"], t_np1=c["np1"], t_desc1=c["desc1"], t_has1=c["has1"], t_np2=c["np2"], t_E=c["E"], t_idx=idx, t_n=n)
    b = T.make_bow_case(rng, m_pts=150, n_clutter=80, n_nodes=25)
    n, pairs = orc.match_triangulation_bow(b["cam"], b["E"], b["np1"], b["desc1"], b["has1"], b["bow1"], b["np2"], b["desc2"],
                                           b["has2"], b["bow2"], 4.0, 50)
    nb, ib = orc.match_triangulation_bf(b["cam"], b["E"], b["np1"], b["desc1"], b["has1"], b["np2"], b["desc2"], b["has2"], 50)
    out.update(b_cam=np.array(b["cam"]), b_E=b["E"], b_np1=b["np1"], b_desc1=b["desc1"], b_has1=b["has1"], b_np2=b["np2"],
               b_desc2=b["desc2"], b_has2=b["has2"], b_ids1=b["bow1"][0], b_start1=b["bow1"][1], b_feat1=b["bow1"][2],
               b_ids2=b["bow2"][0], b_start2=b["bow2"][1], b_feat2=b["bow2"][2], b_pairs=pairs, b_n=n, bf_idx=ib, bf_n=nb)
    rframe, rcam, rpose, qs = T.make_relink_case(orc, rng, n_base=120)
    n, action, best = orc.match_relink(rframe, rcam, rpose, qs)
    out.update(_flat_frame("r", rframe), r_cam=np.array(rcam), r_pose=rpose, r_queries=qs, r_action=action, r_best=best, r_n=n)
    np.savez_compressed(OUT / "track_small.npz", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pose":  # regenerate only pose_small.npz
        make_pose()
    elif len(sys.argv) > 1 and sys.argv[1] == "track":  # regenerate only track_small.npz
        make_track()
    else:
        main()
