"""The per-frame chain of snake_slam_amd/sequence.py restated with the CPU oracle's functions (checker for the GPU
sequence mode; also shows on the CPU that the synthetic sequence is trackable)."""
import numpy as np

from snake_slam_amd import synth

CAM = (458.654, 457.296, 367.215, 248.375, 47.9 * 2.5)
ORB = (1000, 1.2, 4, 20, 7)


def tum_row(stamp, pose):
    q, t = pose[:4], pose[4:]
    R = synth.quat_to_R(q)
    qi = np.array([-q[0], -q[1], -q[2], q[3]])
    if qi[3] < 0:
        qi = -qi
    return np.concatenate([[stamp], -R.T @ t, qi])


def oracle_sequence(orc, frames, width, height, orb=ORB, cam=CAM, threads=4):
    fx, fy, cx, cy, bf = cam
    p = orc.orb_params(*orb)
    ls = (np.float32(orb[1]) ** np.arange(orb[2])).astype(np.float32)
    rect = orc.rectification((1.0, 1.0, 0.0, 0.0))
    bounds = (0.0, 0.0, float(width), float(height))
    prev, rows, poses = None, [], []
    for t, (left, right) in enumerate(frames):
        kl, dl = orc.orb_detect(p, left, threads=threads)
        kr, dr = orc.orb_detect(p, right, threads=threads)
        rl, _ = orc.rectify(rect, kl)
        rr, _ = orc.rectify(rect, kr)
        perm, _, _, _ = orc.feature_grid(rl, bounds)
        g, gd = np.zeros_like(rl), np.zeros_like(dl)
        g[perm], gd[perm] = rl, dl
        _, rp, depth = orc.stereo_match(g, gd, rr, dr, bf, ls, True)
        if prev is None:
            pose = np.array([0, 0, 0, 1.0, 0, 0, 0])
        else:
            # TrackBruteForce (TrackingCoarse.cpp:351-352, 373-377): the current frame is the query set, pairs = (frame feature f,
            # reference feature r) in frame-feature order; kept when the reference feature has a point
            knn = orc.bf_knn2(gd, prev["desc"], threads=threads)
            pairs = np.asarray(orc.bf_filter(knn, 60, 0.8), np.int64).reshape(-1, 2)
            keep = prev["has"][pairs[:, 1]] if len(pairs) else np.zeros(0, bool)
            c, q = pairs[keep, 0], pairs[keep, 1]
            obs = np.zeros(len(c), orc.POSE_OBS)
            obs["x"], obs["y"], obs["depth"] = g["x"][c], g["y"][c], depth[c]
            obs["weight"] = np.sqrt(1.0 / (ls.astype(np.float64)[g["octave"][c]] ** 2))
            if len(c) < 3:
                pose = prev["pose"].copy()
            else:
                pose, _, _ = orc.pose_refine(prev["pose"], orc.Camera(*cam), prev["world"][q], obs)
        has = depth > 0
        z = np.where(has, depth, 1.0).astype(np.float64)
        pc = np.stack([(g["x"] - cx) / fx * z, (g["y"] - cy) / fy * z, z], 1)
        R, tt = synth.quat_to_R(pose[:4]), pose[4:]
        prev = dict(desc=gd, world=(pc - tt) @ R, has=has, pose=pose)
        rows.append(tum_row(float(t), pose))
        poses.append(pose)
    return np.array(rows), poses
