"""One sequence, frame by frame, through the C-ABI seams in the order the reference's threads call them.

BASELINE.json config 5 is "8 sequences, one per GPU": a sequence is causally serial (frame t needs the state of frame
t-1), so a GPU walks its own frame list one frame at a time through the HOST entry points -- the
calls a Snake-SLAM build makes:

    FeatureDetector::Detect (left, right)                      Snake/Preprocess/FeatureDetector.cpp:87-170
    Preprocess: undistortKeypoints, computeFeatureGrid,        Snake/Preprocess/Preprocess.cpp:41-49, 55-77, 122-266
                StereoMatching
    Tracking::TrackBruteForce: matchKnn2 + filterMatches       Snake/Tracking/TrackingCoarse.cpp:350-352
                               + RefinePoseWithMatches         Snake/Tracking/PoseRefinement.cpp:25-79

What stays outside (map, keyframes, local mapping) is replaced by the simplest stand-in that gives the pose refinement
something to do: the previous frame's stereo points are the "map".  The trajectory is kept in the TUM layout the
reference writes (Snake/System/System.cpp:552-563): timestamp, translation and unit quaternion (x y z w) of the INVERSE
pose (camera in the world).
"""
from __future__ import annotations

import numpy as np

from . import synth
from .matcher import BruteForceMatcher, Preprocess, Rectification
from .orb import ORBExtractor
from .tracking import FeatureGrid, PoseRefinement, pose_observations

TUM_COLS = 8  # timestamp tx ty tz qx qy qz qw


def quat_to_R(q):
    return synth.quat_to_R(q)


def inverse_pose_tum(pose) -> np.ndarray:
    """pose (qx qy qz qw tx ty tz, world -> camera) -> [tx ty tz qx qy qz qw] of its inverse (System.cpp:553-555)."""
    q, t = np.asarray(pose[:4], np.float64), np.asarray(pose[4:], np.float64)
    R = quat_to_R(q)
    ti = -R.T @ t
    qi = np.array([-q[0], -q[1], -q[2], q[3]])
    if qi[3] < 0:
        qi = -qi
    return np.concatenate([ti, qi])


def trajectory_block(rows, max_frames: int) -> np.ndarray:
    """the fixed-size padded block a rank contributes to the result gather,
    `{n, [timestamp, tx, ty, tz, qx, qy, qz, qw] x max_frames}` as 1 + 8 * max_frames doubles."""
    rows = np.asarray(rows, np.float64).reshape(-1, TUM_COLS)
    if len(rows) > max_frames:
        raise ValueError(f"{len(rows)} trajectory rows do not fit a block of {max_frames}")
    blk = np.zeros(1 + TUM_COLS * max_frames, np.float64)
    blk[0] = len(rows)
    blk[1:1 + rows.size] = rows.ravel()
    return blk


def trajectory_rows(block) -> np.ndarray:
    """Inverse of trajectory_block: the n valid rows."""
    block = np.asarray(block, np.float64)
    n = int(block[0])
    if n < 0 or 1 + n * TUM_COLS > block.size:
        raise ValueError("corrupt trajectory block")
    return block[1:1 + n * TUM_COLS].reshape(n, TUM_COLS).copy()


def write_tum(path, rows) -> None:
    """The reference's trajectory file (System.cpp:546-563): precision 15, one line per valid frame."""
    with open(path, "w") as f:
        for r in np.asarray(rows, np.float64).reshape(-1, TUM_COLS):
            f.write(" ".join(f"{v:.15g}" for v in r) + "\n")


class SequenceTracker:
    """Per-frame chain of one sequence on one GPU (host entry points, one synchronous call per seam)."""

    def __init__(self, cam, orb=None, device: int = 0, width: int = 752, height: int = 480):
        orb = orb or dict(nfeatures=1000, scale_factor=1.2, n_levels=4, ini_th_fast=20, min_th_fast=7)
        self.cam = tuple(float(v) for v in cam)  # fx fy cx cy bf
        self.ext = ORBExtractor(**orb, device=device)
        self.ext.configure(width, height, 1)
        self.pre = Preprocess(device)
        self.grid = FeatureGrid(device)
        self.bf = BruteForceMatcher(device)
        self.ref = PoseRefinement(device=device)
        self.rect = Rectification.make((1.0, 1.0, 0.0, 0.0))  # synthetic pairs are already rectified
        self.level_scale = (np.float32(orb["scale_factor"]) ** np.arange(orb["n_levels"])).astype(np.float32)
        self.bounds = (0.0, 0.0, float(width), float(height))
        self.prev = None
        self.rows = []
        self.stats = dict(frames=0, keypoints=0, stereo=0, bf_pairs=0, inliers=0)

    def close(self):
        for h in (self.ext, self.pre, self.grid, self.bf, self.ref):
            h.close()

    def process(self, left, right, timestamp: float):
        fx, fy, cx, cy, bf = self.cam
        kl, dl = self.ext.Detect(left)
        kr, dr = self.ext.Detect(right)
        rl, _ = self.pre.rectify(self.rect, kl)
        rr, _ = self.pre.rectify(self.rect, kr)
        perm, _, _, _ = self.grid.create(self.bounds, rl)
        g, gd = np.zeros_like(rl), np.zeros_like(dl)
        g[perm], gd[perm] = rl, dl                      # Preprocess.cpp:254-260: arrays scattered into grid order
        n_st, rp, depth = self.pre.StereoMatching(g, gd, rr, dr, bf, self.level_scale, True)
        inl = 0
        n_pairs = 0
        if self.prev is None:
            pose = np.array([0, 0, 0, 1.0, 0, 0, 0])
        else:
            self.bf.matchKnn2(self.prev["desc"], gd)     # TrackBruteForce: previous (key)frame -> current frame
            n_pairs = self.bf.filterMatches(60, 0.8)
            pairs = np.asarray(self.bf.matches, np.int64).reshape(-1, 2)
            keep = self.prev["has_world"][pairs[:, 0]] if len(pairs) else np.zeros(0, bool)
            q, t = pairs[keep, 0], pairs[keep, 1]
            obs = pose_observations(g[t], depth[t], self.level_scale)
            pose, _, inl = self.ref.RefinePoseWithMatches(self.cam, self.prev["pose"], self.prev["world"][q], obs)
        # this frame's stereo points in the world: the "map" the next frame is tracked against
        has = depth > 0
        z = np.where(has, depth, 1.0).astype(np.float64)
        pc = np.stack([(g["x"] - cx) / fx * z, (g["y"] - cy) / fy * z, z], 1)
        R, tt = quat_to_R(pose[:4]), pose[4:]
        self.prev = dict(desc=gd, world=(pc - tt) @ R, has_world=has, pose=pose)
        self.rows.append(np.concatenate([[float(timestamp)], inverse_pose_tum(pose)]))
        s = self.stats
        s["frames"] += 1
        s["keypoints"] += len(kl) + len(kr)
        s["stereo"] += int(n_st)
        s["bf_pairs"] += int(n_pairs)
        s["inliers"] += int(inl)
        return pose
