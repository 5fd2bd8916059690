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
import ctypes as C

from .tracking import Camera, FeatureGrid, PoseRefinement, pose_observations

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
            # TrackBruteForce (TrackingCoarse.cpp:351-352, 373-377): matchKnn2_omp(frame.descriptors, ref->frame->descriptors) -- the
            # current frame is the QUERY set, m.first a frame feature, m.second a reference feature; pairs come in frame-feature
            # order, which is also the order RefinePoseWithMatches walks frame.mvpMapPoints (PoseRefinement.cpp:37-57)
            self.bf.matchKnn2(gd, self.prev["desc"])
            n_pairs = self.bf.filterMatches(60, 0.8)
            pairs = np.asarray(self.bf.matches, np.int64).reshape(-1, 2)
            keep = self.prev["has_world"][pairs[:, 1]] if len(pairs) else np.zeros(0, bool)
            f, r = pairs[keep, 0], pairs[keep, 1]
            obs = pose_observations(g[f], depth[f], self.level_scale)
            pose, _, inl = self.ref.RefinePoseWithMatches(self.cam, self.prev["pose"], self.prev["world"][r], obs)
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


class MultiSequenceTracker:
    """S sequences on ONE GPU in lockstep, device resident (BASELINE.json config 5, "sequences batched"): frame t of every
    sequence goes through the batched entry points as one batch -- Detect of the 2 S images, rectify, feature grid,
    StereoMatching, matchKnn2 + filterMatches against the previous frame's descriptors, the kept matches as (world point,
    observation) pairs (`snk_track_bf_matches_batch_dev`), RefinePoseWithMatches (`snk_pose_refine_frame_batch_dev`, the previous
    frame's stereo points as the "map"), this frame's stereo points into the world (`snk_track_backproject_batch_dev`) -- on one
    stream, without a host round trip: what crosses PCIe per step is the 2 S images going in and, at the end of the run, the poses
    and four counters coming out.  Same chain and same arithmetic per sequence as `SequenceTracker` (which makes one synchronous
    host call per seam and keeps a GPU ~95 % idle); a sequence is still causally serial, the parallelism is across sequences.

    Image uploads are double buffered on a copy stream (pinned host buffers), so step t + 1's upload runs beside step t's kernels."""

    def __init__(self, cam, n_sequences: int, max_frames: int, orb=None, device: int = 0, width: int = 752, height: int = 480):
        import torch

        from . import _lib
        from .tracking import frames_dev

        orb = orb or dict(nfeatures=1000, scale_factor=1.2, n_levels=4, ini_th_fast=20, min_th_fast=7)
        self.torch, self._lib, self._frames_dev = torch, _lib, frames_dev
        self.cam = tuple(float(v) for v in cam)
        S = self.S = int(n_sequences)
        self.W, self.H, self.T = int(width), int(height), int(max_frames)
        self.dev = torch.device("cuda", device)
        self.stream = torch.cuda.Stream(device=self.dev)
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        sh = self.stream.cuda_stream
        self.ext = ORBExtractor(**orb, device=device, stream=sh)
        cap = self.cap = self.ext.configure(width, height, 2 * S)
        self.pre = Preprocess(device, sh)
        self.grid = FeatureGrid(device, sh)
        self.bf = BruteForceMatcher(device, sh)
        self.ref = PoseRefinement(device=device, stream=sh)
        self.rect = Rectification.make((1.0, 1.0, 0.0, 0.0))
        self.level_scale = (np.float32(orb["scale_factor"]) ** np.arange(orb["n_levels"])).astype(np.float32)
        self.bounds = (0.0, 0.0, float(width), float(height))
        n_cells = int(np.ceil(width / 20.0)) * int(np.ceil(height / 20.0))
        pitch = self.pitch = (width + 63) & ~63
        z = lambda *shape, dtype: torch.zeros(shape, dtype=dtype, device=self.dev)  # noqa: E731
        with torch.cuda.device(self.dev):
            self.host = [torch.zeros((2 * S, height, pitch), dtype=torch.uint8).pin_memory() for _ in range(2)]
            self.images = [z(2 * S, height, pitch, dtype=torch.uint8) for _ in range(2)]
            self.ev_up = [torch.cuda.Event() for _ in range(2)]    # upload of buffer k done (copy stream)
            self.ev_use = [torch.cuda.Event() for _ in range(2)]   # extraction has finished reading buffer k (compute stream)
            self.kps, self.desc, self.nkp = z(2 * S, cap, 24, dtype=torch.uint8), z(2 * S, cap, 4, dtype=torch.int64), z(2 * S, dtype=torch.int32)
            self.kp64 = z(2 * S, cap, 24, dtype=torch.uint8)
            self.kp64_g, self.desc_g = z(S, cap, 24, dtype=torch.uint8), z(S, cap, 4, dtype=torch.int64)
            self.perm, self.cell_start = z(S, cap, dtype=torch.int32), z(S, n_cells + 1, dtype=torch.int32)
            self.right_points, self.depth = z(S, cap, dtype=torch.float32), z(S, cap, dtype=torch.float32)
            self.n_stereo = z(S, dtype=torch.int32)
            self.taken = z(S, cap, dtype=torch.uint8)
            self.knn, self.pairs, self.n_pairs = z(S, cap, 4, dtype=torch.int32), z(S, cap, 2, dtype=torch.int32), z(S, dtype=torch.int32)
            self.match_idx, self.outlier, self.inliers = z(S, cap, dtype=torch.int32), z(S, cap, dtype=torch.uint8), z(S, dtype=torch.int32)
            self.prev_desc, self.prev_n = z(S, cap, 4, dtype=torch.int64), z(S, dtype=torch.int32)
            self.prev_world, self.prev_has = z(S, cap, 3, dtype=torch.float64), z(S, cap, dtype=torch.uint8)
            self.world, self.has = z(S, cap, 3, dtype=torch.float64), z(S, cap, dtype=torch.uint8)
            ident = np.zeros((S, 7))
            ident[:, 3] = 1.0
            self.poses = torch.from_numpy(ident).to(self.dev)
            self.track = z(self.T, S, 7, dtype=torch.float64)          # the pose of every sequence after every step
            self.counters = z(4, dtype=torch.int64)                    # keypoints, stereo matches, kept BF pairs, inliers
        self.stamps = []
        self.t = 0
        torch.cuda.synchronize(self.dev)

    def close(self):
        self.torch.cuda.synchronize(self.dev)
        for h in (self.ext, self.pre, self.grid, self.bf, self.ref):
            h.close()

    def stage(self, lefts, rights, out=None):
        """The 2 S images of one step in the layout the upload wants ([2 S, H, pitch] uint8, left images first), in PINNED host
        memory -- where a camera driver / file reader would put them.  Returns the tensor (a new one unless `out` is given)."""
        torch = self.torch
        if out is None:
            with torch.cuda.device(self.dev):
                out = torch.zeros((2 * self.S, self.H, self.pitch), dtype=torch.uint8).pin_memory()
        hb = out.numpy()
        for s in range(self.S):
            hb[s, :, : self.W] = lefts[s]
            hb[self.S + s, :, : self.W] = rights[s]
        return out

    def _upload(self, k, staged):
        torch = self.torch
        self.ev_use[k].synchronize()  # the extraction that read device buffer k two steps ago is done
        with torch.cuda.stream(self.copy_stream):
            self.images[k].copy_(staged, non_blocking=True)
            self.ev_up[k].record(self.copy_stream)

    def process(self, lefts, rights, timestamp: float):
        """lefts / rights: S images each (uint8 [H, W]), frame t of every sequence.  Asynchronous: returns when the step is enqueued
        (the images are copied into one of the tracker's two pinned staging buffers first)."""
        k = self.t & 1
        self.ev_up[k].synchronize()  # the upload that read host buffer k two steps ago is done
        self.process_staged(self.stage(lefts, rights, out=self.host[k]), timestamp)

    def process_staged(self, staged, timestamp: float):
        """One step from a tensor made by `stage` (pinned, [2 S, H, pitch]): no host-side copy, the upload is one asynchronous DMA.
        The tensor must stay untouched until the step has run (synchronise, or keep one tensor per step)."""
        torch, S, cap, lib = self.torch, self.S, self.cap, self._lib
        if self.t >= self.T:
            raise ValueError("more frames than max_frames")
        k = self.t & 1
        self._upload(k, staged)
        fx, fy, cx, cy, bf = self.cam
        st = self.stream
        with torch.cuda.stream(st):
            st.wait_event(self.ev_up[k])
            self.ext.detect_batch_dev(self.images[k], self.kps, self.desc, self.nkp)
            self.ev_use[k].record(st)
            self.pre.rectify_batch_dev(self.rect, self.kps, self.nkp, self.kp64)
            self.grid.create_batch_dev(self.bounds, self.kp64[:S], self.desc[:S], self.nkp[:S], self.kp64_g, self.desc_g, self.perm, self.cell_start)
            self.right_points.fill_(-1000.0)  # Frame::allocateTmp (Frame.cpp:25-26)
            self.depth.fill_(-1000.0)
            self.pre.match_batch_dev(self.kp64_g, self.desc_g, self.nkp[:S], self.kp64[S:], self.desc[S:], self.nkp[S:], bf,
                                     self.level_scale, True, self.right_points, self.depth, self.n_stereo)
            fd = self._frames_dev(self.bounds, self.nkp[:S], self.kp64_g, self.desc_g, self.right_points, self.taken, self.cell_start)
            if self.t > 0:
                # current frame = query, previous frame = train (TrackingCoarse.cpp:351); frame_pt = mvpMapPoints as indices
                self.bf.knn2_batch_dev(self.desc_g, self.nkp[:S], self.prev_desc, self.prev_n, self.knn)
                self.bf.filter_batch_dev(self.knn, self.nkp[:S], 60, 0.8, self.pairs, self.n_pairs)
                lib.check(lib.load().snk_track_bf_matches_batch_dev(self.ref._h, self.pairs.data_ptr(), self.n_pairs.data_ptr(),
                                                                    self.prev_has.data_ptr(), cap, S, self.match_idx.data_ptr()),
                          "snk_track_bf_matches_batch_dev")
                self.ref.refine_frame_batch_dev(fd, self.depth, self.cam, self.prev_world.view(torch.uint8).view(S, cap, 24), self.match_idx,
                                                self.prev_n, self.level_scale, self.poses, self.outlier, self.inliers)
                self.counters[2] += self.n_pairs.sum()
                self.counters[3] += self.inliers.sum()
            c = Camera(*self.cam)
            lib.check(lib.load().snk_track_backproject_batch_dev(self.ref._h, C.byref(fd), self.depth.data_ptr(), C.byref(c),
                                                                 self.poses.data_ptr(), self.world.data_ptr(), self.has.data_ptr()),
                      "snk_track_backproject_batch_dev")
            self.prev_desc.copy_(self.desc_g)
            self.prev_n.copy_(self.nkp[:S])
            self.prev_world, self.world = self.world, self.prev_world
            self.prev_has, self.has = self.has, self.prev_has
            self.track[self.t].copy_(self.poses)
            self.counters[0] += self.nkp.sum()
            self.counters[1] += self.n_stereo.sum()
        self.stamps.append(float(timestamp))
        self.t += 1

    def results(self):
        """Synchronises and returns (rows per sequence: list of [t, 8] TUM arrays, stats dict)."""
        self.torch.cuda.synchronize(self.dev)
        tr = self.track[: self.t].cpu().numpy()
        rows = [np.array([np.concatenate([[self.stamps[t]], inverse_pose_tum(tr[t, s])]) for t in range(self.t)]) for s in range(self.S)]
        c = self.counters.cpu().numpy()
        return rows, dict(frames=self.t * self.S, keypoints=int(c[0]), stereo=int(c[1]), bf_pairs=int(c[2]), inliers=int(c[3]))
