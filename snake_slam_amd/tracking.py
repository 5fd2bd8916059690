"""Host-side mirror of the tracking matcher interface over the C ABI.

``SnakeORBMatcher`` mirrors ``Snake::SnakeORBMatcher`` (reference Snake/Tracking/SnakeORBMatcher.h:21-30)
with the frame passed as a view (the SoA fields of Snake/Map/Features.h + the taken mask) and
``FeatureGrid.create`` mirrors ``frame.grid.create`` (reference Snake/Preprocess/Preprocess.cpp:246).
``PoseRefinement`` mirrors ``Snake::PoseRefinement`` (reference Snake/Tracking/PoseRefinement.h:22-99).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .matcher import KP64_DTYPE, _Handle, _ptr

LM_COARSE_DTYPE = np.dtype([("pos", "<f8", 3), ("normal", "<f8", 3), ("desc", "<u8", 4), ("octave", "<i4"), ("angle", "<f4")])
LM_FINE_DTYPE = np.dtype([("pos", "<f8", 3), ("normal", "<f8", 3), ("desc", "<u8", 4), ("reference_depth", "<f4"),
                          ("reference_scale_level", "<i4"), ("valid", "u1"), ("pad", "u1", 7)])


class GridBounds(C.Structure):
    _fields_ = [("min_x", C.c_double), ("min_y", C.c_double), ("max_x", C.c_double), ("max_y", C.c_double)]


class FrameView(C.Structure):
    _fields_ = [("n", C.c_int32), ("cols", C.c_int32), ("rows", C.c_int32), ("kps", C.c_void_p), ("desc", C.c_void_p),
                ("right_points", C.c_void_p), ("taken", C.c_void_p), ("cell_start", C.c_void_p), ("bounds", GridBounds)]


class Camera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double)]


def _view(frame):
    if frame is None:
        return None, None
    a = {"kps": np.ascontiguousarray(frame["kps"], KP64_DTYPE), "desc": np.ascontiguousarray(frame["desc"], np.uint64),
         "right_points": np.ascontiguousarray(frame["right_points"], np.float32),
         "taken": np.ascontiguousarray(frame["taken"], np.uint8), "cell_start": np.ascontiguousarray(frame["cell_start"], np.int32)}
    v = FrameView()
    v.n, v.cols, v.rows = len(a["kps"]), int(frame["cols"]), int(frame["rows"])
    for k, arr in a.items():
        setattr(v, k, arr.ctypes.data if arr.size else 0)
    v.bounds = GridBounds(*frame["bounds"])
    return v, a


class FramesDev(C.Structure):
    """snk_frames_dev: a batch of grid-ordered frames resident on the device."""
    _fields_ = [("batch", C.c_int32), ("cap", C.c_int32), ("n", C.c_void_p), ("kps", C.c_void_p), ("desc", C.c_void_p),
                ("right_points", C.c_void_p), ("taken", C.c_void_p), ("cell_start", C.c_void_p), ("bounds", GridBounds)]


def frames_dev(bounds, n, kps, desc, right_points, taken, cell_start) -> FramesDev:
    """Device tensors: n [B] int32, kps [B, cap, 24] uint8 (snk_kp64), desc [B, cap, 4] int64, right_points [B, cap] float32,
    taken [B, cap] uint8, cell_start [B, cols*rows+1] int32 -- the outputs of FeatureGrid.create_batch_dev /
    Preprocess.match_batch_dev.  The tensors must outlive the calls that use the view."""
    f = FramesDev()
    f.batch, f.cap = int(desc.shape[0]), int(desc.shape[1])
    f.n, f.kps, f.desc = n.data_ptr(), kps.data_ptr(), desc.data_ptr()
    f.right_points, f.taken, f.cell_start = right_points.data_ptr(), taken.data_ptr(), cell_start.data_ptr()
    f.bounds = GridBounds(*bounds)
    return f


class FeatureGrid(_Handle):
    def create(self, bounds, undistorted_keypoints):
        """Returns (perm, cell_start, cols, rows); perm[i] = new index of feature i."""
        k = np.ascontiguousarray(undistorted_keypoints, KP64_DTYPE)
        b = GridBounds(*bounds)
        cols, rows = C.c_int(), C.c_int()
        nc = (int(np.ceil((b.max_x - b.min_x) / 20.0)) or 1) * (int(np.ceil((b.max_y - b.min_y) / 20.0)) or 1)
        perm = np.zeros(max(len(k), 1), np.int32)
        cs = np.zeros(nc + 1, np.int32)
        _lib.check(self._lib.snk_feature_grid(self._h, _ptr(k), len(k), C.byref(b), _ptr(perm), _ptr(cs), C.byref(cols),
                                              C.byref(rows)), "snk_feature_grid")
        return perm[: len(k)], cs, cols.value, rows.value


    def create_batch_dev(self, bounds, kps, desc, n, kps_out, desc_out, perm, cell_start):
        """Device tensors: kps [B, cap, 24] uint8 (snk_kp64), desc [B, cap, 4] int64, n [B] int32;
        outputs in grid order + perm [B, cap] int32 + cell_start [B, cols*rows+1] int32."""
        b = GridBounds(*bounds)
        _lib.check(self._lib.snk_feature_grid_batch_dev(self._h, C.byref(b), kps.data_ptr(), desc.data_ptr(), n.data_ptr(),
                                                        desc.shape[1], desc.shape[0], kps_out.data_ptr(), desc_out.data_ptr(),
                                                        perm.data_ptr(), cell_start.data_ptr()), "snk_feature_grid_batch_dev")


class _FrameBinding(_Handle):
    """snk_match_bind_frame / snk_match_bound_taken of a matcher handle (shared by the tracking and the mapping matchers)."""

    def bind_frame(self, frame) -> None:
        """Uploads the frame view once; the Search* methods called with frame=None then use it (1-2 coarse calls and one fine
        call look at the same frame).  frame=None unbinds."""
        v, keep = _view(frame)
        _lib.check(self._lib.snk_match_bind_frame(self._h, C.byref(v) if v is not None else None), "snk_match_bind_frame")

    def bound_taken(self, taken) -> None:
        """New taken mask (mvpMapPoints[i] != nullptr) for the bound frame."""
        t = np.ascontiguousarray(taken, np.uint8)
        _lib.check(self._lib.snk_match_bound_taken(self._h, _ptr(t)), "snk_match_bound_taken")


class SnakeORBMatcher(_FrameBinding):
    def SearchByProjectionFrameFrame2(self, frame, cam, pose, lm_points, th, feature_error, direction, level_scale):
        """Coarse tracking match.  Returns (matches, match_idx[m])."""
        v, keep = _view(frame)
        pts = np.ascontiguousarray(lm_points, LM_COARSE_DTYPE)
        ls = np.ascontiguousarray(level_scale, np.float32)
        pose = np.ascontiguousarray(pose, np.float64)
        out = np.full(max(len(pts), 1), -1, np.int32)
        n = C.c_int(0)
        c = Camera(*cam)
        _lib.check(self._lib.snk_match_project_coarse(self._h, (C.byref(v) if v is not None else None), C.byref(c), _ptr(pose), _ptr(pts), len(pts), float(th),
                                                      int(feature_error), int(direction), _ptr(ls), len(ls), _ptr(out),
                                                      C.byref(n)), "snk_match_project_coarse")
        return n.value, out[: len(pts)]

    def SearchByProjection2(self, frame, cam, pose, lm_points, th, ratio, level_scale):
        """Fine tracking match.  Returns (matches, match_idx[m], visible[m], valid[m])."""
        v, keep = _view(frame)
        pts = np.array(lm_points, LM_FINE_DTYPE, order="C")
        ls = np.ascontiguousarray(level_scale, np.float32)
        pose = np.ascontiguousarray(pose, np.float64)
        out = np.full(max(len(pts), 1), -1, np.int32)
        vis = np.zeros(max(len(pts), 1), np.uint8)
        n = C.c_int(0)
        c = Camera(*cam)
        _lib.check(self._lib.snk_match_project_fine(self._h, (C.byref(v) if v is not None else None), C.byref(c), _ptr(pose), _ptr(pts), len(pts), float(th),
                                                    float(ratio), _ptr(ls), len(ls), _ptr(out), _ptr(vis), C.byref(n)),
                   "snk_match_project_fine")
        return n.value, out[: len(pts)], vis[: len(pts)], pts["valid"].copy()

    # ---- device-resident, batched forms (frames = frames_dev(...); every other argument a device tensor) ----
    def coarse_batch_dev(self, frames: FramesDev, cam, poses, pts, n_pts, th, feature_error, direction, level_scale, match_idx,
                         n_matches):
        """poses [B, 7] float64; pts [B, m_cap, 88] uint8 (snk_lm_coarse); n_pts [B] int32; outputs match_idx [B, m_cap] int32,
        n_matches [B] int32.  Asynchronous on the handle's stream."""
        ls = np.ascontiguousarray(level_scale, np.float32)
        c = Camera(*cam)
        _lib.check(self._lib.snk_match_project_coarse_batch_dev(self._h, C.byref(frames), C.byref(c), poses.data_ptr(), pts.data_ptr(),
                                                                n_pts.data_ptr(), int(pts.shape[1]), float(th), int(feature_error),
                                                                int(direction), _ptr(ls), len(ls), match_idx.data_ptr(),
                                                                n_matches.data_ptr()), "snk_match_project_coarse_batch_dev")

    def fine_batch_dev(self, frames: FramesDev, cam, poses, pts, n_pts, th, ratio, level_scale, match_idx, visible, n_matches,
                       write_valid: bool = True):
        """pts [B, m_cap, 96] uint8 (snk_lm_fine, .valid updated in place); visible [B, m_cap] uint8.  write_valid=False: the records
        are read-only (snk_match_project_fine_batch_ro_dev); the flag the reference leaves in .valid is `visible`."""
        ls = np.ascontiguousarray(level_scale, np.float32)
        c = Camera(*cam)
        name = "snk_match_project_fine_batch_dev" if write_valid else "snk_match_project_fine_batch_ro_dev"
        _lib.check(getattr(self._lib, name)(self._h, C.byref(frames), C.byref(c), poses.data_ptr(), pts.data_ptr(), n_pts.data_ptr(),
                                            int(pts.shape[1]), float(th), float(ratio), _ptr(ls), len(ls), match_idx.data_ptr(),
                                            visible.data_ptr(), n_matches.data_ptr()), name)

    def mark_taken_batch_dev(self, match_idx, n_pts, taken):
        """taken[b, match_idx[b, i]] = 1 for every matched point (the adaptor's mvpMapPoints[idx] = mp, on the device)."""
        _lib.check(self._lib.snk_match_mark_taken_batch_dev(self._h, match_idx.data_ptr(), n_pts.data_ptr(), int(match_idx.shape[1]),
                                                            int(match_idx.shape[0]), taken.data_ptr(), int(taken.shape[1])),
                   "snk_match_mark_taken_batch_dev")

    def SearchByProjectionFrameToKeyframe(self, frame, cam, pose, positions, descriptors, skip, th, feature_error):
        v, keep = _view(frame)
        pos = np.ascontiguousarray(positions, np.float64).reshape(-1, 3)
        desc = np.ascontiguousarray(descriptors, np.uint64).reshape(-1, 4)
        sk = np.ascontiguousarray(skip, np.uint8)
        pose = np.ascontiguousarray(pose, np.float64)
        out = np.full(max(len(pos), 1), -1, np.int32)
        n = C.c_int(0)
        c = Camera(*cam)
        _lib.check(self._lib.snk_match_project_keyframe(self._h, (C.byref(v) if v is not None else None), C.byref(c), _ptr(pose), _ptr(pos), _ptr(desc),
                                                        _ptr(sk), len(pos), float(th), int(feature_error), _ptr(out), C.byref(n)),
                   "snk_match_project_keyframe")
        return n.value, out[: len(pos)]


# ------------------------------------------------------------------ local-mapping matchers -----
FUSION_POINT_DTYPE = np.dtype([("pos", "<f8", 3), ("normal", "<f8", 3), ("desc", "<u8", 4), ("reference_depth", "<f4"),
                               ("reference_scale_level", "<i4"), ("observations", "<i4"), ("id", "<i4")])


class MappingORBMatcher(_FrameBinding):
    """Mirrors ``Snake::MappingORBMatcher`` (reference Snake/LocalMapping/MappingORBMatcher.h:15-45): ``Fuse``
    (LocalMap overload), ``SearchForTriangulation2`` (bag-of-words feature vectors as input),
    ``SearchForTriangulationBF`` and ``SearchForTriangulationProject``."""

    def Fuse(self, frame, cam, pose, points, point_mask, th, obs_factor, feature_th, level_scale):
        """Returns (fusedPoints, fuseCandidates [(feature index, point id)] in point order, best_idx[m])."""
        v, keep = _view(frame)
        pts = np.ascontiguousarray(points, FUSION_POINT_DTYPE)
        ls = np.ascontiguousarray(level_scale, np.float32)
        pose = np.ascontiguousarray(pose, np.float64)
        mask = None if point_mask is None else np.ascontiguousarray(point_mask, np.uint8)
        if mask is not None and len(mask) != len(pts):
            raise ValueError("point_mask size")  # SAIGA_ASSERT, MappingORBMatcher.cpp:369
        out = np.full(max(len(pts), 1), -1, np.int32)
        n = C.c_int(0)
        c = Camera(*cam)
        _lib.check(self._lib.snk_match_fuse(self._h, (C.byref(v) if v is not None else None), C.byref(c), _ptr(pose), _ptr(pts), None if mask is None else _ptr(mask),
                                            len(pts), float(th), float(obs_factor), int(feature_th), _ptr(ls), len(ls), _ptr(out),
                                            C.byref(n)), "snk_match_fuse")
        out = out[: len(pts)]
        cands = [(int(out[i]), int(pts["id"][i])) for i in np.nonzero(out >= 0)[0]]
        return n.value, cands, out

    def SearchForTriangulationProject(self, grid, pose1, pose2, cam, kps1, np1, desc1, has_mp1, frame2, np2, E12,
                                      epipolarDistance, featureDistance):
        """Returns (nmatches, vMatchedPairs [(idx1, idx2)], match_idx2[n1])."""
        v, keep = _view(frame2)
        g = np.ascontiguousarray(grid, np.float64)
        if g.ndim != 2:
            raise ValueError("depth grid must be 2-D (rows x cols, row-major)")
        k1 = np.ascontiguousarray(kps1, KP64_DTYPE)
        n1 = np.ascontiguousarray(np1, np.float64).reshape(-1, 2)
        n2 = np.ascontiguousarray(np2, np.float64).reshape(-1, 2)
        d1 = np.ascontiguousarray(desc1, np.uint64).reshape(-1, 4)
        h1 = np.ascontiguousarray(has_mp1, np.uint8)
        if not (len(k1) == len(n1) == len(d1) == len(h1)) or (v is not None and len(n2) != v.n):
            raise ValueError("array lengths")  # frame2 = None: the bound frame (bind_frame); np2 must hold its feature count
        p1, p2 = np.ascontiguousarray(pose1, np.float64), np.ascontiguousarray(pose2, np.float64)
        E = np.ascontiguousarray(E12, np.float64).reshape(9)
        out = np.full(max(len(k1), 1), -1, np.int32)
        n = C.c_int(0)
        c = Camera(*cam)
        _lib.check(self._lib.snk_match_triangulation_project(self._h, _ptr(g), g.shape[0], g.shape[1], _ptr(p1), _ptr(p2), C.byref(c),
                                                             _ptr(k1), _ptr(n1), _ptr(d1), _ptr(h1), len(k1), (C.byref(v) if v is not None else None), _ptr(n2),
                                                             _ptr(E), float(epipolarDistance), int(featureDistance), _ptr(out),
                                                             C.byref(n)), "snk_match_triangulation_project")
        out = out[: len(k1)]
        return n.value, [(int(i), int(out[i])) for i in np.nonzero(out >= 0)[0]], out


    @staticmethod
    def _kf(np_, desc, has_mp):
        p = np.ascontiguousarray(np_, np.float64).reshape(-1, 2)
        d = np.ascontiguousarray(desc, np.uint64).reshape(-1, 4)
        h = np.ascontiguousarray(has_mp, np.uint8)
        if not (len(p) == len(d) == len(h)):
            raise ValueError("array lengths")
        return p, d, h

    def SearchForTriangulation2(self, cam, E, np1, desc1, has_mp1, bow1, np2, desc2, has_mp2, bow2, epipolarDistance,
                                featureDistance):
        """bow = (node_id ascending, node_start[k + 1], features) = frame->bow_feature_vec flattened.
        Returns (nmatches, vMatchedPairs [(idx1, idx2)] in the reference's emplace order)."""
        p1, d1, h1 = self._kf(np1, desc1, has_mp1)
        p2, d2, h2 = self._kf(np2, desc2, has_mp2)
        b1, keep1 = bow_features(bow1)
        b2, keep2 = bow_features(bow2)
        Ef = np.ascontiguousarray(E, np.float64).reshape(9)
        pairs = np.zeros((max(len(keep1[2]), 1), 2), np.int32)
        n = C.c_int(0)
        c = Camera(*cam)
        _lib.check(self._lib.snk_match_triangulation_bow(self._h, C.byref(c), _ptr(Ef), _ptr(p1), _ptr(d1), _ptr(h1), len(p1),
                                                         C.byref(b1), _ptr(p2), _ptr(d2), _ptr(h2), len(p2), C.byref(b2),
                                                         float(epipolarDistance), int(featureDistance), _ptr(pairs), C.byref(n)),
                   "snk_match_triangulation_bow")
        return n.value, [(int(a), int(b)) for a, b in pairs[: n.value]]

    def SearchForTriangulationBF(self, cam, E12, np1, desc1, has_mp1, np2, desc2, has_mp2, featureDistance):
        """Returns (nmatches, vMatchedPairs [(idx1, idx2)], match_idx2[n1])."""
        p1, d1, h1 = self._kf(np1, desc1, has_mp1)
        p2, d2, h2 = self._kf(np2, desc2, has_mp2)
        Ef = np.ascontiguousarray(E12, np.float64).reshape(9)
        out = np.full(max(len(p1), 1), -1, np.int32)
        n = C.c_int(0)
        c = Camera(*cam)
        _lib.check(self._lib.snk_match_triangulation_bf(self._h, C.byref(c), _ptr(Ef), _ptr(p1), _ptr(d1), _ptr(h1), len(p1), _ptr(p2),
                                                        _ptr(d2), _ptr(h2), len(p2), int(featureDistance), _ptr(out), C.byref(n)),
                   "snk_match_triangulation_bf")
        out = out[: len(p1)]
        return n.value, [(int(i), int(out[i])) for i in np.nonzero(out >= 0)[0]], out


RELINK_QUERY_DTYPE = np.dtype([("pos", "<f8", 3), ("desc", "<u8", 4), ("alt_desc", "<u8", 4), ("feature", "<i4"), ("has_alt", "<i4")])
RELINK_KEEP, RELINK_ERASE, RELINK_MOVE = 0, 1, 2


class DeferredMapper(_Handle):
    """The per-observation search of ``Snake::DeferredMapper::Relink`` (reference
    Snake/Optimizer/DeferredMapper.cpp:39-165).  The map edits stay with the caller (see ``snk_match_relink``)."""

    # reference constants, DeferredMapper.cpp:41-43
    relink_reprojection_error_threshold = 0.8
    relink_outlier_threshold = 2.1
    relink_feature_threshold = 25

    def RelinkSearch(self, frame, cam, pose, queries):
        """queries: RELINK_QUERY_DTYPE, one per feature of the keyframe holding a good map point.
        Returns (n_changed, action[n], best_idx[n])."""
        v, keep = _view(frame)
        q = np.ascontiguousarray(queries, RELINK_QUERY_DTYPE)
        pose = np.ascontiguousarray(pose, np.float64)
        action = np.zeros(max(len(q), 1), np.int32)
        best = np.full(max(len(q), 1), -1, np.int32)
        n = C.c_int(0)
        c = Camera(*cam)
        _lib.check(self._lib.snk_match_relink(self._h, (C.byref(v) if v is not None else None), C.byref(c), _ptr(pose), _ptr(q), len(q),
                                              float(self.relink_reprojection_error_threshold), float(self.relink_outlier_threshold),
                                              int(self.relink_feature_threshold), _ptr(action), _ptr(best), C.byref(n)),
                   "snk_match_relink")
        return n.value, action[: len(q)], best[: len(q)]


class BowFeatures(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("pad", C.c_int32), ("node_id", C.c_void_p), ("node_start", C.c_void_p),
                ("features", C.c_void_p)]


def bow_features(bow):
    """(node_id, node_start, features) -> (snk_bow_features, arrays to keep alive)."""
    ids = np.ascontiguousarray(bow[0], np.uint32)
    start = np.ascontiguousarray(bow[1], np.int32)
    feat = np.ascontiguousarray(bow[2], np.int32)
    if len(start) != len(ids) + 1:
        raise ValueError("node_start must have n_nodes + 1 entries")
    b = BowFeatures(len(ids), 0, ids.ctypes.data if len(ids) else None, start.ctypes.data,
                    feat.ctypes.data if len(feat) else None)
    return b, (ids, start, feat)


# ------------------------------------------------------------------ pose refinement ------------
POSE_OBS_DTYPE = np.dtype([("x", "<f8"), ("y", "<f8"), ("depth", "<f8"), ("weight", "<f8")])

# Snake/System/SnakeGlobal.h:145-146
REPROJECTION_ERROR_THRESHOLD_MONO = 2.1
REPROJECTION_ERROR_THRESHOLD_STEREO = 2.3


class PoseOptions(C.Structure):
    _fields_ = [("th_mono", C.c_double), ("th_stereo", C.c_double), ("outer_iterations", C.c_int32),
                ("inner_iterations", C.c_int32), ("robust_rounds", C.c_int32), ("pad", C.c_int32), ("lambda_", C.c_double)]


class PoseProblem(C.Structure):
    _fields_ = [("n", C.c_int32), ("inliers", C.c_int32), ("wps", C.c_void_p), ("obs", C.c_void_p), ("outlier", C.c_void_p),
                ("pose", C.c_double * 7), ("prediction", C.c_double * 7), ("w_rot", C.c_double), ("w_trans", C.c_double)]


def pose_observations(kps, depth, level_scale):
    """obs[i] of PoseRefinement.h:47-55 from undistorted keypoints: weight = sqrt(InverseSquaredScale(octave))."""
    kps = np.ascontiguousarray(kps, KP64_DTYPE)
    ls = np.asarray(level_scale, np.float64)
    obs = np.zeros(len(kps), POSE_OBS_DTYPE)
    obs["x"], obs["y"] = kps["x"], kps["y"]
    obs["depth"] = np.asarray(depth, np.float64)
    obs["weight"] = np.sqrt(1.0 / (ls[kps["octave"]] ** 2))
    return obs


class PoseRefinement(_Handle):
    """``PoseRefinement(errorFactor)``: thresholds = reprojectionErrorThreshold{Mono,Stereo} * errorFactor
    (reference Snake/Tracking/PoseRefinement.cpp:13-15)."""

    def __init__(self, errorFactor: float = 1.0, device: int = 0, outer: int = 4, inner: int = 10, robust_rounds: int = 3,
                 lam: float = 1e-4, stream: int | None = None):
        super().__init__(device, stream)
        self.options = PoseOptions(REPROJECTION_ERROR_THRESHOLD_MONO * errorFactor, REPROJECTION_ERROR_THRESHOLD_STEREO * errorFactor,
                                   outer, inner, robust_rounds, 0, lam)

    def refine_batch(self, cam, frames):
        """frames: list of dict(pose[7], wps[n,3], obs[n], optional prediction[7], w_rot, w_trans).
        Returns a list of (pose[7], outlier[n] uint8, inliers) — one launch for the whole list."""
        probs = (PoseProblem * max(len(frames), 1))()
        keep = []
        for P, f in zip(probs, frames):
            wps = np.ascontiguousarray(f["wps"], np.float64).reshape(-1, 3)
            obs = np.ascontiguousarray(f["obs"], POSE_OBS_DTYPE)
            if len(wps) != len(obs):
                raise ValueError("wps / obs length mismatch")
            outl = np.zeros(max(len(obs), 1), np.uint8)
            keep.append((wps, obs, outl))
            P.n = len(obs)
            P.wps, P.obs, P.outlier = (wps.ctypes.data if wps.size else 0), (obs.ctypes.data if obs.size else 0), outl.ctypes.data
            P.pose[:] = [float(v) for v in f["pose"]]
            pred = f.get("prediction")
            P.prediction[:] = [float(v) for v in (pred if pred is not None else f["pose"])]
            P.w_rot, P.w_trans = float(f.get("w_rot", 0.0)), float(f.get("w_trans", 0.0))
        c = Camera(*cam)
        _lib.check(self._lib.snk_pose_refine(self._h, C.byref(c), C.byref(self.options), probs, len(frames)), "snk_pose_refine")
        return [(np.array(P.pose[:]), k[2][: P.n].copy(), int(P.inliers)) for P, k in zip(probs, keep)]

    def refine_matches_batch_dev(self, frames: FramesDev, depth, cam, pts, match_idx, n_pts, level_scale, poses, outlier, inliers):
        """Device-resident RefinePoseWithMatches for a batch: frames = frames_dev(...), depth [B, cap] float32, pts [B, m_cap, stride]
        uint8 (snk_lm_coarse / snk_lm_fine), match_idx [B, m_cap] int32 (a batched matcher's output), n_pts [B] int32, poses [B, 7]
        float64 in / out, outlier [B, m_cap] uint8 out, inliers [B] int32 out.  Asynchronous on the handle's stream."""
        ls = np.ascontiguousarray(level_scale, np.float32)
        c = Camera(*cam)
        _lib.check(self._lib.snk_pose_refine_matches_batch_dev(self._h, C.byref(frames), depth.data_ptr(), C.byref(c), C.byref(self.options),
                                                               pts.data_ptr(), int(pts.shape[2]), match_idx.data_ptr(), n_pts.data_ptr(),
                                                               int(pts.shape[1]), _ptr(ls), len(ls), poses.data_ptr(), outlier.data_ptr(),
                                                               inliers.data_ptr()), "snk_pose_refine_matches_batch_dev")

    def refine_frame_batch_dev(self, frames: FramesDev, depth, cam, pts, frame_pt, n_pts, level_scale, poses, outlier, inliers):
        """RefinePoseWithMatches(Frame&) for a batch, fed like the reference (PoseRefinement.cpp:37-57): frame_pt [B, cap] int32 =
        `frame.mvpMapPoints` as indices into pts [B, m_cap, stride] (-1 = nullptr), pairs in FEATURE order; outlier [B, cap] uint8 =
        `frame.mvbOutlier`.  Everything else as refine_matches_batch_dev."""
        ls = np.ascontiguousarray(level_scale, np.float32)
        c = Camera(*cam)
        _lib.check(self._lib.snk_pose_refine_frame_batch_dev(self._h, C.byref(frames), depth.data_ptr(), C.byref(c), C.byref(self.options),
                                                             pts.data_ptr(), int(pts.shape[2]), frame_pt.data_ptr(), n_pts.data_ptr(),
                                                             int(pts.shape[1]), _ptr(ls), len(ls), poses.data_ptr(), outlier.data_ptr(),
                                                             inliers.data_ptr()), "snk_pose_refine_frame_batch_dev")

    def refinePose(self, cam, pose, wps, obs, prediction=None, prediction_weight_rotation=0.0,
                   prediction_weight_translation=0.0):
        """The optimiser call of ``refinePose`` (PoseRefinement.h:62-76): the smooth variant when
        prediction_weight_rotation > 0.  Returns (pose, outlier, inliers)."""
        f = dict(pose=pose, wps=wps, obs=obs)
        if prediction_weight_rotation > 0:
            f.update(prediction=prediction, w_rot=prediction_weight_rotation, w_trans=prediction_weight_translation)
        return self.refine_batch(cam, [f])[0]

    def RefinePoseWithMatches(self, cam, pose, wps, obs):
        """PoseRefinement.cpp:25-79: fewer than 3 correspondences -> 0 inliers, pose untouched."""
        if len(obs) < 3:
            return np.array(pose, np.float64), np.zeros(len(obs), np.uint8), 0
        return self.refine_batch(cam, [dict(pose=pose, wps=wps, obs=obs)])[0]
