"""Host-side mirror of the reference's matcher interfaces over the C ABI.

* ``BruteForceMatcher`` mirrors ``Saiga::BruteForceMatcher<DescriptorORB>`` as Snake uses it
  (reference Snake/Tracking/TrackingCoarse.cpp:350-352,373-387): ``matchKnn2`` /
  ``matchKnn2_omp`` then ``filterMatches(threshold, ratio)`` and the public ``matches``.
* ``StereoMatcher.StereoMatching`` mirrors ``Snake::Preprocess::StereoMatching``
  (reference Snake/Preprocess/Preprocess.cpp:122-242).

Everything runs in the HIP library; numpy only carries host buffers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

KP64_DTYPE = np.dtype([("x", "<f8"), ("y", "<f8"), ("angle", "<f4"), ("octave", "<i4")])
KNN2_DTYPE = np.dtype([("idx1", "<i4"), ("dist1", "<i4"), ("idx2", "<i4"), ("dist2", "<i4")])


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None and a.size else C.c_void_p(0)


def _as_desc(d) -> np.ndarray:
    d = np.ascontiguousarray(d)
    if d.dtype == np.uint8:
        if d.ndim != 2 or d.shape[1] != 32:
            raise ValueError("uint8 descriptors must be [N, 32]")
        d = d.view("<u8")
    if d.dtype != np.uint64 or d.ndim != 2 or d.shape[1] != 4:
        raise ValueError("descriptors must be [N, 4] uint64 (or [N, 32] uint8)")
    return d


class _Handle:
    def __init__(self, device: int = 0, stream: int | None = None):
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.snk_matcher_create(device, C.c_void_p(stream or 0), C.byref(h)), "snk_matcher_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.snk_matcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _lib.check(self._lib.snk_matcher_sync(self._h), "snk_matcher_sync")


class BruteForceMatcher(_Handle):
    """matchKnn2 + filterMatches with the reference's call shape."""

    def __init__(self, device: int = 0, stream: int | None = None):
        super().__init__(device, stream)
        self.knn = np.zeros(0, KNN2_DTYPE)
        self.matches = np.zeros((0, 2), np.int32)

    def matchKnn2(self, desc1, desc2) -> None:
        q, t = _as_desc(desc1), _as_desc(desc2)
        out = np.zeros(q.shape[0], KNN2_DTYPE)
        _lib.check(self._lib.snk_bf_knn2(self._h, _ptr(q), q.shape[0], _ptr(t), t.shape[0], _ptr(out)), "snk_bf_knn2")
        self.knn = out

    def matchKnn2_omp(self, desc1, desc2, threads: int = 1) -> None:  # thread count is meaningless on the GPU
        self.matchKnn2(desc1, desc2)

    def filterMatches(self, threshold: int, ratio: float) -> int:
        nq = self.knn.shape[0]
        pairs = np.zeros((max(nq, 1), 2), np.int32)
        n = C.c_int(0)
        _lib.check(
            self._lib.snk_bf_filter(self._h, _ptr(self.knn), nq, int(threshold), float(ratio), _ptr(pairs), C.byref(n)),
            "snk_bf_filter",
        )
        self.matches = pairs[: n.value].copy()
        return n.value

    # ---- device-resident batched forms (torch tensors on the handle's device) ----
    def knn2_batch_dev(self, query, nq, train, nt, out):
        """query [B, capq, 4] int64/uint64 cuda tensor, nq [B] int32; train likewise; out [B, capq, 4] int32."""
        B, capq = query.shape[0], query.shape[1]
        _lib.check(
            self._lib.snk_bf_knn2_batch_dev(self._h, query.data_ptr(), nq.data_ptr(), capq, train.data_ptr(),
                                            nt.data_ptr(), train.shape[1], B, out.data_ptr()),
            "snk_bf_knn2_batch_dev",
        )

    def filter_batch_dev(self, knn, nq, threshold, ratio, pairs, n_pairs):
        B, capq = knn.shape[0], knn.shape[1]
        _lib.check(
            self._lib.snk_bf_filter_batch_dev(self._h, knn.data_ptr(), nq.data_ptr(), capq, B, int(threshold),
                                              float(ratio), pairs.data_ptr(), n_pairs.data_ptr()),
            "snk_bf_filter_batch_dev",
        )


class StereoMatcher(_Handle):
    def StereoMatching(self, left_kps, desc_left, right_kps, desc_right, bf: float, level_scale,
                       relaxed: bool = True, right_points=None, depth=None):
        """Returns (num_matches, right_points, depth).  left_kps/right_kps: KP64_DTYPE arrays of
        RECTIFIED keypoints.  right_points/depth default to the -1000 fill of Frame::allocateTmp."""
        lk = np.ascontiguousarray(left_kps, dtype=KP64_DTYPE)
        rk = np.ascontiguousarray(right_kps, dtype=KP64_DTYPE)
        dl, dr = _as_desc(desc_left), _as_desc(desc_right)
        nl, nr = lk.shape[0], rk.shape[0]
        if dl.shape[0] != nl or dr.shape[0] != nr:
            raise ValueError("keypoint / descriptor count mismatch")
        rp = np.full(nl, -1000.0, np.float32) if right_points is None else np.ascontiguousarray(right_points, np.float32)
        dp = np.full(nl, -1000.0, np.float32) if depth is None else np.ascontiguousarray(depth, np.float32)
        ls = np.ascontiguousarray(level_scale, np.float32)
        n = C.c_int(0)
        _lib.check(
            self._lib.snk_stereo_match(self._h, _ptr(lk), _ptr(dl), nl, _ptr(rk), _ptr(dr), nr, float(bf), _ptr(ls),
                                       ls.shape[0], int(bool(relaxed)), _ptr(rp), _ptr(dp), C.byref(n)),
            "snk_stereo_match",
        )
        return n.value, rp, dp

    def match_batch_dev(self, left, desc_left, nl, right, desc_right, nr, bf, level_scale, relaxed, right_points,
                        depth, n_matches):
        """Device tensors: left [B, capl, 24 bytes] (uint8 view of snk_kp64), desc [B, cap, 4] int64, counts int32."""
        ls = np.ascontiguousarray(level_scale, np.float32)
        B = desc_left.shape[0]
        _lib.check(
            self._lib.snk_stereo_match_batch_dev(self._h, left.data_ptr(), desc_left.data_ptr(), nl.data_ptr(),
                                                 desc_left.shape[1], right.data_ptr(), desc_right.data_ptr(),
                                                 nr.data_ptr(), desc_right.shape[1], B, float(bf), _ptr(ls),
                                                 ls.shape[0], int(bool(relaxed)), right_points.data_ptr(),
                                                 depth.data_ptr(), n_matches.data_ptr()),
            "snk_stereo_match_batch_dev",
        )


class Rectification(C.Structure):
    """Mirror of the Saiga::Rectification fields Snake uses (reference Snake/System/SnakeGlobal.h:107-108)."""
    _fields_ = [("K_src", C.c_double * 4), ("D_src", C.c_double * 8), ("R", C.c_double * 9), ("K_dst", C.c_double * 4),
                ("bf", C.c_double)]

    @classmethod
    def make(cls, K_src, D_src=None, R=None, K_dst=None, bf=0.0):
        r = cls()
        r.K_src[:] = list(K_src)
        r.D_src[:] = list(D_src) if D_src is not None else [0.0] * 8
        r.R[:] = list(np.asarray(R, np.float64).reshape(9)) if R is not None else [1, 0, 0, 0, 1, 0, 0, 0, 1]
        r.K_dst[:] = list(K_dst) if K_dst is not None else list(K_src)
        r.bf = bf
        return r


class RgbdModel(C.Structure):
    """K, rgbd_intrinsics.depthModel.dis / .K and rgbd_intrinsics.bf as ComputeStereoFromRGBD reads them (Preprocess.cpp:93-95,108)."""
    _fields_ = [("K", C.c_double * 4), ("D_depth", C.c_double * 8), ("K_depth", C.c_double * 4), ("bf", C.c_double)]

    @classmethod
    def make(cls, K, D_depth=None, K_depth=None, bf=40.0):
        r = cls()
        r.K[:] = list(K)
        d = list(D_depth) if D_depth is not None else []
        r.D_depth[:] = d + [0.0] * (8 - len(d))
        r.K_depth[:] = list(K_depth) if K_depth is not None else list(K)
        r.bf = bf
        return r


class Preprocess(StereoMatcher):
    """undistortKeypoints / Rectification::Forward + StereoMatching on one handle (the reference's
    "Preprocess" thread, Snake/Preprocess/Preprocess.cpp:35-53)."""

    def rectify(self, rect: Rectification, kps, want_normalized: bool = True):
        from .orb import KEYPOINT_DTYPE

        k = np.ascontiguousarray(kps, dtype=KEYPOINT_DTYPE)
        out = np.zeros(k.shape[0], KP64_DTYPE)
        norm = np.zeros((k.shape[0], 2), np.float64) if want_normalized else None
        _lib.check(
            self._lib.snk_rectify(self._h, C.byref(rect), _ptr(k), k.shape[0], _ptr(out),
                                  _ptr(norm) if norm is not None else C.c_void_p(0)),
            "snk_rectify",
        )
        return out, norm

    def ComputeStereoFromRGBD(self, model: "RgbdModel", undistorted, depth_image):
        """Preprocess::ComputeStereoFromRGBD (Preprocess.cpp:79-120).  Returns (matches, right_points, depth); raises SnakeHipError where
        the reference aborts (a keypoint outside the depth image, a depth outside [0, 20))."""
        u = np.ascontiguousarray(undistorted, KP64_DTYPE)
        img = np.ascontiguousarray(depth_image, np.float32)
        rp, dp = np.zeros(max(len(u), 1), np.float32), np.zeros(max(len(u), 1), np.float32)
        n = C.c_int(0)
        _lib.check(self._lib.snk_rgbd_stereo(self._h, C.byref(model), _ptr(u), len(u), _ptr(img), img.shape[1], img.shape[0], img.shape[1],
                                             _ptr(rp), _ptr(dp), C.byref(n)), "snk_rgbd_stereo")
        return n.value, rp[: len(u)], dp[: len(u)]

    def rgbd_batch_dev(self, model: "RgbdModel", undistorted, n, depth_images, right_points, depth, n_matches, status):
        """undistorted [B, cap, 24] uint8 (snk_kp64), n [B] int32, depth_images [B, H, W] float32; outputs right_points / depth [B, cap]
        float32, n_matches / status [B] int32 (status 0x7FFFFFFF = fine).  Asynchronous on the handle's stream."""
        B, cap = undistorted.shape[0], undistorted.shape[1]
        _lib.check(self._lib.snk_rgbd_stereo_batch_dev(self._h, C.byref(model), undistorted.data_ptr(), n.data_ptr(), cap, B, depth_images.data_ptr(),
                                                       int(depth_images.shape[2]), int(depth_images.shape[1]), int(depth_images.stride(1)),
                                                       int(depth_images.stride(0)), right_points.data_ptr(), depth.data_ptr(), n_matches.data_ptr(),
                                                       status.data_ptr()), "snk_rgbd_stereo_batch_dev")

    def rectify_batch_dev(self, rect: Rectification, kps, n, out, normalized=None):
        B, cap = kps.shape[0], kps.shape[1]
        _lib.check(
            self._lib.snk_rectify_batch_dev(self._h, C.byref(rect), kps.data_ptr(), n.data_ptr(), cap, B, out.data_ptr(),
                                            normalized.data_ptr() if normalized is not None else 0),
            "snk_rectify_batch_dev",
        )
