"""Host-side mirror of the reference's per-frame front-end over the C ABI: ``Frontend(...).Process(left, right)`` is
``FeatureDetector::Detect`` for both images (reference Snake/Preprocess/FeatureDetector.cpp:116-156) followed by
``Preprocess::Process`` (allocateTmp, undistortKeypoints, computeFeatureGrid, StereoMatching; Snake/Preprocess/Preprocess.cpp:35-53)
in ONE call and ONE synchronisation (``snk_frontend_process``).  The returned dict carries the members of ``Snake::Frame`` those
two modules fill, left arrays in feature-grid order, right arrays in extractor order."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .matcher import KP64_DTYPE, Rectification
from .orb import KEYPOINT_DTYPE, OrbParams
from .tracking import GridBounds


class FrontendParams(C.Structure):
    _fields_ = [("orb", OrbParams), ("rect_left", Rectification), ("rect_right", Rectification), ("bounds", GridBounds), ("bf", C.c_double),
                ("relaxed_stereo", C.c_int32), ("stereo", C.c_int32)]


class FrontendFrame(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("n", C.c_int32), ("n_right", C.c_int32), ("n_stereo", C.c_int32), ("cols", C.c_int32),
                ("rows", C.c_int32), ("keypoints", C.c_void_p), ("descriptors", C.c_void_p), ("undistorted_keypoints", C.c_void_p),
                ("normalized_points", C.c_void_p), ("permutation", C.c_void_p), ("cell_start", C.c_void_p), ("right_points", C.c_void_p),
                ("depth", C.c_void_p), ("keypoints_right", C.c_void_p), ("descriptors_right", C.c_void_p)]


class Frontend:
    def __init__(self, orb=(1000, 1.2, 4, 20, 7), rect_left: Rectification | None = None, rect_right: Rectification | None = None,
                 bounds=(0.0, 0.0, 752.0, 480.0), bf: float = 47.9, relaxed_stereo: bool = True, stereo: bool = True, device: int = 0):
        self._lib = _lib.load()
        rl = rect_left or Rectification.make((1.0, 1.0, 0.0, 0.0))
        p = FrontendParams()
        p.orb = OrbParams(int(orb[0]), float(orb[1]), int(orb[2]), int(orb[3]), int(orb[4]), 0)
        p.rect_left, p.rect_right = rl, rect_right or rl
        p.bounds = GridBounds(*[float(v) for v in bounds])
        p.bf, p.relaxed_stereo, p.stereo = float(bf), int(bool(relaxed_stereo)), int(bool(stereo))
        self.params = p
        h = C.c_void_p()
        _lib.check(self._lib.snk_frontend_create(C.byref(p), device, C.byref(h)), "snk_frontend_create")
        self._h = h
        self.level_scale = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(int(orb[2]) - 1, np.float32(orb[1]))]).astype(np.float32),
                                      dtype=np.float32)
        self._arrays = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.snk_frontend_destroy(self._h)
            self._h = None
            for p in getattr(self, "_pinned", []):
                self._lib.snk_pinned_free(p)
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _alloc(self, w, h):
        cap = C.c_int(0)
        _lib.check(self._lib.snk_frontend_max_keypoints(self._h, w, h, C.byref(cap)), "snk_frontend_max_keypoints")
        cols, rows = C.c_int(0), C.c_int(0)
        _lib.check(self._lib.snk_frontend_grid_dims(self._h, C.byref(cols), C.byref(rows)), "snk_frontend_grid_dims")
        n = cap.value
        a = dict(keypoints=np.zeros(n, KEYPOINT_DTYPE), descriptors=np.zeros((n, 4), np.uint64), undistorted_keypoints=np.zeros(n, KP64_DTYPE),
                 normalized_points=np.zeros((n, 2), np.float64), permutation=np.zeros(n, np.int32),
                 cell_start=np.zeros(cols.value * rows.value + 1, np.int32), right_points=np.zeros(n, np.float32), depth=np.zeros(n, np.float32),
                 keypoints_right=np.zeros(n, KEYPOINT_DTYPE), descriptors_right=np.zeros((n, 4), np.uint64))
        fr = FrontendFrame()
        fr.capacity = n
        for k, v in a.items():
            setattr(fr, k, v.ctypes.data)
        self._arrays, self._frame, self._size = a, fr, (w, h)

    def set_depth(self, depth: int) -> None:
        """Frames that may be in flight between Submit and Collect (1..8, default 3)."""
        _lib.check(self._lib.snk_frontend_set_depth(self._h, int(depth)), "snk_frontend_set_depth")

    def Submit(self, left: np.ndarray, right: np.ndarray | None = None) -> None:
        """snk_frontend_submit: enqueue one frame and return (blocks only while `depth` frames are uncollected) -- the producer side
        of the reference's FeatureDetection -> Preprocess slot (Snake/Preprocess/FeatureDetector.h:39)."""
        left = np.ascontiguousarray(left, np.uint8)
        h, w = left.shape
        if right is not None:
            right = np.ascontiguousarray(right, np.uint8)
            assert right.shape == left.shape
        if self._arrays is None or self._size != (w, h):
            self._alloc(w, h)
        _lib.check(self._lib.snk_frontend_submit(self._h, left.ctypes.data, w, right.ctypes.data if right is not None else None, w, w, h),
                   "snk_frontend_submit")

    def pinned_images(self, width: int, height: int, count: int = 2, pitch: int | None = None) -> np.ndarray:
        """`count` images of page-locked host memory (snk_pinned_alloc) as one (count, height, pitch) uint8 array -- what Snake's Input
        thread would allocate its image buffers from (Snake/Preprocess/Input.h:48).  Freed when the Frontend is closed."""
        pitch = int(pitch or width)
        p = C.c_void_p()
        _lib.check(self._lib.snk_pinned_alloc(count * height * pitch, C.byref(p)), "snk_pinned_alloc")
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        buf = (C.c_uint8 * (count * height * pitch)).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(count, height, pitch)

    def SubmitPinned(self, left: np.ndarray, right: np.ndarray | None = None) -> None:
        """snk_frontend_submit_pinned: like Submit, but the upload reads the caller's (page-locked) arrays directly -- they must stay
        untouched until the frame has been collected.  `left` / `right`: 2-D uint8 views with contiguous rows (pitch = strides[0])."""
        h, w = left.shape
        assert left.dtype == np.uint8 and left.strides[1] == 1 and (right is None or (right.shape == left.shape and right.strides[1] == 1))
        if self._arrays is None or self._size != (w, h):
            self._alloc(w, h)
        _lib.check(self._lib.snk_frontend_submit_pinned(self._h, left.ctypes.data, left.strides[0], right.ctypes.data if right is not None else None,
                                                        right.strides[0] if right is not None else 0, w, h), "snk_frontend_submit_pinned")

    def Collect(self, timeout_ms: int = -1, copy: bool = True) -> dict:
        """snk_frontend_collect: the oldest submitted frame, as Process returns it.  copy=False hands out views of the handle's own
        arrays (valid until the next Collect / Process): what a consumer that reads the frame before asking for the next one needs."""
        fr, a = self._frame, self._arrays
        _lib.check(self._lib.snk_frontend_collect(self._h, C.byref(fr), int(timeout_ms)), "snk_frontend_collect")
        return self._result(fr, a, copy)

    def in_flight(self) -> int:
        n = C.c_int(0)
        _lib.check(self._lib.snk_frontend_in_flight(self._h, C.byref(n)), "snk_frontend_in_flight")
        return n.value

    def _result(self, fr, a, copy: bool = True) -> dict:
        n, nr = fr.n, fr.n_right
        out = {k: (v[:nr] if k.endswith("_right") else (v if k == "cell_start" else v[:n])) for k, v in a.items()}
        if copy:
            out = {k: v.copy() for k, v in out.items()}
        out.update(N=n, n_right=nr, n_stereo=fr.n_stereo, cols=fr.cols, rows=fr.rows)
        return out

    def Process(self, left: np.ndarray, right: np.ndarray | None = None) -> dict:
        left = np.ascontiguousarray(left, np.uint8)
        h, w = left.shape
        if right is not None:
            right = np.ascontiguousarray(right, np.uint8)
            assert right.shape == left.shape
        if self._arrays is None or self._size != (w, h):
            self._alloc(w, h)
        fr, a = self._frame, self._arrays
        _lib.check(self._lib.snk_frontend_process(self._h, left.ctypes.data, w, right.ctypes.data if right is not None else None, w, w, h,
                                                  C.byref(fr)), "snk_frontend_process")
        return self._result(fr, a)
