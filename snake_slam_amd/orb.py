"""Host-side mirror of the reference's extractor interface over the C ABI.

``ORBExtractor(nfeatures, scale_factor, levels, iniThFAST, minThFAST).Detect(image)`` mirrors
``Saiga::ORBExtractor`` / ``ORBExtractorGPU`` as Snake constructs and calls them (reference
Snake/Preprocess/FeatureDetector.cpp:31-41,119,124): one synchronous call per image returning
index-aligned keypoints and 256-bit descriptors.  ``detect_batch_dev`` is the device-resident
batched form used by the benchmark.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4")])

DEBUG_PYRAMID, DEBUG_CELL_COUNTS, DEBUG_CELL_CANDIDATES, DEBUG_SELECTED, DEBUG_SELECTED_COUNT, DEBUG_LEVEL_INFO = range(1, 7)


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("n_levels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("level_cap", C.c_int32)]


class ORBExtractor:
    def __init__(self, nfeatures: int = 1000, scale_factor: float = 1.2, n_levels: int = 4, ini_th_fast: int = 20,
                 min_th_fast: int = 7, level_cap: int = 0, device: int = 0, stream: int | None = None):
        self._lib = _lib.load()
        self.params = OrbParams(nfeatures, scale_factor, n_levels, ini_th_fast, min_th_fast, level_cap)
        h = C.c_void_p()
        _lib.check(self._lib.snk_orb_create(C.byref(self.params), device, C.c_void_p(stream or 0), C.byref(h)),
                   "snk_orb_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.snk_orb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _lib.check(self._lib.snk_orb_sync(self._h), "snk_orb_sync")

    def configure(self, width: int, height: int, max_batch: int = 1) -> int:
        _lib.check(self._lib.snk_orb_configure(self._h, width, height, max_batch), "snk_orb_configure")
        return self.max_keypoints()

    def max_keypoints(self) -> int:
        n = C.c_int(0)
        _lib.check(self._lib.snk_orb_max_keypoints(self._h, C.byref(n)), "snk_orb_max_keypoints")
        return n.value

    def Detect(self, image: np.ndarray):
        """image: uint8 [H, W] (any row pitch).  Returns (keypoints KEYPOINT_DTYPE[N], descriptors uint64[N,4])."""
        if image.dtype != np.uint8 or image.ndim != 2 or image.strides[1] != 1:
            image = np.ascontiguousarray(image, np.uint8)
        h, w = image.shape
        self.configure(w, h, 1)
        cap = max(self.max_keypoints(), 1)
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 4), np.uint64)
        n = C.c_int(0)
        _lib.check(
            self._lib.snk_orb_detect(self._h, C.c_void_p(image.ctypes.data), w, h, image.strides[0],
                                     C.c_void_p(kps.ctypes.data), C.c_void_p(desc.ctypes.data), cap, C.byref(n)),
            "snk_orb_detect",
        )
        return kps[: n.value].copy(), desc[: n.value].copy()

    def detect_batch_dev(self, images, kps, desc, n):
        """images: uint8 cuda tensor [B, H, pitch]; kps: uint8 [B, cap, 24]; desc: int64 [B, cap, 4]; n: int32 [B]."""
        B, H, pitch = images.shape
        _lib.check(
            self._lib.snk_orb_detect_batch_dev(self._h, images.data_ptr(), pitch, H * pitch, B, kps.data_ptr(),
                                               desc.data_ptr(), n.data_ptr(), desc.shape[1]),
            "snk_orb_detect_batch_dev",
        )

    def set_profiling(self, enable: bool) -> None:
        _lib.check(self._lib.snk_orb_set_profiling(self._h, int(enable)), "snk_orb_set_profiling")

    def set_chains(self, chains: int) -> None:
        """Launch chains per detect_batch_dev call (1 = one chain on the handle's stream, 2 = two half batches on two streams)."""
        _lib.check(self._lib.snk_orb_set_chains(self._h, int(chains)), "snk_orb_set_chains")

    def set_stagger(self, parts: int) -> None:
        """Staggered schedule (0 = off, 2..16 parts): front halves back to back on the handle's stream, the back half of part p on a
        second stream beside the front half of part p + 1."""
        _lib.check(self._lib.snk_orb_set_stagger(self._h, int(parts)), "snk_orb_set_stagger")

    def stage_times(self):
        """(ms per stage [pyramid, blur, fast, distribute, describe] summed over calls, number of calls)."""
        ms = (C.c_float * 5)()
        n = C.c_int(0)
        _lib.check(self._lib.snk_orb_stage_times(self._h, ms, C.byref(n)), "snk_orb_stage_times")
        return list(ms), n.value

    def debug_fetch(self, what: int, image: int, level: int, dtype, max_bytes: int = 1 << 24) -> np.ndarray:
        buf = np.zeros(max_bytes, np.uint8)
        nb = C.c_size_t(0)
        _lib.check(self._lib.snk_orb_debug_fetch(self._h, what, image, level, C.c_void_p(buf.ctypes.data), max_bytes,
                                                 C.byref(nb)), "snk_orb_debug_fetch")
        return buf[: nb.value].view(dtype).copy()
