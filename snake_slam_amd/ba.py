"""Host-side mirror of the reference's local-BA solver interface over the C ABI.

``BARec`` mirrors how Snake drives ``Saiga::BARecRel`` (reference
Snake/Optimizer/LocalBundleAdjustment.cpp:357-365,403-407): set ``optimizationOptions`` /
``baOptions``, ``create(scene)``, ``initAndSolve()``, then ``solve()`` again after the chi-square
outlier pass.  A scene is a dict of flat arrays (the fields MakeLocalScene fills, :187-293).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class BaOptions(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("max_pcg_iterations", C.c_int32), ("pcg_tol", C.c_double),
                ("huber_mono", C.c_double), ("huber_stereo", C.c_double), ("lambda_init", C.c_double)]


class BaProblem(C.Structure):
    _fields_ = [("n_img", C.c_int32), ("n_pt", C.c_int32), ("n_obs", C.c_int32), ("pose", C.c_void_p),
                ("img_const", C.c_void_p), ("pt", C.c_void_p), ("pt_const", C.c_void_p), ("obs_img", C.c_void_p),
                ("obs_pt", C.c_void_p), ("obs_uv", C.c_void_p), ("obs_depth", C.c_void_p), ("obs_weight", C.c_void_p),
                ("K", C.c_double * 4), ("bf", C.c_double), ("n_rpc", C.c_int32), ("pad", C.c_int32), ("rpc", C.c_void_p)]


# Saiga RelPoseConstraint (IMU scenes, reference LocalBundleAdjustment.cpp:294-346)
BA_RPC_DTYPE = np.dtype([("img1", "<i4"), ("img2", "<i4"), ("rel_pose", "<f8", 7), ("weight_rotation", "<f8"),
                         ("weight_translation", "<f8")])


def lba_options(max_iterations=3, max_pcg_iterations=30, pcg_tol=1e-10, huber_mono=2.1, huber_stereo=2.3, lambda_init=0.0):
    """Defaults = reference LocalBundleAdjustment.cpp:47-64, SnakeGlobal.h:145-150."""
    return BaOptions(max_iterations, max_pcg_iterations, pcg_tol, huber_mono, huber_stereo, lambda_init)


def _pack(scene):
    a = {
        "pose": np.ascontiguousarray(scene["pose"], np.float64),
        "img_const": np.ascontiguousarray(scene["img_const"], np.uint8),
        "pt": np.ascontiguousarray(scene["pt"], np.float64),
        "pt_const": np.ascontiguousarray(scene["pt_const"], np.uint8),
        "obs_img": np.ascontiguousarray(scene["obs_img"], np.int32),
        "obs_pt": np.ascontiguousarray(scene["obs_pt"], np.int32),
        "obs_uv": np.ascontiguousarray(scene["obs_uv"], np.float64),
        "obs_depth": np.ascontiguousarray(scene["obs_depth"], np.float64),
        "obs_weight": np.ascontiguousarray(scene["obs_weight"], np.float64),
    }
    P = BaProblem()
    P.n_img, P.n_pt, P.n_obs = a["pose"].shape[0], a["pt"].shape[0], a["obs_img"].shape[0]
    for k, v in a.items():
        setattr(P, k, v.ctypes.data if v.size else 0)
    P.K[:] = list(scene["K"])
    P.bf = float(scene["bf"])
    a["rpc"] = np.ascontiguousarray(scene.get("rpc", np.zeros(0, BA_RPC_DTYPE)), BA_RPC_DTYPE)
    P.n_rpc, P.pad = len(a["rpc"]), 0
    P.rpc = a["rpc"].ctypes.data if a["rpc"].size else 0
    return P, a


class BARec:
    def __init__(self, options: BaOptions | None = None, device: int = 0, stream: int | None = None):
        self._lib = _lib.load()
        self.optimizationOptions = options or lba_options()
        h = C.c_void_p()
        _lib.check(self._lib.snk_ba_create(C.byref(self.optimizationOptions), device, C.c_void_p(stream or 0), C.byref(h)),
                   "snk_ba_create")
        self._h = h
        self._scenes = []

    def close(self):
        if getattr(self, "_h", None):
            self._lib.snk_ba_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def create(self, scene_or_scenes) -> None:
        import time

        scenes = scene_or_scenes if isinstance(scene_or_scenes, (list, tuple)) else [scene_or_scenes]
        t0 = time.perf_counter()
        packed = [_pack(s) for s in scenes]
        arr = (BaProblem * len(packed))(*[p for p, _ in packed])
        t1 = time.perf_counter()
        _lib.check(self._lib.snk_ba_set_problems(self._h, arr, len(packed)), "snk_ba_set_problems")
        # what this binding adds (numpy -> snk_ba_problem structs) and what the C call took (bench.py reports them apart: a C++ host pays
        # only the second); the call returns with the uploads enqueued, snk_ba_sync waits for them
        self.last_pack_ms, self.last_set_problems_ms = (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3
        self._scenes = scenes

    def _solve(self, iterations):
        n = len(self._scenes)
        ci = np.zeros(n, np.float64)
        cf = np.zeros(n, np.float64)
        _lib.check(self._lib.snk_ba_solve(self._h, int(iterations), C.c_void_p(ci.ctypes.data), C.c_void_p(cf.ctypes.data)),
                   "snk_ba_solve")
        return ci, cf

    def initAndSolve(self):
        """Returns (cost_initial, cost_final) per loaded scene (OptimizationResults)."""
        return self._solve(self.optimizationOptions.max_iterations)

    def solve(self, iterations=None):
        return self._solve(self.optimizationOptions.max_iterations if iterations is None else iterations)

    def solve_async(self, iterations=None):
        it = self.optimizationOptions.max_iterations if iterations is None else iterations
        _lib.check(self._lib.snk_ba_solve_async(self._h, int(it)), "snk_ba_solve_async")

    def reset(self):
        _lib.check(self._lib.snk_ba_reset(self._h), "snk_ba_reset")

    def sync(self):
        _lib.check(self._lib.snk_ba_sync(self._h), "snk_ba_sync")

    def set_outliers(self, problem: int, mask) -> None:
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        _lib.check(self._lib.snk_ba_set_outliers(self._h, problem, C.c_void_p(m.ctypes.data) if m is not None else None),
                   "snk_ba_set_outliers")

    def state(self, problem: int = 0):
        s = self._scenes[problem]
        pose = np.zeros((len(s["pose"]), 7), np.float64)
        pt = np.zeros((len(s["pt"]), 3), np.float64)
        it = C.c_int(0)
        _lib.check(self._lib.snk_ba_get_state(self._h, problem, C.c_void_p(pose.ctypes.data), C.c_void_p(pt.ctypes.data),
                                              C.byref(it)), "snk_ba_get_state")
        return pose, pt, it.value

    def solve_local_scene(self, chi2_mono: float, chi2_stereo: float, problem: int = 0, extra_iterations: int = 1):
        """`LocalBundleAdjustment::SolveLocalScene` after create() in one call (LocalBundleAdjustment.cpp:357-410): initAndSolve,
        the chi-square pass on the device, one more iteration when anything was marked.  Returns (outlierPoints, cost_initial,
        cost_final of the FIRST solve, poses, points, observation outlier flags)."""
        s = self._scenes[problem]
        pose = np.zeros((len(s["pose"]), 7), np.float64)
        pt = np.zeros((len(s["pt"]), 3), np.float64)
        flags = np.zeros(max(len(s["obs_img"]), 1), np.uint8)
        n, ci, cf = C.c_int(0), C.c_double(0), C.c_double(0)
        _lib.check(self._lib.snk_ba_solve_local_scene(self._h, problem, float(chi2_mono), float(chi2_stereo), int(extra_iterations),
                                                      C.c_void_p(flags.ctypes.data), C.byref(n), C.byref(ci), C.byref(cf),
                                                      C.c_void_p(pose.ctypes.data), C.c_void_p(pt.ctypes.data)),
                   "snk_ba_solve_local_scene")
        return n.value, ci.value, cf.value, pose, pt, flags[:len(s["obs_img"])]

    def residuals(self, problem: int = 0) -> np.ndarray:
        n = len(self._scenes[problem]["obs_img"])
        out = np.zeros(max(n, 1), np.float64)
        _lib.check(self._lib.snk_ba_residuals(self._h, problem, C.c_void_p(out.ctypes.data)), "snk_ba_residuals")
        return out[:n]


def gba_options(max_iterations=4, max_pcg_iterations=40, pcg_tol=1e-10, huber_mono=2.1, huber_stereo=2.3, lambda_init=0.0):
    """global_op_options / global_ba_options of the reference (GlobalBundleAdjustment.cpp:32-43)."""
    return BaOptions(max_iterations, max_pcg_iterations, pcg_tol, huber_mono, huber_stereo, lambda_init)


class BAPointOnly(BARec):
    """`Saiga::BAPointOnly` as GlobalBundleAdjustment::PointBA uses it (GlobalBundleAdjustment.cpp:103-122): create(scene),
    initAndSolve() -- world points optimised, every camera held.  [DEFINED] (the solver lives in the absent saiga): the same
    robust LM iteration as BARec ("snk-ba v1") on the scene with every image constant; the reduced camera system is then
    empty and each point solves its own damped 3x3 system."""

    def create(self, scene_or_scenes) -> None:
        scenes = scene_or_scenes if isinstance(scene_or_scenes, (list, tuple)) else [scene_or_scenes]
        super().create([dict(s, img_const=np.ones(len(s["pose"]), np.uint8)) for s in scenes])


class BAPoseOnly(BARec):
    """`Saiga::BAPoseOnly` as GlobalBundleAdjustment::RealignIntermiediateFrames uses it (GlobalBundleAdjustment.cpp:306-316):
    camera poses optimised (constant images stay), every world point held.  [DEFINED]: the same LM iteration as BARec on the
    scene with every point constant; the reduced system is then block diagonal (one 6x6 block per free camera)."""

    def create(self, scene_or_scenes) -> None:
        scenes = scene_or_scenes if isinstance(scene_or_scenes, (list, tuple)) else [scene_or_scenes]
        super().create([dict(s, pt_const=np.ones(len(s["pt"]), np.uint8)) for s in scenes])
