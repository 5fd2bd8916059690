"""ctypes binding of ``libsnake_hip.so`` (the C ABI declared in ``include/snake_hip.h``).

There is no CPU fallback: if the library is missing or does not load, importing the product
API raises.  Tests and the bench call the kernels only through this binding.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "lib" / "libsnake_hip.so"
if os.environ.get("SNK_HIP_LIB"):  # build-time A/B variants of the SAME library (snake_slam_amd.build.build_variant); measurements only
    LIB_PATH = Path(os.environ["SNK_HIP_LIB"]).resolve()


class SnakeHipError(RuntimeError):
    pass


class Knn2(C.Structure):
    _fields_ = [("idx1", C.c_int32), ("dist1", C.c_int32), ("idx2", C.c_int32), ("dist2", C.c_int32)]


class Kp64(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("angle", C.c_float), ("octave", C.c_int32)]


_lib = None

vp = C.c_void_p
i32 = C.c_int
f32 = C.c_float
f64 = C.c_double

# name -> (restype, argtypes).  Kept in one table so tests can check it against the header.
SIGNATURES = {
    "snk_last_error": (C.c_char_p, []),
    "snk_version": (C.c_char_p, []),
    "snk_device_count": (i32, []),
    "snk_matcher_create": (i32, [i32, vp, C.POINTER(vp)]),
    "snk_matcher_destroy": (i32, [vp]),
    "snk_matcher_sync": (i32, [vp]),
    "snk_bf_knn2": (i32, [vp, vp, i32, vp, i32, vp]),
    "snk_bf_filter": (i32, [vp, vp, i32, i32, f32, vp, C.POINTER(i32)]),
    "snk_bf_knn2_batch_dev": (i32, [vp, vp, vp, i32, vp, vp, i32, i32, vp]),
    "snk_bf_filter_batch_dev": (i32, [vp, vp, vp, i32, i32, i32, f32, vp, vp]),
    "snk_stereo_match": (i32, [vp, vp, vp, i32, vp, vp, i32, f64, vp, i32, i32, vp, vp, C.POINTER(i32)]),
    "snk_stereo_match_batch_dev": (i32, [vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, f64, vp, i32, i32, vp, vp, vp]),
    "snk_rectify": (i32, [vp, vp, vp, i32, vp, vp]),
    "snk_rectify_batch_dev": (i32, [vp, vp, vp, vp, i32, i32, vp, vp]),
    "snk_rgbd_stereo": (i32, [vp, vp, vp, i32, vp, i32, i32, i32, vp, vp, C.POINTER(i32)]),
    "snk_rgbd_stereo_batch_dev": (i32, [vp, vp, vp, vp, i32, i32, vp, i32, i32, i32, C.c_size_t, vp, vp, vp, vp]),
    "snk_feature_grid": (i32, [vp, vp, i32, vp, vp, vp, C.POINTER(i32), C.POINTER(i32)]),
    "snk_feature_grid_batch_dev": (i32, [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]),
    "snk_match_bind_frame": (i32, [vp, vp]),
    "snk_match_bound_taken": (i32, [vp, vp]),
    "snk_match_project_coarse": (i32, [vp, vp, vp, vp, vp, i32, f32, i32, i32, vp, i32, vp, C.POINTER(i32)]),
    "snk_match_project_fine": (i32, [vp, vp, vp, vp, vp, i32, f32, f32, vp, i32, vp, vp, C.POINTER(i32)]),
    "snk_match_project_coarse_batch_dev": (i32, [vp, vp, vp, vp, vp, vp, i32, f32, i32, i32, vp, i32, vp, vp]),
    "snk_match_project_fine_batch_dev": (i32, [vp, vp, vp, vp, vp, vp, i32, f32, f32, vp, i32, vp, vp, vp]),
    "snk_match_project_fine_batch_ro_dev": (i32, [vp, vp, vp, vp, vp, vp, i32, f32, f32, vp, i32, vp, vp, vp]),
    "snk_match_mark_taken_batch_dev": (i32, [vp, vp, vp, i32, i32, vp, i32]),
    "snk_match_project_keyframe": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, f32, i32, vp, C.POINTER(i32)]),
    "snk_pose_refine": (i32, [vp, vp, vp, vp, i32]),
    "snk_pose_refine_matches_batch_dev": (i32, [vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp, i32, vp, vp, vp]),
    "snk_pose_refine_frame_batch_dev": (i32, [vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp, i32, vp, vp, vp]),
    "snk_match_fuse": (i32, [vp, vp, vp, vp, vp, vp, i32, f32, f32, i32, vp, i32, vp, C.POINTER(i32)]),
    "snk_match_triangulation_project": (i32, [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, f32, i32, vp,
                                              C.POINTER(i32)]),
    "snk_match_triangulation_bow": (i32, [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, vp, f32, i32, vp, C.POINTER(i32)]),
    "snk_match_triangulation_bf": (i32, [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, vp, C.POINTER(i32)]),
    "snk_match_relink": (i32, [vp, vp, vp, vp, vp, i32, f32, f64, i32, vp, vp, C.POINTER(i32)]),
    "snk_frontend_create": (i32, [vp, i32, C.POINTER(vp)]),
    "snk_frontend_destroy": (i32, [vp]),
    "snk_frontend_max_keypoints": (i32, [vp, i32, i32, C.POINTER(i32)]),
    "snk_frontend_grid_dims": (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
    "snk_frontend_process": (i32, [vp, vp, i32, vp, i32, i32, i32, vp]),
    "snk_frontend_set_depth": (i32, [vp, i32]),
    "snk_frontend_submit": (i32, [vp, vp, i32, vp, i32, i32, i32]),
    "snk_frontend_collect": (i32, [vp, vp, i32]),
    "snk_frontend_in_flight": (i32, [vp, C.POINTER(i32)]),
    "snk_frontend_peek": (i32, [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "snk_frontend_submit_pinned": (i32, [vp, vp, i32, vp, i32, i32, i32]),
    "snk_pinned_alloc": (i32, [C.c_size_t, C.POINTER(vp)]),
    "snk_pinned_free": (i32, [vp]),
    "snk_orb_create": (i32, [vp, i32, vp, C.POINTER(vp)]),
    "snk_orb_destroy": (i32, [vp]),
    "snk_orb_sync": (i32, [vp]),
    "snk_orb_configure": (i32, [vp, i32, i32, i32]),
    "snk_orb_max_keypoints": (i32, [vp, C.POINTER(i32)]),
    "snk_orb_detect": (i32, [vp, vp, i32, i32, i32, vp, vp, i32, C.POINTER(i32)]),
    "snk_orb_detect_batch_dev": (i32, [vp, vp, i32, C.c_size_t, i32, vp, vp, vp, i32]),
    "snk_orb_set_profiling": (i32, [vp, i32]),
    "snk_orb_set_chains": (i32, [vp, i32]),
    "snk_orb_set_stagger": (i32, [vp, i32]),
    "snk_orb_stage_times": (i32, [vp, vp, C.POINTER(i32)]),
    "snk_orb_debug_fetch": (i32, [vp, i32, i32, i32, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "snk_track_bf_matches_batch_dev": (i32, [vp, vp, vp, vp, i32, i32, vp]),
    "snk_track_backproject_batch_dev": (i32, [vp, vp, vp, vp, vp, vp, vp]),
    "snk_set_definition": (i32, [C.c_char_p, i32]),
    "snk_get_definition": (i32, [C.c_char_p, C.POINTER(i32)]),
    "snk_ba_create": (i32, [vp, i32, vp, C.POINTER(vp)]),
    "snk_ba_destroy": (i32, [vp]),
    "snk_ba_sync": (i32, [vp]),
    "snk_ba_set_problem": (i32, [vp, vp]),
    "snk_ba_set_problems": (i32, [vp, vp, i32]),
    "snk_ba_set_outliers": (i32, [vp, i32, vp]),
    "snk_ba_solve": (i32, [vp, i32, vp, vp]),
    "snk_ba_solve_async": (i32, [vp, i32]),
    "snk_ba_reset": (i32, [vp]),
    "snk_ba_get_state": (i32, [vp, i32, vp, vp, C.POINTER(i32)]),
    "snk_ba_residuals": (i32, [vp, i32, vp]),
    "snk_ba_solve_local_scene": (i32, [vp, i32, f64, f64, i32, vp, vp, vp, vp, vp, vp]),
    "snk_dist_get_unique_id": (i32, [vp]),
    "snk_dist_init": (i32, [vp, i32, i32, i32, C.POINTER(vp)]),
    "snk_dist_init_file": (i32, [C.c_char_p, i32, i32, i32, f64, C.POINTER(vp)]),
    "snk_dist_destroy": (i32, [vp]),
    "snk_dist_rank": (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
    "snk_dist_all_gather": (i32, [vp, vp, C.c_size_t, vp]),
    "snk_dist_all_gather_dev": (i32, [vp, vp, C.c_size_t, vp]),
    "snk_dist_max_i64": (i32, [vp, C.c_int64, C.POINTER(C.c_int64)]),
    "snk_dist_rccl_version": (i32, [C.POINTER(i32)]),
}


def load() -> C.CDLL:
    """Load the C-ABI library; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise SnakeHipError(
            f"{LIB_PATH} not found: build it with `python -m snake_slam_amd.build` "
            "(or __graft_entry__.build()).  There is no CPU fallback."
        )
    # One HIP runtime per process: PyTorch bundles its own libamdhip64 (SONAME libamdhip64.so.7).
    # Importing torch first makes the loader resolve this library's DT_NEEDED to that same copy
    # instead of mapping /opt/rocm's next to it (two runtimes in one process fight over the GPU).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def hip_runtimes_mapped() -> list[str]:
    """Distinct libamdhip64 files mapped into this process (should be exactly one)."""
    out = set()
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                out.add(line.split()[-1])
    return sorted(out)


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().snk_last_error().decode(errors="replace")
        raise SnakeHipError(f"{what} failed with status {rc}: {msg}")


DEFINITIONS = {"bf_filter.threshold_strict": (0, 1), "bf_filter.ratio_strict": (0, 1), "iround.mode": (0, 2), "orb.response": (0, 1)}  # key -> (min, max); default 0


def set_definition(key: str, value: int) -> None:
    """snk_set_definition: select one of the [DEFINED] comparison / rounding rules (include/snake_hip.h); process-wide."""
    check(load().snk_set_definition(key.encode(), int(value)), "snk_set_definition")


def get_definition(key: str) -> int:
    v = C.c_int32(0)
    check(load().snk_get_definition(key.encode(), C.byref(v)), "snk_get_definition")
    return int(v.value)
