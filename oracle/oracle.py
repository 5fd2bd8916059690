"""TEST INFRASTRUCTURE — ctypes loader for the CPU oracle (oracle/liborc.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED: see oracle/*.c headers.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "liborc.so"

KP64 = np.dtype([("x", "<f8"), ("y", "<f8"), ("angle", "<f4"), ("octave", "<i4")])
KNN2 = np.dtype([("idx1", "<i4"), ("dist1", "<i4"), ("idx2", "<i4"), ("dist2", "<i4")])

_lib = None


def build(force: bool = False) -> Path:
    srcs = list(HERE.glob("*.c")) + list(HERE.glob("*.h")) + [HERE / "Makefile"]
    if force or not LIB.exists() or any(s.stat().st_mtime > LIB.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-s", "-C", str(HERE)] + (["-B"] if force else []), check=True)
    return LIB


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB.exists():
            build()
        _lib = C.CDLL(str(LIB))
        _lib.orc_hamming.restype = C.c_int
        _lib.orc_bf_filter.restype = C.c_int
        _lib.orc_stereo_match.restype = C.c_int
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data if a.size else 0)


def hamming(a, b) -> int:
    a = np.ascontiguousarray(a, np.uint64)
    b = np.ascontiguousarray(b, np.uint64)
    return lib().orc_hamming(_p(a), _p(b))


def set_definition(key: str, value: int) -> None:
    """Mirror of snk_set_definition: the same keys / values select the same [DEFINED] rule in the oracle."""
    lib().orc_set_definition.restype = C.c_int
    if lib().orc_set_definition(key.encode(), C.c_int(int(value))) != 0:
        raise ValueError(f"oracle: unknown definition {key!r} or bad value {value}")


def bf_knn2(q, t, threads: int = 1) -> np.ndarray:
    q = np.ascontiguousarray(q, np.uint64).reshape(-1, 4)
    t = np.ascontiguousarray(t, np.uint64).reshape(-1, 4)
    out = np.zeros(q.shape[0], KNN2)
    lib().orc_bf_knn2(_p(q), C.c_int(q.shape[0]), _p(t), C.c_int(t.shape[0]), _p(out), C.c_int(threads))
    return out


def bf_filter(knn, threshold: int, ratio: float) -> np.ndarray:
    knn = np.ascontiguousarray(knn, KNN2)
    pairs = np.zeros((max(knn.shape[0], 1), 2), np.int32)
    n = lib().orc_bf_filter(_p(knn), C.c_int(knn.shape[0]), C.c_int(threshold), C.c_float(ratio), _p(pairs))
    return pairs[:n].copy()


def stereo_match(left, dl, right, dr, bf, level_scale, relaxed=True, right_points=None, depth=None):
    left = np.ascontiguousarray(left, KP64)
    right = np.ascontiguousarray(right, KP64)
    dl = np.ascontiguousarray(dl, np.uint64).reshape(-1, 4)
    dr = np.ascontiguousarray(dr, np.uint64).reshape(-1, 4)
    nl = left.shape[0]
    rp = np.full(nl, -1000.0, np.float32) if right_points is None else np.array(right_points, np.float32)
    dp = np.full(nl, -1000.0, np.float32) if depth is None else np.array(depth, np.float32)
    ls = np.ascontiguousarray(level_scale, np.float32)
    n = lib().orc_stereo_match(_p(left), _p(dl), C.c_int(nl), _p(right), _p(dr), C.c_int(right.shape[0]),
                               C.c_double(bf), _p(ls), C.c_int(int(relaxed)), _p(rp), _p(dp))
    return n, rp, dp


# ------------------------------------------------------------------ ORB ------------------------
class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("n_levels", C.c_int32),
                ("ini_th", C.c_int32), ("min_th", C.c_int32)]


class OrbLayout(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("scale", C.c_float * 16), ("w", C.c_int32 * 16), ("h", C.c_int32 * 16),
                ("nfeat", C.c_int32 * 16)]


KEYPOINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4")])
CAND = np.dtype([("x", "<u2"), ("y", "<u2"), ("score", "<u2"), ("cell", "<u2")])


def orb_params(nfeatures=1000, scale_factor=1.2, n_levels=4, ini_th=20, min_th=7) -> OrbParams:
    return OrbParams(nfeatures, scale_factor, n_levels, ini_th, min_th)


def orb_layout(p: OrbParams, w: int, h: int) -> OrbLayout:
    L = OrbLayout()
    rc = lib().orc_orb_layout(C.byref(p), w, h, C.byref(L))
    if rc != 0:
        raise ValueError("bad ORB parameters")
    return L


def resize(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(dst), dw, dh, dw)
    return dst


def pyramid(p: OrbParams, img: np.ndarray):
    L = orb_layout(p, img.shape[1], img.shape[0])
    levels = [np.ascontiguousarray(img, np.uint8)]
    for l in range(1, L.n_levels):
        levels.append(resize(levels[-1], L.w[l], L.h[l]))
    return levels, L


def fast_score(img: np.ndarray, x: int, y: int) -> int:
    lib().orc_fast_score.restype = C.c_int
    return lib().orc_fast_score(_p(img), img.shape[1], x, y)


def candidates(img: np.ndarray, ini_th=20, min_th=7, cap=8192) -> np.ndarray:
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros(cap, CAND)
    lib().orc_orb_candidates.restype = C.c_int
    n = lib().orc_orb_candidates(_p(img), img.shape[1], img.shape[0], img.shape[1], ini_th, min_th, _p(out), cap)
    return out[:n].copy()


def distribute(cands: np.ndarray, w: int, h: int, N: int) -> np.ndarray:
    cands = np.ascontiguousarray(cands, CAND)
    out = np.zeros(lib().orc_orb_distribute_bound(w, h, N) + 8, np.int32)
    lib().orc_orb_distribute.restype = C.c_int
    n = lib().orc_orb_distribute(_p(cands), cands.shape[0], w, h, N, _p(out))
    return out[:n].copy()


def distribute_ranked(cands: np.ndarray, rank: np.ndarray, w: int, h: int, N: int) -> np.ndarray:
    cands = np.ascontiguousarray(cands, CAND)
    rank = np.ascontiguousarray(rank, np.uint32)
    out = np.zeros(lib().orc_orb_distribute_bound(w, h, N) + 8, np.int32)
    lib().orc_orb_distribute_ranked.restype = C.c_int
    n = lib().orc_orb_distribute_ranked(_p(cands), _p(rank), cands.shape[0], w, h, N, _p(out))
    return out[:n].copy()


def harris_abc(img: np.ndarray, x: int, y: int):
    """The three integer sums of OpenCV's HarrisResponses over the 7 x 7 block around (x, y)."""
    a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
    lib().orc_harris_abc(_p(img), img.shape[1], int(x), int(y), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def harris_response(img: np.ndarray, x: int, y: int) -> np.float32:
    lib().orc_harris_response.restype = C.c_float
    return np.float32(lib().orc_harris_response(_p(img), img.shape[1], int(x), int(y)))


def harris_rank(r) -> int:
    lib().orc_harris_rank.restype = C.c_uint32
    return lib().orc_harris_rank(C.c_float(float(r)))


def point_key(x: int, y: int, W: int, H: int) -> int:
    lib().orc_point_key.restype = C.c_uint64
    return lib().orc_point_key(x, y, W, H)


def ic_moments(img: np.ndarray, x: int, y: int):
    m10, m01 = C.c_int(), C.c_int()
    lib().orc_ic_moments(_p(img), img.shape[1], x, y, C.byref(m10), C.byref(m01))
    return m10.value, m01.value


def fast_atan2(y: float, x: float) -> np.float32:
    lib().orc_fast_atan2.restype = C.c_float
    return np.float32(lib().orc_fast_atan2(C.c_float(y), C.c_float(x)))


def sincos_deg(deg: float):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincos_deg(C.c_float(deg), C.byref(s), C.byref(c))
    return np.float32(s.value), np.float32(c.value)


def blur_at(img: np.ndarray, x: int, y: int) -> int:
    lib().orc_blur_at.restype = C.c_int
    return lib().orc_blur_at(_p(img), img.shape[1], img.shape[0], img.shape[1], x, y)


def blur_image(img: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros_like(img)
    lib().orc_blur_image(_p(img), img.shape[1], img.shape[0], img.shape[1], _p(out), img.shape[1])
    return out


def descriptor(img: np.ndarray, x: int, y: int, angle: float) -> np.ndarray:
    out = np.zeros(4, np.uint64)
    lib().orc_descriptor(_p(img), img.shape[1], img.shape[0], img.shape[1], x, y, C.c_float(angle), _p(out))
    return out


def descriptor_on_plane(plane: np.ndarray, x: int, y: int, angle: float) -> np.ndarray:
    """The 256 steered tests evaluated on the plane GIVEN (what orb_detect hands the blurred level): isolates steering, rounding and
    bit order from the blur (tests/test_oracle_pin.py pins it against scikit-image's _orb_loop on an unblurred plane)."""
    plane = np.ascontiguousarray(plane, np.uint8)
    out = np.zeros(4, np.uint64)
    lib().orc_descriptor_blurred(_p(plane), plane.shape[1], x, y, C.c_float(angle), _p(out))
    return out


def brief_pattern() -> np.ndarray:
    arr = (C.c_int8 * 1024).in_dll(lib(), "orc_brief_pattern")
    return np.frombuffer(arr, np.int8).copy()


def orb_detect(p: OrbParams, img: np.ndarray, level_cap: int = 0, threads: int = 1):
    if img.dtype != np.uint8 or img.strides[1] != 1:
        img = np.ascontiguousarray(img, np.uint8)  # row pitch (strides[0]) may exceed the width
    L = orb_layout(p, img.shape[1], img.shape[0])
    cap = sum(lib().orc_orb_distribute_bound(L.w[l], L.h[l], L.nfeat[l]) for l in range(L.n_levels)) + 8
    kps = np.zeros(cap, KEYPOINT)
    desc = np.zeros((cap, 4), np.uint64)
    lib().orc_orb_detect.restype = C.c_int
    n = lib().orc_orb_detect(C.byref(p), _p(img), img.shape[1], img.shape[0], img.strides[0], _p(kps), _p(desc), cap,
                             level_cap, threads)
    if n < 0:
        raise RuntimeError(f"orc_orb_detect failed: {n}")
    return kps[:n].copy(), desc[:n].copy()


# ------------------------------------------------------------------ Preprocess -----------------
class Rectification(C.Structure):
    _fields_ = [("K_src", C.c_double * 4), ("D_src", C.c_double * 8), ("R", C.c_double * 9), ("K_dst", C.c_double * 4),
                ("bf", C.c_double)]


def rectification(K_src, D_src=None, R=None, K_dst=None, bf=0.0) -> Rectification:
    r = Rectification()
    r.K_src[:] = list(K_src)
    r.D_src[:] = list(D_src) if D_src is not None else [0.0] * 8
    r.R[:] = list(np.asarray(R, np.float64).reshape(9)) if R is not None else [1, 0, 0, 0, 1, 0, 0, 0, 1]
    r.K_dst[:] = list(K_dst) if K_dst is not None else list(K_src)
    r.bf = bf
    return r


def rectify(rect: Rectification, kps):
    k = np.ascontiguousarray(kps, KEYPOINT)
    out = np.zeros(k.shape[0], KP64)
    norm = np.zeros((k.shape[0], 2), np.float64)
    lib().orc_rectify(C.byref(rect), _p(k), C.c_int(k.shape[0]), _p(out), _p(norm))
    return out, norm


def rgbd_stereo(und, K, D_depth, K_depth, bf, depth_image):
    """Preprocess::ComputeStereoFromRGBD.  Returns (matches, right_points, depth); matches < 0: the reference would abort at keypoint
    -matches - 1 (outside the depth image / depth not in [0, 20))."""
    und = np.ascontiguousarray(und, KP64)
    img = np.ascontiguousarray(depth_image, np.float32)
    K = np.ascontiguousarray(K, np.float64)
    Dd = np.zeros(8, np.float64)
    Dd[: len(D_depth)] = D_depth
    Kd = np.ascontiguousarray(K_depth, np.float64)
    rp, dp = np.zeros(max(len(und), 1), np.float32), np.zeros(max(len(und), 1), np.float32)
    lib().orc_rgbd_stereo.restype = C.c_int
    n = lib().orc_rgbd_stereo(_p(und), C.c_int(len(und)), _p(K), _p(Dd), _p(Kd), C.c_double(bf), _p(img), C.c_int(img.shape[1]),
                              C.c_int(img.shape[0]), C.c_int(img.shape[1]), _p(rp), _p(dp))
    return n, rp[: len(und)], dp[: len(und)]


# ------------------------------------------------------------------ BA -------------------------
class BaOptions(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("max_pcg_iterations", C.c_int32), ("pcg_tol", C.c_double),
                ("huber_mono", C.c_double), ("huber_stereo", C.c_double), ("lambda_init", C.c_double)]


class BaProblem(C.Structure):
    _fields_ = [("n_img", C.c_int32), ("n_pt", C.c_int32), ("n_obs", C.c_int32), ("pose", C.c_void_p),
                ("img_const", C.c_void_p), ("pt", C.c_void_p), ("pt_const", C.c_void_p), ("obs_img", C.c_void_p),
                ("obs_pt", C.c_void_p), ("obs_uv", C.c_void_p), ("obs_depth", C.c_void_p), ("obs_weight", C.c_void_p),
                ("obs_outlier", C.c_void_p), ("K", C.c_double * 4), ("bf", C.c_double), ("n_rpc", C.c_int32), ("pad", C.c_int32),
                ("rpc", C.c_void_p)]


BA_RPC = np.dtype([("img1", "<i4"), ("img2", "<i4"), ("rel_pose", "<f8", 7), ("weight_rotation", "<f8"),
                   ("weight_translation", "<f8")])


def ba_options(max_iterations=3, max_pcg_iterations=30, pcg_tol=1e-10, huber_mono=2.1, huber_stereo=2.3, lambda_init=0.0):
    """Defaults = reference Snake/Optimizer/LocalBundleAdjustment.cpp:47-64 and SnakeGlobal.h:145-150."""
    return BaOptions(max_iterations, max_pcg_iterations, pcg_tol, huber_mono, huber_stereo, lambda_init)


def _ba_pack(scene, outlier=None):
    """scene: dict with pose[n,7], img_const[n], pt[m,3], pt_const[m], obs_img, obs_pt, obs_uv[k,2], obs_depth,
    obs_weight, K[4], bf.  Returns (BaProblem, keepalive arrays)."""
    a = {
        "pose": np.array(scene["pose"], np.float64, order="C"),
        "img_const": np.ascontiguousarray(scene["img_const"], np.uint8),
        "pt": np.array(scene["pt"], np.float64, order="C"),
        "pt_const": np.ascontiguousarray(scene["pt_const"], np.uint8),
        "obs_img": np.ascontiguousarray(scene["obs_img"], np.int32),
        "obs_pt": np.ascontiguousarray(scene["obs_pt"], np.int32),
        "obs_uv": np.ascontiguousarray(scene["obs_uv"], np.float64),
        "obs_depth": np.ascontiguousarray(scene["obs_depth"], np.float64),
        "obs_weight": np.ascontiguousarray(scene["obs_weight"], np.float64),
    }
    if outlier is not None:
        a["obs_outlier"] = np.ascontiguousarray(outlier, np.uint8)
    P = BaProblem()
    P.n_img, P.n_pt, P.n_obs = a["pose"].shape[0], a["pt"].shape[0], a["obs_img"].shape[0]
    for k in ("pose", "img_const", "pt", "pt_const", "obs_img", "obs_pt", "obs_uv", "obs_depth", "obs_weight"):
        setattr(P, k, a[k].ctypes.data if a[k].size else 0)
    P.obs_outlier = a["obs_outlier"].ctypes.data if outlier is not None else 0
    P.K[:] = list(scene["K"])
    P.bf = float(scene["bf"])
    a["rpc"] = np.ascontiguousarray(scene.get("rpc", np.zeros(0, BA_RPC)), BA_RPC)
    P.n_rpc, P.pad = len(a["rpc"]), 0
    P.rpc = a["rpc"].ctypes.data if a["rpc"].size else 0
    return P, a


def ba_rpc_linearize(pose1, pose2, rpc):
    """(r[6], J1[6,6]) of one relative pose constraint (d r / d delta2 = diag(w_t x3, w_r x3))."""
    q = np.ascontiguousarray(rpc, BA_RPC).reshape(1)
    r, J1 = np.zeros(6), np.zeros((6, 6))
    lib().orc_ba_rpc_linearize(_p(np.ascontiguousarray(pose1, np.float64)), _p(np.ascontiguousarray(pose2, np.float64)), _p(q),
                               _p(r), _p(J1))
    return r, J1


def ba_chi2(scene, outlier=None) -> np.ndarray:
    P, a = _ba_pack(scene, outlier)
    out = np.zeros(P.n_obs, np.float64)
    lib().orc_ba_chi2(C.byref(P), _p(out))
    return out


def ba_solve(scene, options: BaOptions = None, iterations=None, outlier=None, sum_order: int = 0):
    """Returns (pose, pt, cost_initial, cost_final, pcg_iterations).  sum_order != 0: the oracle with its floating-point sums
    accumulated in another order (1 reversed, 2 pairwise; ba_oracle.c orc_ba_set_sum_order) -- the control of the truncated-PCG
    rule in tests/ba_parity.py, never used by a parity comparison."""
    options = options or ba_options()
    P, a = _ba_pack(scene, outlier)
    ci, cf, it = C.c_double(), C.c_double(), C.c_int()
    lib().orc_ba_set_sum_order(C.c_int(int(sum_order)))
    try:
        lib().orc_ba_solve(C.byref(P), C.byref(options), C.c_int(options.max_iterations if iterations is None else iterations),
                           C.byref(ci), C.byref(cf), C.byref(it))
    finally:
        lib().orc_ba_set_sum_order(C.c_int(0))
    return a["pose"], a["pt"], ci.value, cf.value, it.value


# ------------------------------------------------------------------ tracking matchers ----------
class GridBounds(C.Structure):
    _fields_ = [("min_x", C.c_double), ("min_y", C.c_double), ("max_x", C.c_double), ("max_y", C.c_double)]


class FrameView(C.Structure):
    _fields_ = [("n", C.c_int32), ("cols", C.c_int32), ("rows", C.c_int32), ("kps", C.c_void_p), ("desc", C.c_void_p),
                ("right_points", C.c_void_p), ("taken", C.c_void_p), ("cell_start", C.c_void_p), ("bounds", GridBounds)]


class Camera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double)]


LM_COARSE = np.dtype([("pos", "<f8", 3), ("normal", "<f8", 3), ("desc", "<u8", 4), ("octave", "<i4"), ("angle", "<f4")])
LM_FINE = np.dtype([("pos", "<f8", 3), ("normal", "<f8", 3), ("desc", "<u8", 4), ("reference_depth", "<f4"),
                    ("reference_scale_level", "<i4"), ("valid", "u1"), ("pad", "u1", 7)])


def det_log(x):
    lib().orc_det_log.restype = C.c_double
    return lib().orc_det_log(C.c_double(x))


def det_exp(x):
    lib().orc_det_exp.restype = C.c_double
    return lib().orc_det_exp(C.c_double(x))


def grid_dims(bounds):
    b = GridBounds(*bounds)
    c, r = C.c_int(), C.c_int()
    lib().orc_grid_dims(C.byref(b), C.byref(c), C.byref(r))
    return c.value, r.value


def feature_grid(kps, bounds):
    """Returns (perm [n] new index of old feature, cell_start [cols*rows+1], cols, rows)."""
    kps = np.ascontiguousarray(kps, KP64)
    cols, rows = grid_dims(bounds)
    perm = np.zeros(max(len(kps), 1), np.int32)
    cs = np.zeros(cols * rows + 1, np.int32)
    b = GridBounds(*bounds)
    lib().orc_feature_grid(_p(kps), C.c_int(len(kps)), C.byref(b), _p(perm), _p(cs))
    return perm[: len(kps)], cs, cols, rows


def make_frame_view(frame):
    """frame: dict kps (KP64, grid order), desc, right_points, taken, cell_start, bounds (4), cols, rows."""
    a = {"kps": np.ascontiguousarray(frame["kps"], KP64), "desc": np.ascontiguousarray(frame["desc"], np.uint64),
         "right_points": np.ascontiguousarray(frame["right_points"], np.float32),
         "taken": np.ascontiguousarray(frame["taken"], np.uint8), "cell_start": np.ascontiguousarray(frame["cell_start"], np.int32)}
    v = FrameView()
    v.n, v.cols, v.rows = len(a["kps"]), frame["cols"], frame["rows"]
    for k in a:
        setattr(v, k, a[k].ctypes.data if a[k].size else 0)
    v.bounds = GridBounds(*frame["bounds"])
    return v, a


def set_match_threads(n: int) -> None:
    """Threads of the per-point phase of match_coarse / match_fine (the reference: num_tracking_threads = 4)."""
    lib().orc_set_match_threads(C.c_int(int(n)))


def match_coarse(frame, cam, pose, pts, th, feature_error, direction, level_scale):
    v, keep = make_frame_view(frame)
    pts = np.ascontiguousarray(pts, LM_COARSE)
    ls = np.ascontiguousarray(level_scale, np.float32)
    pose = np.ascontiguousarray(pose, np.float64)
    out = np.zeros(max(len(pts), 1), np.int32)
    c = Camera(*cam)
    lib().orc_match_coarse.restype = C.c_int
    n = lib().orc_match_coarse(C.byref(v), C.byref(c), _p(pose), _p(pts), C.c_int(len(pts)), C.c_float(th), C.c_int(feature_error),
                               C.c_int(direction), _p(ls), C.c_int(len(ls)), _p(out))
    return n, out[: len(pts)]


def match_fine(frame, cam, pose, pts, th, ratio, level_scale):
    """Returns (n, match_idx, visible, valid_out)."""
    v, keep = make_frame_view(frame)
    pts = np.array(pts, LM_FINE, order="C")
    ls = np.ascontiguousarray(level_scale, np.float32)
    pose = np.ascontiguousarray(pose, np.float64)
    out = np.zeros(max(len(pts), 1), np.int32)
    vis = np.zeros(max(len(pts), 1), np.uint8)
    c = Camera(*cam)
    lib().orc_match_fine.restype = C.c_int
    n = lib().orc_match_fine(C.byref(v), C.byref(c), _p(pose), _p(pts), C.c_int(len(pts)), C.c_float(th), C.c_float(ratio),
                             _p(ls), C.c_int(len(ls)), _p(out), _p(vis))
    return n, out[: len(pts)], vis[: len(pts)], pts["valid"].copy()


def match_keyframe(frame, cam, pose, pos, desc, skip, th, feature_error):
    v, keep = make_frame_view(frame)
    pos = np.ascontiguousarray(pos, np.float64).reshape(-1, 3)
    desc = np.ascontiguousarray(desc, np.uint64).reshape(-1, 4)
    skip = np.ascontiguousarray(skip, np.uint8)
    pose = np.ascontiguousarray(pose, np.float64)
    out = np.zeros(max(len(pos), 1), np.int32)
    c = Camera(*cam)
    lib().orc_match_keyframe.restype = C.c_int
    n = lib().orc_match_keyframe(C.byref(v), C.byref(c), _p(pose), _p(pos), _p(desc), _p(skip), C.c_int(len(pos)), C.c_float(th),
                                 C.c_int(feature_error), _p(out))
    return n, out[: len(pos)]


FUSION_POINT = np.dtype([("pos", "<f8", 3), ("normal", "<f8", 3), ("desc", "<u8", 4), ("reference_depth", "<f4"),
                         ("reference_scale_level", "<i4"), ("observations", "<i4"), ("id", "<i4")])


def match_fuse(frame, cam, pose, pts, point_mask, th, obs_factor, feature_th, level_scale):
    """MappingORBMatcher::Fuse (LocalMap overload).  Returns (fused, best_idx[m])."""
    v, keep = make_frame_view(frame)
    pts = np.ascontiguousarray(pts, FUSION_POINT)
    ls = np.ascontiguousarray(level_scale, np.float32)
    pose = np.ascontiguousarray(pose, np.float64)
    mask = None if point_mask is None else np.ascontiguousarray(point_mask, np.uint8)
    out = np.zeros(max(len(pts), 1), np.int32)
    c = Camera(*cam)
    lib().orc_match_fuse.restype = C.c_int
    n = lib().orc_match_fuse(C.byref(v), C.byref(c), _p(pose), _p(pts), None if mask is None else _p(mask), C.c_int(len(pts)),
                             C.c_float(th), C.c_float(obs_factor), C.c_int(feature_th), _p(ls), C.c_int(len(ls)), _p(out))
    return n, out[: len(pts)]


def match_triangulation_project(depth_grid, pose1, pose2, cam, kps1, np1, desc1, has_mp1, frame2, np2, E12, epipolar_distance,
                                feature_distance):
    """MappingORBMatcher::SearchForTriangulationProject.  Returns (n, match_idx2[n1])."""
    v, keep = make_frame_view(frame2)
    g = np.ascontiguousarray(depth_grid, np.float64)
    kps1 = np.ascontiguousarray(kps1, KP64)
    np1 = np.ascontiguousarray(np1, np.float64).reshape(-1, 2)
    np2 = np.ascontiguousarray(np2, np.float64).reshape(-1, 2)
    desc1 = np.ascontiguousarray(desc1, np.uint64).reshape(-1, 4)
    has1 = np.ascontiguousarray(has_mp1, np.uint8)
    p1, p2 = np.ascontiguousarray(pose1, np.float64), np.ascontiguousarray(pose2, np.float64)
    E = np.ascontiguousarray(E12, np.float64).reshape(9)
    out = np.zeros(max(len(kps1), 1), np.int32)
    c = Camera(*cam)
    lib().orc_match_triangulation_project.restype = C.c_int
    n = lib().orc_match_triangulation_project(_p(g), C.c_int(g.shape[0]), C.c_int(g.shape[1]), _p(p1), _p(p2), C.byref(c),
                                              _p(kps1), _p(np1), _p(desc1), _p(has1), C.c_int(len(kps1)), C.byref(v), _p(np2),
                                              _p(E), C.c_float(epipolar_distance), C.c_int(feature_distance), _p(out))
    return n, out[: len(kps1)]


RELINK_QUERY = np.dtype([("pos", "f8", 3), ("desc", "u8", 4), ("alt_desc", "u8", 4), ("feature", "i4"), ("has_alt", "i4")])


def match_relink(frame, cam, pose, queries, radius=0.8, outlier_threshold=2.1, feature_threshold=25):
    """DeferredMapper::Relink, per-observation search.  Returns (n_changed, action[n], best_idx[n])."""
    v, keep = make_frame_view(frame)
    q = np.ascontiguousarray(queries, RELINK_QUERY)
    p = np.ascontiguousarray(pose, np.float64)
    action = np.zeros(max(len(q), 1), np.int32)
    best = np.zeros(max(len(q), 1), np.int32)
    c = Camera(*cam)
    lib().orc_match_relink.restype = C.c_int
    n = lib().orc_match_relink(C.byref(v), C.byref(c), _p(p), _p(q), C.c_int(len(q)), C.c_float(radius), C.c_double(outlier_threshold),
                               C.c_int(feature_threshold), _p(action), _p(best))
    return n, action[: len(q)], best[: len(q)]


def bow_arrays(bow):
    """bow: (node_id[k] ascending, node_start[k + 1], features[...]) -> contiguous arrays of the C layout."""
    ids = np.ascontiguousarray(bow[0], np.uint32)
    start = np.ascontiguousarray(bow[1], np.int32)
    feat = np.ascontiguousarray(bow[2], np.int32)
    assert len(start) == len(ids) + 1 and (len(ids) == 0 or start[-1] == len(feat))
    return ids, start, feat


def match_triangulation_bow(cam, E12, np1, desc1, has_mp1, bow1, np2, desc2, has_mp2, bow2, epipolar_distance, feature_distance):
    """MappingORBMatcher::SearchForTriangulation2.  Returns (n, pairs[n, 2]) in the reference's emplace order."""
    np1 = np.ascontiguousarray(np1, np.float64).reshape(-1, 2)
    np2 = np.ascontiguousarray(np2, np.float64).reshape(-1, 2)
    d1 = np.ascontiguousarray(desc1, np.uint64).reshape(-1, 4)
    d2 = np.ascontiguousarray(desc2, np.uint64).reshape(-1, 4)
    h1, h2 = np.ascontiguousarray(has_mp1, np.uint8), np.ascontiguousarray(has_mp2, np.uint8)
    i1, s1, f1 = bow_arrays(bow1)
    i2, s2, f2 = bow_arrays(bow2)
    E = np.ascontiguousarray(E12, np.float64).reshape(9)
    pairs = np.zeros((max(len(f1), 1), 2), np.int32)
    c = Camera(*cam)
    lib().orc_match_triangulation_bow.restype = C.c_int
    n = lib().orc_match_triangulation_bow(C.byref(c), _p(E), _p(np1), _p(d1), _p(h1), C.c_int(len(i1)), _p(i1), _p(s1), _p(f1),
                                          _p(np2), _p(d2), _p(h2), C.c_int(len(i2)), _p(i2), _p(s2), _p(f2),
                                          C.c_float(epipolar_distance), C.c_int(feature_distance), _p(pairs))
    return n, pairs[:n]


def match_triangulation_bf(cam, E12, np1, desc1, has_mp1, np2, desc2, has_mp2, feature_distance):
    """MappingORBMatcher::SearchForTriangulationBF.  Returns (n, match_idx2[n1])."""
    np1 = np.ascontiguousarray(np1, np.float64).reshape(-1, 2)
    np2 = np.ascontiguousarray(np2, np.float64).reshape(-1, 2)
    d1 = np.ascontiguousarray(desc1, np.uint64).reshape(-1, 4)
    d2 = np.ascontiguousarray(desc2, np.uint64).reshape(-1, 4)
    h1, h2 = np.ascontiguousarray(has_mp1, np.uint8), np.ascontiguousarray(has_mp2, np.uint8)
    E = np.ascontiguousarray(E12, np.float64).reshape(9)
    out = np.zeros(max(len(np1), 1), np.int32)
    c = Camera(*cam)
    lib().orc_match_triangulation_bf.restype = C.c_int
    n = lib().orc_match_triangulation_bf(C.byref(c), _p(E), _p(np1), _p(d1), _p(h1), C.c_int(len(np1)), _p(np2), _p(d2), _p(h2),
                                         C.c_int(len(np2)), C.c_int(feature_distance), _p(out))
    return n, out[: len(np1)]


# ------------------------------------------------------------------ pose refinement ------------
class PoseObs(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("depth", C.c_double), ("weight", C.c_double)]


POSE_OBS = np.dtype([("x", "f8"), ("y", "f8"), ("depth", "f8"), ("weight", "f8")])


class PoseOptions(C.Structure):
    _fields_ = [("th_mono", C.c_double), ("th_stereo", C.c_double), ("outer_iterations", C.c_int32),
                ("inner_iterations", C.c_int32), ("robust_rounds", C.c_int32), ("pad", C.c_int32), ("lambda_", C.c_double)]


def pose_options(th_mono=2.1, th_stereo=2.3, outer=4, inner=10, robust_rounds=3, lam=1e-4) -> PoseOptions:
    return PoseOptions(th_mono, th_stereo, outer, inner, robust_rounds, 0, lam)


def se3_log_rel(pose, pred) -> np.ndarray:
    e = np.zeros(6)
    lib().orc_se3_log_rel(_p(np.ascontiguousarray(pose, np.float64)), _p(np.ascontiguousarray(pred, np.float64)), _p(e))
    return e


def pose_chi2(pose, cam, wps, obs) -> np.ndarray:
    wps = np.ascontiguousarray(wps, np.float64)
    obs = np.ascontiguousarray(obs, POSE_OBS)
    out = np.zeros(len(obs))
    lib().orc_pose_chi2(_p(np.ascontiguousarray(pose, np.float64)), C.byref(cam), _p(wps), _p(obs), len(obs), _p(out))
    return out


def pose_refine(pose, cam, wps, obs, options: PoseOptions = None, prediction=None, w_rot=0.0, w_trans=0.0):
    """Returns (pose[7], outlier[n] uint8, inliers)."""
    options = options or pose_options()
    pose = np.array(pose, np.float64)
    wps = np.ascontiguousarray(wps, np.float64)
    obs = np.ascontiguousarray(obs, POSE_OBS)
    outl = np.zeros(len(obs), np.uint8)
    pred = None if prediction is None else np.ascontiguousarray(prediction, np.float64)
    lib().orc_pose_refine.restype = C.c_int
    lib().orc_pose_refine.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_double, C.c_double, C.c_void_p]
    n = lib().orc_pose_refine(_p(pose), C.addressof(cam), C.addressof(options), _p(wps), _p(obs), len(obs),
                              None if pred is None else _p(pred), float(w_rot), float(w_trans), _p(outl))
    return pose, outl, n
