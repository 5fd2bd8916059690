"""Seeded synthetic inputs of the benchmark workloads: corner-rich stereo frames,
random descriptor sets and the 20-keyframe x 2000-point local-BA scene.  Pure numpy; used by
tests/ and bench.py (there is no network for datasets)."""
from __future__ import annotations

import numpy as np

SEED = 363456635  # the reference's randomSeed (reference configs/euroc.ini:3)


def _draw_rects(img, rects, shift=None, tex=None):
    """tex[k] = (cell_u, cell_v, phase_u, phase_v, amplitude) or None: a random patchwork in the rectangle's own (u, v)
    coordinates, so it moves with the rectangle (the interior corners of a textured rectangle are the same surface points in
    both images of a stereo pair / in every frame of a sequence)."""
    h, w = img.shape
    for k, (cx, cy, hw, hh, ang, g) in enumerate(rects):
        if shift is not None:
            cx = cx - shift[k]
        r = int(np.ceil(np.hypot(hw, hh))) + 1
        x0, x1 = max(0, int(cx) - r), min(w, int(cx) + r + 1)
        y0, y1 = max(0, int(cy) - r), min(h, int(cy) + r + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        yy, xx = np.mgrid[y0:y1, x0:x1]
        c, s = np.cos(ang), np.sin(ang)
        u = (xx - cx) * c + (yy - cy) * s
        v = -(xx - cx) * s + (yy - cy) * c
        m = (np.abs(u) <= hw) & (np.abs(v) <= hh)
        if tex is None or tex[k] is None:
            img[y0:y1, x0:x1][m] = g
        else:
            pu, pv, fu, fv, amp = tex[k]
            # a random (not a periodic) pattern: every patch of the rectangle gets its own gray level from an integer hash of its
            # cell coordinates, so the corners inside a rectangle do not look alike (a checker fails the matchers' ratio tests)
            iu = np.floor((u + fu) / pu).astype(np.int64)
            iv = np.floor((v + fv) / pv).astype(np.int64)
            hsh = (iu * 73856093) ^ (iv * 19349663) ^ (k * 83492791)
            hsh = (hsh ^ (hsh >> 13)) * 1274126177
            lvl = ((hsh >> 7) & 1023).astype(np.float64) / 1023.0
            img[y0:y1, x0:x1][m] = (g + amp * (2.0 * lvl - 1.0))[m]


def _textures(rng, n_rects, frac):
    """Checker parameters for a fraction `frac` of the rectangles (own generator: the rectangles themselves are drawn exactly
    as before)."""
    tex = []
    for _ in range(n_rects):
        pu, pv = rng.uniform(7, 18), rng.uniform(7, 18)
        fu, fv = rng.uniform(0, 18), rng.uniform(0, 18)
        amp = rng.uniform(12, 45)
        tex.append((pu, pv, fu, fv, amp) if rng.random() < frac else None)
    return tex


BACKGROUND_DISPARITY = 3.0  # of the textured far wall behind the rectangles
OBJECTS = 0.05  # share of the n_rects rectangles that the textured scenes keep as objects in front of the wall (400 -> 20)
TEXTURE = 0.6  # fraction of the rectangles that carry a checker texture (0 = the round-1/2 images: flat rectangles, painted in draw order)


def stereo_frame(index: int = 0, width: int = 752, height: int = 480, n_rects: int = 400, seed: int = SEED, texture: float = None):
    """Returns (left, right) uint8 images: low-frequency gradient + random (rotated) rectangles +
    Gaussian noise sigma=2; the right image shifts every rectangle by its own disparity in [2,60].
    texture > 0 (default TEXTURE): that fraction of the rectangles carries a checker pattern that moves with the rectangle, and
    the rectangles are painted far to near (consistent occlusion in both images) -- interior corners are then the same
    surface points left and right, so a realistic share of the keypoints has a stereo match (round 2's flat rectangles in
    draw order gave ~12 %: most of their corners are occlusion corners, different in the two images)."""
    texture = TEXTURE if texture is None else texture
    rng = np.random.default_rng(seed + index)
    yy, xx = np.mgrid[0:height, 0:width]
    base = 96.0 + 40.0 * np.sin(xx / width * 2.1 + 0.3) + 30.0 * np.cos(yy / height * 1.7)
    rects = []
    for _ in range(n_rects):
        cx, cy = rng.uniform(0, width), rng.uniform(0, height)
        hw, hh = rng.uniform(4, 40), rng.uniform(4, 40)
        ang = rng.uniform(0, np.pi) if rng.random() < 0.5 else 0.0
        g = rng.uniform(10, 245)
        rects.append((cx, cy, hw, hh, ang, g))
    disp = rng.uniform(2, 60, n_rects)
    left = base.copy()
    right = base.copy()
    tex = None
    if texture > 0:
        trng = np.random.default_rng([seed + index, 77])
        tex = _textures(trng, n_rects, texture)
        order = np.argsort(disp)  # far first, near painted over them
        # few objects in front of the wall: the 400 flat rectangles of rounds 1/2 cover the image twice over and nearly every
        # 31-pixel descriptor patch straddles a depth discontinuity (measured with the oracle, 1000 features: wall only 53 % of
        # the left keypoints get a stereo match, + 16 rectangles 46 %, + 25: 36 %, + 40: 28 %, + 400: 15 %)
        order = order[np.sort(np.random.default_rng([seed + index, 78]).permutation(n_rects)[: int(round(n_rects * OBJECTS))])] \
            if n_rects > 0 else order
        rects, tex, disp = [rects[i] for i in order], [tex[i] for i in order], disp[order]
        # the far wall: the whole image is a textured plane of disparity BACKGROUND_DISPARITY (drawn as one big rectangle)
        wall = [(width / 2.0, height / 2.0, float(width), float(height), 0.0, 0.0)]
        wtex = [(trng.uniform(9, 16), trng.uniform(9, 16), trng.uniform(0, 16), trng.uniform(0, 16), trng.uniform(10, 22))]
        lw, rw = np.zeros_like(left), np.zeros_like(right)
        _draw_rects(lw, wall, None, wtex)
        _draw_rects(rw, wall, [BACKGROUND_DISPARITY], wtex)
        left += lw
        right += rw
    _draw_rects(left, rects, None, tex)
    _draw_rects(right, rects, disp, tex)
    left += rng.normal(0, 2.0, left.shape)
    right += rng.normal(0, 2.0, right.shape)
    return (np.clip(np.rint(left), 0, 255).astype(np.uint8), np.clip(np.rint(right), 0, 255).astype(np.uint8))


def _stereo_frame_job(a):
    return stereo_frame(*a)


def _ba_scene_job(kw):
    return ba_scene(**kw)[0]


def _fork_is_safe() -> bool:
    import sys

    import os

    # under rocprofv3 every forked child carries the profiler's signal handlers and finalisation (a counter-collection pass
    # hung on them): generate serially there
    if any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return False
    t = sys.modules.get("torch")
    lib = sys.modules.get(__package__ + "._lib")
    return (t is None or not t.cuda.is_initialized()) and (lib is None or lib._lib is None)  # libsnake_hip.so not loaded yet


def _pool_map(fn, jobs, workers):
    """Seeded generators are pure functions of their arguments, so a batch of them is built by `workers` forked
    processes -- only while this process holds no HIP context (a forked child of a process with one is undefined);
    otherwise, or where processes cannot be created, a serial loop gives the same data."""
    workers = max(1, min(int(workers), len(jobs)))
    if workers > 1 and _fork_is_safe():
        try:
            import multiprocessing as mp

            pool = mp.get_context("fork").Pool(workers)
            try:
                out = pool.map(fn, jobs, chunksize=max(1, len(jobs) // (4 * workers)))
            finally:
                pool.close()  # the workers leave by themselves (no SIGTERM as with terminate())
                pool.join()
            return out
        except OSError:
            pass
    return [fn(j) for j in jobs]


def default_workers() -> int:
    import os

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    return max(1, min(16, n // max(1, world)))


def stereo_frames(indices, width: int = 752, height: int = 480, n_rects: int = 400, seed: int = SEED, workers=None, texture: float = None):
    """[stereo_frame(i, ...) for i in indices], generated by several processes."""
    jobs = [(int(i), width, height, n_rects, seed, texture) for i in indices]
    return _pool_map(_stereo_frame_job, jobs, default_workers() if workers is None else workers)


def ba_scenes(seeds, workers=None, **kw):
    """[ba_scene(seed=s, **kw)[0] for s in seeds], generated by several processes."""
    jobs = [dict(kw, seed=int(s)) for s in seeds]
    return _pool_map(_ba_scene_job, jobs, default_workers() if workers is None else workers)


def sequence_frames(sequence: int, n_frames: int, width: int = 752, height: int = 480, n_rects: int = 400, step: float = 0.05,
                    seed: int = SEED, texture: float = None):
    """A synthetic stereo SEQUENCE (BASELINE.json config 5): the scene of `stereo_frame` seen by a rig that moves `step`
    baselines to the right per frame.  A rectangle of disparity d (depth bf / d) therefore moves d * step pixels to the
    left per frame in both images -- physically consistent with a pure x-translation, so a tracker that back-projects with
    the same bf recovers it: ground-truth camera position of frame t = (t * step * baseline, 0, 0).
    Yields (left, right) uint8 images."""
    rng = np.random.default_rng(seed + 7919 * (sequence + 1))
    yy, xx = np.mgrid[0:height, 0:width]
    base = 96.0 + 40.0 * np.sin(xx / width * 2.1 + 0.3) + 30.0 * np.cos(yy / height * 1.7)
    rects = []
    for _ in range(n_rects):
        # a little wider than the image so that content enters as the rig moves
        cx, cy = rng.uniform(0, width * 1.15), rng.uniform(0, height)
        hw, hh = rng.uniform(4, 40), rng.uniform(4, 40)
        ang = rng.uniform(0, np.pi) if rng.random() < 0.5 else 0.0
        g = rng.uniform(10, 245)
        rects.append((cx, cy, hw, hh, ang, g))
    disp = rng.uniform(2, 60, n_rects)
    order = np.argsort(disp)  # far rectangles first, near ones painted over them (consistent occlusion)
    rects = [rects[i] for i in order]
    disp = disp[order]
    texture = TEXTURE if texture is None else texture
    tex = wall = wtex = None
    if texture > 0:  # as stereo_frame: a textured far wall and a few textured objects in front of it
        trng = np.random.default_rng([seed, sequence, 77])
        tex = _textures(trng, n_rects, texture)
        tex = [tex[i] for i in order]
        keep = np.sort(np.random.default_rng([seed, sequence, 78]).permutation(n_rects)[: int(round(n_rects * OBJECTS))])
        rects, tex, disp = [rects[i] for i in keep], [tex[i] for i in keep], disp[keep]
        wall = [(width * 0.75, height / 2.0, float(width) * 1.5, float(height), 0.0, 0.0)]  # wide enough for the whole run
        wtex = [(trng.uniform(9, 16), trng.uniform(9, 16), trng.uniform(0, 16), trng.uniform(0, 16), trng.uniform(10, 22))]
    for t in range(n_frames):
        nrng = np.random.default_rng([seed, sequence, t])
        left, right = base.copy(), base.copy()
        if wall is not None:
            lw, rw = np.zeros_like(left), np.zeros_like(right)
            _draw_rects(lw, wall, [BACKGROUND_DISPARITY * (step * t)], wtex)
            _draw_rects(rw, wall, [BACKGROUND_DISPARITY * (step * t + 1.0)], wtex)
            left += lw
            right += rw
        _draw_rects(left, rects, disp * (step * t), tex)
        _draw_rects(right, rects, disp * (step * t + 1.0), tex)
        left += nrng.normal(0, 2.0, left.shape)
        right += nrng.normal(0, 2.0, right.shape)
        yield (np.clip(np.rint(left), 0, 255).astype(np.uint8), np.clip(np.rint(right), 0, 255).astype(np.uint8))


def _sequence_job(a):
    return list(sequence_frames(*a))


def sequences(ids, n_frames: int, width: int = 752, height: int = 480, n_rects: int = 400, workers=None):
    """[list(sequence_frames(i, n_frames, ...)) for i in ids], generated by several processes (one job per sequence)."""
    jobs = [(int(i), int(n_frames), width, height, n_rects) for i in ids]
    return _pool_map(_sequence_job, jobs, default_workers() if workers is None else workers)


def random_descriptors(n: int, seed: int = SEED):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)


def _quat_from_R(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    x = (R[2, 1] - R[1, 2]) / (4 * w)
    y = (R[0, 2] - R[2, 0]) / (4 * w)
    z = (R[1, 0] - R[0, 1]) / (4 * w)
    return np.array([x, y, z, w])


def _rot(axis, ang):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def ba_scene(n_kf: int = 20, n_pt: int = 2000, obs_per_pt: int = 8, seed: int = SEED, pixel_noise: float = 0.5,
             perturb: bool = True, stereo_frac: float = 0.5, n_fixed: int = 1, outlier_frac: float = 0.0):
    """The synthetic local-BA problem of SURVEY.md §8(d): cameras on an arc looking at a box of
    points, every point seen by `obs_per_pt` consecutive cameras (round-robin windows), EuRoC-like
    intrinsics, half of the observations stereo, weights 1/1.44^octave, noisy pixels, perturbed
    initial poses / points, first camera(s) constant.  Returns (scene, ground_truth)."""
    rng = np.random.default_rng(seed)
    K = (458.654, 457.296, 367.215, 248.375)
    bf = 47.9
    poses_R, poses_t = [], []
    for i in range(n_kf):
        a = (i / max(1, n_kf - 1) - 0.5) * 0.9  # arc angle
        c = np.array([10.0 * np.sin(a), 0.15 * np.sin(3 * a), 10.0 * (1 - np.cos(a)) - 1.0])  # camera centre
        Rwc = _rot(np.array([0.0, 1.0, 0.0]), -a * 0.8)  # camera-to-world: look roughly at the box
        Rcw = Rwc.T
        poses_R.append(Rcw)
        poses_t.append(-Rcw @ c)
    pts = np.stack([rng.uniform(-4, 4, n_pt), rng.uniform(-2, 2, n_pt), rng.uniform(5, 9, n_pt)], axis=1)
    obs_img, obs_pt, obs_uv, obs_depth, obs_w = [], [], [], [], []
    for p in range(n_pt):
        first = p % n_kf
        for k in range(min(obs_per_pt, n_kf)):
            i = (first + k) % n_kf
            pc = poses_R[i] @ pts[p] + poses_t[i]
            if pc[2] <= 0.5:
                continue
            u = K[0] * pc[0] / pc[2] + K[2]
            v = K[1] * pc[1] / pc[2] + K[3]
            octave = int(rng.integers(0, 4))
            stereo = rng.random() < stereo_frac
            nu, nv, nd = rng.normal(0, pixel_noise, 3) if pixel_noise > 0 else (0.0, 0.0, 0.0)
            if outlier_frac > 0 and rng.random() < outlier_frac:
                nu += rng.uniform(15, 40) * rng.choice([-1, 1])
            obs_img.append(i)
            obs_pt.append(p)
            obs_uv.append((u + nu, v + nv))
            # depth consistent with a noisy right-image coordinate u_r = u - bf/z
            if stereo:
                ur = (u - bf / pc[2]) + nd
                disp = max((u + nu) - ur, 1e-3)
                obs_depth.append(bf / disp)
            else:
                obs_depth.append(-1.0)
            obs_w.append(1.0 / 1.44**octave)
    gt_pose = np.array([np.concatenate([_quat_from_R(R), t]) for R, t in zip(poses_R, poses_t)])
    pose0, pt0 = gt_pose.copy(), pts.copy()
    if perturb:
        for i in range(n_fixed, n_kf):
            dR = _rot(rng.normal(size=3), np.radians(0.5) * rng.uniform(0.5, 1.0))
            R = dR @ poses_R[i]
            t = dR @ poses_t[i] + rng.normal(0, 0.02 / np.sqrt(3), 3)
            pose0[i] = np.concatenate([_quat_from_R(R), t])
        pt0 = pts + rng.normal(0, 0.05 / np.sqrt(3), pts.shape)
    img_const = np.zeros(n_kf, np.uint8)
    img_const[:n_fixed] = 1
    scene = dict(pose=pose0, img_const=img_const, pt=pt0, pt_const=np.zeros(n_pt, np.uint8),
                 obs_img=np.array(obs_img, np.int32), obs_pt=np.array(obs_pt, np.int32), obs_uv=np.array(obs_uv, np.float64),
                 obs_depth=np.array(obs_depth, np.float64), obs_weight=np.array(obs_w, np.float64), K=K, bf=bf)
    return scene, dict(pose=gt_pose, pt=pts)


def ba_add_rpcs(scene, gt, seed: int = 0, weight_rotation: float = 30.0, weight_translation: float = 8.0,
                noise_rot: float = 2e-3, noise_trans: float = 5e-3):
    """Relative pose constraints between consecutive keyframes (what MakeLocalScene adds from the IMU
    pre-integration, reference LocalBundleAdjustment.cpp:294-346): rel_pose = T_{i+1} T_i^-1 of the ground
    truth with a little noise.  Adds scene["rpc"] in place and returns the scene."""
    rng = np.random.default_rng(seed)
    dt = np.dtype([("img1", "<i4"), ("img2", "<i4"), ("rel_pose", "<f8", 7), ("weight_rotation", "<f8"),
                   ("weight_translation", "<f8")])
    n = len(gt["pose"])
    rp = np.zeros(max(n - 1, 0), dt)
    for i in range(n - 1):
        R1, t1 = quat_to_R(gt["pose"][i][:4]), gt["pose"][i][4:]
        R2, t2 = quat_to_R(gt["pose"][i + 1][:4]), gt["pose"][i + 1][4:]
        R21 = _rot(rng.normal(size=3), noise_rot * rng.uniform(0.2, 1.0)) @ R2 @ R1.T
        t21 = t2 - R2 @ R1.T @ t1 + rng.normal(0, noise_trans / np.sqrt(3), 3)
        rp[i]["img1"], rp[i]["img2"] = i, i + 1
        rp[i]["rel_pose"] = np.concatenate([_quat_from_R(R21), t21])
        rp[i]["weight_rotation"], rp[i]["weight_translation"] = weight_rotation, weight_translation
    scene["rpc"] = rp
    return scene


# ------------------------------------------------------------------ pose refinement ------------
POSE_CAM = (458.654, 457.296, 367.215, 248.375, 47.9)  # fx fy cx cy bf (EuRoC-like)


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def random_pose(rng, rot=0.3, trans=1.0):
    w = rng.normal(size=3)
    w *= rot * rng.uniform(0.2, 1.0) / np.linalg.norm(w)
    th = np.linalg.norm(w)
    q = np.concatenate([np.sin(th / 2) * w / th, [np.cos(th / 2)]])
    return np.concatenate([q, rng.uniform(-trans, trans, 3)])


def perturb_pose(rng, pose, rot=0.01, trans=0.03):
    d = random_pose(rng, rot, trans)
    R = quat_to_R(d[:4]) @ quat_to_R(pose[:4])
    t = quat_to_R(d[:4]) @ pose[4:] + d[4:]
    # back to quaternion (w >= 0)
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    q = np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])
    return np.concatenate([q / np.linalg.norm(q), t])


def pose_problem(seed, n=300, outlier_frac=0.2, stereo_frac=0.5, noise=0.5, behind=0):
    """Returns dict(pose_gt, pose0, wps [n,3], obs [n] (x,y,depth,weight), is_outlier [n])."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy, bf = POSE_CAM
    pose = random_pose(rng)
    R, t = quat_to_R(pose[:4]), pose[4:]
    # points in the camera frame, then to world
    z = rng.uniform(2.0, 12.0, n)
    u = rng.uniform(30, 720, n)
    v = rng.uniform(30, 450, n)
    pc = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
    wps = (pc - t) @ R  # R^T (pc - t)
    octave = rng.integers(0, 4, n)
    weight = 1.0 / 1.2 ** octave
    obs = np.zeros(n, [("x", "f8"), ("y", "f8"), ("depth", "f8"), ("weight", "f8")])
    obs["x"] = u + rng.normal(0, noise, n) * 1.2 ** octave
    obs["y"] = v + rng.normal(0, noise, n) * 1.2 ** octave
    stereo = rng.random(n) < stereo_frac
    obs["depth"] = np.where(stereo, z * (1 + rng.normal(0, 0.002, n)), -1.0)
    obs["weight"] = weight
    is_out = rng.random(n) < outlier_frac
    obs["x"][is_out] += rng.choice([-1, 1], is_out.sum()) * rng.uniform(15, 80, is_out.sum())
    obs["y"][is_out] += rng.choice([-1, 1], is_out.sum()) * rng.uniform(15, 80, is_out.sum())
    if behind:
        wps[:behind] = (np.array([0.0, 0.0, -3.0]) - t) @ R  # behind the camera
        is_out[:behind] = True
    return dict(pose_gt=pose, pose0=perturb_pose(rng, pose), wps=np.ascontiguousarray(wps), obs=obs, is_outlier=is_out)
