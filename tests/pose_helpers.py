"""Synthetic 2D-3D match sets for the pose-refinement tests (oracle and GPU)."""
import numpy as np

CAM = (458.654, 457.296, 367.215, 248.375, 47.9)  # fx fy cx cy bf (EuRoC-like)


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


def perturb(rng, pose, rot=0.01, trans=0.03):
    d = random_pose(rng, rot, trans)
    R = quat_to_R(d[:4]) @ quat_to_R(pose[:4])
    t = quat_to_R(d[:4]) @ pose[4:] + d[4:]
    # back to quaternion (w >= 0)
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    q = np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])
    return np.concatenate([q / np.linalg.norm(q), t])


def make_problem(seed, n=300, outlier_frac=0.2, stereo_frac=0.5, noise=0.5, behind=0):
    """Returns dict(pose_gt, pose0, wps [n,3], obs [n] (x,y,depth,weight), is_outlier [n])."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy, bf = CAM
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
    return dict(pose_gt=pose, pose0=perturb(rng, pose), wps=np.ascontiguousarray(wps), obs=obs, is_outlier=is_out)


def pose_error(a, b):
    """(rotation angle [rad], translation distance) between two world->camera poses."""
    Ra, Rb = quat_to_R(a[:4]), quat_to_R(b[:4])
    c = (np.trace(Ra @ Rb.T) - 1) / 2
    return float(np.arccos(np.clip(c, -1, 1))), float(np.linalg.norm(a[4:] - b[4:]))
