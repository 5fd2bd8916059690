"""Independent numpy restatement of the BA observation model / robust cost (snk-ba v1)."""
import numpy as np


def quat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def residuals(scene, pose=None, pt=None, outlier=None):
    """list of per-observation weighted residual vectors (None when skipped)."""
    pose = scene["pose"] if pose is None else pose
    pt = scene["pt"] if pt is None else pt
    fx, fy, cx, cy = scene["K"]
    bf = scene["bf"]
    out = []
    Rs = [quat_R(p[:4]) for p in pose]
    for o in range(len(scene["obs_img"])):
        i, p = scene["obs_img"][o], scene["obs_pt"][o]
        if (outlier is not None and outlier[o]) or (scene["img_const"][i] and scene["pt_const"][p]):
            out.append(None)
            continue
        pc = Rs[i] @ pt[p] + pose[i][4:]
        if pc[2] <= 0:
            out.append(None)
            continue
        u = fx * pc[0] / pc[2] + cx
        v = fy * pc[1] / pc[2] + cy
        w = scene["obs_weight"][o]
        r = [w * (u - scene["obs_uv"][o][0]), w * (v - scene["obs_uv"][o][1])]
        d = scene["obs_depth"][o]
        if d > 0:
            r.append(w * ((u - bf / pc[2]) - (scene["obs_uv"][o][0] - bf / d)))
        out.append(np.array(r))
    return out


def robust_cost(scene, pose=None, pt=None, huber_mono=2.1, huber_stereo=2.3, outlier=None):
    c = 0.0
    for r in residuals(scene, pose, pt, outlier):
        if r is None:
            continue
        s = float(r @ r)
        d = huber_stereo if len(r) == 3 else huber_mono
        c += s if s <= d * d else 2 * d * np.sqrt(s) - d * d
    return c


def se3_exp_update(pose, delta):
    """exp(delta) * pose with scipy's matrix exponential (independent of the closed form)."""
    from scipy.linalg import expm

    v, w = delta[:3], delta[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = v
    E = expm(M)
    T = np.eye(4)
    T[:3, :3] = quat_R(pose[:4])
    T[:3, 3] = pose[4:]
    return E @ T
