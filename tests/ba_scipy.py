"""The optimum of a "snk-ba v1" scene from an INDEPENDENT solver: scipy.optimize.least_squares (trust-region reflective, finite-
difference Jacobian) on the robust cost written from the definition (tests/ba_numpy.py's observation model, vectorised) -- no line of
the oracle or of the kernels, another parameterisation (rotation vectors composed on the left of the start rotation), another
algorithm.  VERDICT round 3, item 6: the converged LM of the oracle / of the HIP solver must land on the same minimiser.

Huber on the NORM of a 2- / 3-vector residual (not per component) is expressed for least_squares as the residual vector scaled by
sqrt(rho(s) / s), s = ||e||^2: the sum of squares is then exactly sum rho(s), and its minimiser is the robust optimum."""
import numpy as np
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation


def _rodrigues(w):
    """exp of rotation vectors [n, 3] -> [n, 3, 3]; written with plain arithmetic so that it also takes COMPLEX input (the complex-step
    Jacobian of least_squares, jac="cs": derivatives exact to round-off, no finite-difference floor)."""
    th2 = (w * w).sum(1)
    small = th2.real < 1e-16
    th2s = np.where(small, 1.0, th2)
    th = np.sqrt(th2s)
    a = np.where(small, 1.0 - th2 / 6.0, np.sin(th) / th)
    b = np.where(small, 0.5 - th2 / 24.0, (1.0 - np.cos(th)) / th2s)
    K = np.zeros((len(w), 3, 3), w.dtype)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -w[:, 2], w[:, 1], w[:, 2], -w[:, 0], -w[:, 1], w[:, 0]
    return np.eye(3)[None] + a[:, None, None] * K + b[:, None, None] * (K @ K)


def _unpack(scene, x, free_img, free_pt):
    R0 = Rotation.from_quat(scene["pose"][:, :4]).as_matrix()
    nf = len(free_img)
    d = x[: 6 * nf].reshape(nf, 6)
    R = R0.astype(x.dtype)
    t = np.array(scene["pose"][:, 4:], np.float64).astype(x.dtype)
    R[free_img] = _rodrigues(d[:, :3]) @ R0[free_img]
    t[free_img] = t[free_img] + d[:, 3:]
    pt = np.array(scene["pt"], np.float64).astype(x.dtype)
    pt[free_pt] = x[6 * nf:].reshape(-1, 3)
    return R, t, pt


def _residuals(scene, R, t, pt, keep, huber_mono, huber_stereo):
    fx, fy, cx, cy = scene["K"]
    bf = float(scene["bf"])
    i, p = scene["obs_img"][keep], scene["obs_pt"][keep]
    pc = np.einsum("nij,nj->ni", R[i], pt[p]) + t[i]
    u = fx * pc[:, 0] / pc[:, 2] + cx
    v = fy * pc[:, 1] / pc[:, 2] + cy
    w = scene["obs_weight"][keep]
    uv, d = scene["obs_uv"][keep], scene["obs_depth"][keep]
    stereo = d > 0
    e = np.stack([w * (u - uv[:, 0]), w * (v - uv[:, 1]), np.where(stereo, w * ((u - bf / pc[:, 2]) - (uv[:, 0] - bf / np.where(stereo, d, 1.0))), 0.0)], 1)
    s = (e * e).sum(1)
    th = np.where(stereo, huber_stereo, huber_mono)
    ss = np.where(s.real > 1e-300, s, 1.0)  # complex-safe: branch on the real part
    rho = np.where(s.real <= th * th, s, 2 * th * np.sqrt(ss) - th * th)
    scale = np.where(s.real > 1e-300, np.sqrt(rho / ss), 1.0)
    return (e * scale[:, None]).ravel(), float(rho.real.sum())


def optimum(scene, huber_mono=2.1, huber_stereo=2.3, outlier=None):
    """Returns (R [n_img, 3, 3], t [n_img, 3], pt [n_pt, 3], cost) at the minimiser of the robust cost; constant cameras / points held."""
    img_const, pt_const = np.asarray(scene["img_const"]) != 0, np.asarray(scene["pt_const"]) != 0
    free_img, free_pt = np.nonzero(~img_const)[0], np.nonzero(~pt_const)[0]
    oi, op = np.asarray(scene["obs_img"]), np.asarray(scene["obs_pt"])
    keep = (oi >= 0) & (oi < len(img_const)) & (op >= 0) & (op < len(pt_const))
    keep[keep] &= ~(img_const[oi[keep]] & pt_const[op[keep]])
    if outlier is not None:
        keep &= np.asarray(outlier) == 0
    x0 = np.concatenate([np.zeros(6 * len(free_img)), np.asarray(scene["pt"], np.float64)[free_pt].ravel()])

    def fun(x):
        return _residuals(scene, *_unpack(scene, x, free_img, free_pt), keep, huber_mono, huber_stereo)[0]

    # dense complex-step Jacobian (exact to round-off) + exact trust-region steps (a few hundred unknowns); restarted from the reached
    # point until a run no longer moves
    res = None
    for _ in range(4):
        res = least_squares(fun, x0 if res is None else res.x, jac="cs", method="trf", tr_solver="exact", ftol=1e-15, xtol=1e-15,
                            gtol=1e-15, max_nfev=200, x_scale=1.0)
    R, t, pt = _unpack(scene, res.x, free_img, free_pt)
    return R, t, pt, _residuals(scene, R, t, pt, keep, huber_mono, huber_stereo)[1], res
