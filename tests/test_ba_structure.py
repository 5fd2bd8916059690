"""The algebra behind the batch BA kernels of round 5 (snake_slam_amd/csrc/ba.hip: obs_core, obs_moments; DESIGN.md section 4, last
subsection), checked in numpy: with Xc = R p + t and the rows m_k = w * d proj_k / d Xc the observation Jacobians of obs_linearize are
J_c,k = [m_k | Xc x m_k] and J_p,k = m_k^T R, and every product the solver takes of them is a function of M = sum_k m_k m_k^T,
h = sum_k m_k r_k, Xc and R.  The kernels use the right-hand sides; the oracle (and the single-window kernels) the left-hand sides.
No GPU, no library: this pins the identities themselves."""
import numpy as np


def _rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _linearize(R, t, p, K, bf, w, stereo):
    """obs_linearize of ba.hip / the oracle's observation model: J_c (3 x 6), J_p (3 x 3), rows beyond `dim` zero."""
    fx, fy = K[0], K[1]
    X, Y, Z = R @ p + t
    iz, iz2 = 1.0 / Z, 1.0 / (Z * Z)
    P = np.array([[fx * iz, 0.0, -fx * X * iz2], [0.0, fy * iz, -fy * Y * iz2], [fx * iz, 0.0, -fx * X * iz2 + bf * iz2]])
    Jc, Jp = np.zeros((3, 6)), np.zeros((3, 3))
    for k in range(3 if stereo else 2):
        a, b, c = w * P[k]
        Jc[k] = [a, b, c, -b * Z + c * Y, a * Z - c * X, -a * Y + b * X]
        Jp[k] = np.array([a, b, c]) @ R
    return Jc, Jp, np.array([X, Y, Z]), w * P * (np.arange(3) < (3 if stereo else 2))[:, None]


def test_products_of_the_observation_jacobians_by_structure():
    rng = np.random.default_rng(7)
    K, bf = (458.654, 457.296, 367.215, 248.375), 47.9
    for case in range(200):
        R, t = _rot(rng), rng.normal(size=3)
        p = R.T @ (np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(2, 10)]) - t)  # in front of the camera
        w, stereo = 1.0 / 1.2 ** rng.integers(0, 4), bool(case % 2)
        Jc, Jp, Xc, m = _linearize(R, t, p, K, bf, w, stereo)
        r = rng.normal(size=3) * (np.arange(3) < (3 if stereo else 2))
        s2 = rng.uniform(0.2, 1.0)  # the squared IRLS scale
        # the structure itself
        for k in range(3):
            assert np.allclose(Jc[k, :3], m[k]) and np.allclose(Jc[k, 3:], np.cross(Xc, m[k])) and np.allclose(Jp[k], m[k] @ R)
        assert m[0, 1] == 0 and m[1, 0] == 0 and m[2, 1] == 0 and (not stereo or m[2, 0] == m[0, 0])
        M, h = s2 * (m.T @ m), s2 * (m.T @ r)
        assert M[0, 1] == 0
        cross = lambda v: np.cross(Xc, v)  # noqa: E731
        # schur_fused: W = J_c^T J_p, V = J_p^T J_p, b_p = -J_p^T r
        N = M @ R
        W = np.vstack([N, np.stack([cross(N[:, b]) for b in range(3)], axis=1)])
        assert np.allclose(s2 * Jc.T @ Jp, W, rtol=1e-12, atol=1e-12)
        assert np.allclose(s2 * Jp.T @ Jp, R.T @ N, rtol=1e-12, atol=1e-12)
        assert np.allclose(-s2 * Jp.T @ r, -R.T @ h, rtol=1e-12, atol=1e-12)
        # cam_pass: U = J_c^T J_c, b_c = -J_c^T r, Y b_p = J_c^T (J_p v)
        T = np.stack([cross(M[:, i]) for i in range(3)])            # T[i] = Xc x M[:, i]
        B = np.stack([cross(T[:, b]) for b in range(3)], axis=1)    # Xc x (the columns of T)
        U = np.block([[M, T], [T.T, B]])
        assert np.allclose(s2 * Jc.T @ Jc, U, rtol=1e-11, atol=1e-11)
        assert np.allclose(-s2 * Jc.T @ r, -np.concatenate([h, cross(h)]), rtol=1e-12, atol=1e-12)
        v = rng.normal(size=3)
        f = M @ (R @ v)
        assert np.allclose(s2 * Jc.T @ (Jp @ v), np.concatenate([f, cross(f)]), rtol=1e-11, atol=1e-11)
        # update_cost: J_p^T (J_c dc) = R^T M (dt + dr x Xc)
        dc = rng.normal(size=6) * 0.01
        assert np.allclose(s2 * Jp.T @ (Jc @ dc), R.T @ (M @ (dc[:3] + np.cross(dc[3:], Xc))), rtol=1e-11, atol=1e-13)


def test_transpose_and_add_index_map():
    """wave_reduce33 (ba.hip): after the six halving steps lane l holds value 17 b5 + 9 b4 + 5 b3 + 3 b2 + 2 b1 + b0 when every partial
    index is inside its (odd) split -- the valid lanes must cover 0..32 exactly once."""
    seen = []
    for lane in range(64):
        b = [(lane >> k) & 1 for k in range(6)]
        i3 = 2 * b[1] + b[0]
        i5 = 3 * b[2] + i3
        i9 = 5 * b[3] + i5
        i17 = 9 * b[4] + i9
        idx = 17 * b[5] + i17
        if i3 < 3 and i5 < 5 and i9 < 9 and i17 < 17 and idx < 33:
            seen.append(idx)
    assert sorted(seen) == list(range(33))
