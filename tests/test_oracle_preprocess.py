"""CPU: pin the undistort / rectify oracle (PARITY UNPINNED vs saiga; see oracle/preprocess_oracle.c)."""
import numpy as np

EUROC_K = (458.654, 457.296, 367.215, 248.375)
EUROC_D = (-0.28340811, 0.07395907, 0.0, 0.0, 0.0, 0.0, 0.00019359, 1.76187114e-05)  # k1 k2 k3 k4 k5 k6 p1 p2


def distort(D, x, y):
    k1, k2, k3, k4, k5, k6, p1, p2 = D
    r2 = x * x + y * y
    rad = (1 + k1 * r2 + k2 * r2**2 + k3 * r2**3) / (1 + k4 * r2 + k5 * r2**2 + k6 * r2**3)
    return x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x), y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y


def make_kps(orc, n, seed=0):
    rng = np.random.default_rng(seed)
    k = np.zeros(n, orc.KEYPOINT)
    k["x"] = rng.uniform(0, 752, n).astype(np.float32)
    k["y"] = rng.uniform(0, 480, n).astype(np.float32)
    k["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    k["octave"] = rng.integers(0, 4, n)
    return k


def test_identity_rectification_is_exact_cast(orc):
    k = make_kps(orc, 100)
    r = orc.rectification((1.0, 1.0, 0.0, 0.0))
    out, norm = orc.rectify(r, k)
    assert np.array_equal(out["x"], k["x"].astype(np.float64)) and np.array_equal(out["y"], k["y"].astype(np.float64))
    assert np.array_equal(out["angle"], k["angle"]) and np.array_equal(out["octave"], k["octave"])
    assert np.array_equal(norm[:, 0], out["x"])


def test_undistort_inverts_the_distortion_model(orc):
    k = make_kps(orc, 400, 1)
    r = orc.rectification(EUROC_K, EUROC_D)
    out, norm = orc.rectify(r, k)
    fx, fy, cx, cy = EUROC_K
    xd, yd = distort(EUROC_D, norm[:, 0], norm[:, 1])
    assert np.abs(xd - (k["x"].astype(np.float64) - cx) / fx).max() < 1e-9
    assert np.abs(yd - (k["y"].astype(np.float64) - cy) / fy).max() < 1e-9
    assert np.abs(out["x"] - (norm[:, 0] * fx + cx)).max() < 1e-12


def test_rotation_and_new_intrinsics(orc):
    k = make_kps(orc, 50, 2)
    a = 0.01
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    Kd = (435.2, 435.2, 367.4, 252.2)
    r = orc.rectification(EUROC_K, None, R, Kd)
    out, norm = orc.rectify(r, k)
    p = np.stack([(k["x"].astype(np.float64) - EUROC_K[2]) / EUROC_K[0], (k["y"].astype(np.float64) - EUROC_K[3]) / EUROC_K[1],
                  np.ones(50)])
    q = R @ p
    assert np.abs(norm[:, 0] - q[0] / q[2]).max() < 1e-14
    assert np.abs(out["y"] - (q[1] / q[2] * Kd[1] + Kd[3])).max() < 1e-10
