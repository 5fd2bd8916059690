"""CPU: pin the undistort / rectify oracle (PARITY UNPINNED vs saiga; see oracle/preprocess_oracle.c)."""
import ctypes as C

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


def test_rgbd_stereo_hand_checkable(orc):
    """Preprocess::ComputeStereoFromRGBD (Preprocess.cpp:79-120) on cases one can do by hand: no distortion, K_depth = K, so the
    depth pixel of a keypoint is (int)(x + 0.5), (int)(y + 0.5); depth 0 -> -1 / -1; outside the image or depth >= 20 -> the
    reference's abort is reported with the index."""
    from oracle.oracle import KP64

    K = (500.0, 500.0, 320.0, 240.0)
    img = np.zeros((480, 640), np.float32)
    img[100, 200], img[101, 200], img[300, 50] = 2.0, 4.0, 0.0
    und = np.zeros(4, KP64)
    und["x"], und["y"] = [200.4, 200.49, 50.0, 199.5], [100.4, 100.5, 300.0, 99.6]
    n, rp, dp = orc.rgbd_stereo(und, K, np.zeros(8), K, 40.0, img)
    # (200, 100) -> 2 m; (200, 101) -> 4 m (100.5 + 0.5 = 101); (50, 300) -> no depth; (200, 100) again (199.5 + 0.5 = 200, 99.6 + 0.5 = 100.1)
    assert n == 3
    assert np.array_equal(dp, np.array([2.0, 4.0, -1.0, 2.0], np.float32))
    assert np.array_equal(rp, np.array([np.float32(200.4 - 20.0), np.float32(200.49 - 10.0), -1.0, np.float32(199.5 - 20.0)], np.float32))
    und["x"][2] = 700.0  # right of the image
    assert orc.rgbd_stereo(und, K, np.zeros(8), K, 40.0, img)[0] == -3
    und["x"][2] = 50.0
    img[300, 50] = 25.0  # SAIGA_ASSERT(depth < 20)
    assert orc.rgbd_stereo(und, K, np.zeros(8), K, 40.0, img)[0] == -3
    # with distortion: the same forward model undistortPointGN inverts -- distort(undistort(p)) = p
    D = np.array([-0.28, 0.07, 0.0, 0.0, 0.0, 0.0, 2e-4, 1e-5])
    for px, py in ((0.31, -0.22), (-0.4, 0.3), (0.0, 0.0)):
        ux, uy = C.c_double(), C.c_double()
        orc.lib().orc_undistort_gn(D.ctypes.data_as(C.c_void_p), C.c_double(px), C.c_double(py), C.byref(ux), C.byref(uy))
        one = np.zeros(1, KP64)
        one["x"], one["y"] = ux.value * K[0] + K[2], uy.value * K[1] + K[3]
        big = np.full((480, 640), 1.0, np.float32)
        big[int(py * K[1] + K[3] + 0.5), int(px * K[0] + K[2] + 0.5)] = 7.0  # the distorted point's pixel, and only that one
        assert orc.rgbd_stereo(one, K, D, K, 40.0, big)[2][0] == 7.0
