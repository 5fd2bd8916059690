"""GPU: rectify / undistort kernel vs the oracle — bit-exact fp64 (fixed operation order)."""
import numpy as np
import pytest

from test_oracle_preprocess import EUROC_D, EUROC_K, make_kps

pytestmark = pytest.mark.gpu


def test_rectify_parity(orc):
    from snake_slam_amd.matcher import Preprocess, Rectification

    pp = Preprocess(0)
    a = 0.013
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    for n, D, Rm, Kd in [(1000, EUROC_D, R, (435.2, 435.2, 367.4, 252.2)), (257, None, None, None), (0, EUROC_D, None, None),
                         (64, (0.1, -0.05, 0.01, 0.02, 0.003, 0.0001, 0.001, -0.002), R, None)]:
        k = make_kps(orc, n, n)
        ro = orc.rectification(EUROC_K, D, Rm, Kd)
        rg = Rectification.make(EUROC_K, D, Rm, Kd)
        want, wn = orc.rectify(ro, k)
        got, gn = pp.rectify(rg, k)
        for f in ("x", "y", "angle", "octave"):
            assert np.array_equal(got[f], want[f]), f
        assert np.array_equal(gn, wn)
    pp.close()
