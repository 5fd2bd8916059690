#!/opt/conda/bin/python3.9
"""Round 4: two more pieces of the ORB oracle pinned against scikit-image 0.18.3 (independent code; only /opt/conda's python3.9
has it in the build container -- the fixture travels, skimage does not).  VERDICT round 3, item 6.

 (iii) the intensity-centroid orientation: for keypoints of the golden image, skimage.feature.corner_orientations(level,
       corners, OFAST_MASK) -- atan2(m01, m10) over skimage's own radius-15 disc -- and the two moments summed over that mask
       with plain numpy.  tests/test_oracle_pin.py compares the oracle's integer moments (equal) and its fastAtan2 angle (OpenCV's
       polynomial: within 0.02 degrees of the exact angle).
 (iv)  the steering of the 256 BRIEF tests: skimage.feature.orb_cy._orb_loop(plane, keypoints, orientations) evaluates
       plane[r + round(sin a * x + cos a * y), c + round(cos a * x - sin a * y)] pairs on the plane it is GIVEN, so on an
       unblurred plane and with forced angles it isolates pattern order, rotation direction, x / y roles and bit order from
       the blur and from the orientation.  The oracle's descriptor on the same raw plane must give the same 256 bits except where
       a steered coordinate sits on a rounding boundary (skimage: double sin / cos, round half away from zero; the oracle: its
       float sincos, half to even).

    /opt/conda/bin/python3.9 tests/golden/pin_against_skimage2.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

import skimage  # noqa: E402
from skimage.feature import corner_orientations  # noqa: E402
from skimage.feature.orb import OFAST_MASK  # noqa: E402
from skimage.feature.orb_cy import _orb_loop  # noqa: E402

OUT = Path(__file__).resolve().parent


def main():
    g = np.load(OUT / "skimage_pin.npz")
    img = g["img"]
    rng = np.random.default_rng(20260928)
    h, w = img.shape
    n = 96
    xs = rng.integers(20, w - 20, n)
    ys = rng.integers(20, h - 20, n)
    corners = np.stack([ys, xs], 1).astype(np.intp)  # skimage: (row, col)
    f = np.ascontiguousarray(img, np.float64)
    ori = corner_orientations(f, corners, OFAST_MASK)
    # moments over skimage's mask, plain numpy (x = column offset, y = row offset)
    dy, dx = np.mgrid[-15:16, -15:16]
    m10 = np.array([(f[y - 15:y + 16, x - 15:x + 16] * OFAST_MASK * dx).sum() for x, y in zip(xs, ys)])
    m01 = np.array([(f[y - 15:y + 16, x - 15:x + 16] * OFAST_MASK * dy).sum() for x, y in zip(xs, ys)])
    assert np.allclose(np.arctan2(m01, m10), ori, atol=1e-12)
    angles_deg = np.array([0.0, 37.0, 90.0, 123.456, 180.0, 200.25, 270.0, 311.0, 359.5])
    bits = np.zeros((len(angles_deg), n, 256), bool)
    for a, deg in enumerate(angles_deg):
        bits[a] = _orb_loop(f, corners, np.full(n, np.deg2rad(deg))) != 0
    np.savez_compressed(OUT / "skimage_pin2.npz", xs=xs.astype(np.int32), ys=ys.astype(np.int32), m10=m10.astype(np.int64), m01=m01.astype(np.int64),
                        orientation_rad=ori, angles_deg=angles_deg, bits=np.packbits(bits, axis=-1), mask=OFAST_MASK.astype(np.uint8),
                        skimage_version=np.array(skimage.__version__))
    print(f"wrote {OUT / 'skimage_pin2.npz'}: {n} keypoints, {len(angles_deg)} forced angles, scikit-image {skimage.__version__}")


if __name__ == "__main__":
    main()
