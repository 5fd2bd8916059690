#!/opt/conda/bin/python3.9
"""Round 5: the Harris response of the ORB oracle ("orb.response" = 1) pinned against scikit-image 0.18.3 / scipy.ndimage
(independent code; only /opt/conda's python3.9 has skimage in the build container -- the fixture travels, skimage does not).

 (v)  skimage.feature.corner._compute_derivatives(level) -- scipy.ndimage.sobel along both axes, the 3 x 3 operator of OpenCV's
      HarrisResponses -- gives Ix, Iy; the 7 x 7 box sums of Ix^2, Iy^2, Ix Iy (scipy.ndimage.convolve with ones) are the three
      integers a, b, c the oracle must reproduce EXACTLY at every sampled point; the response det - k trace^2 (the expression of
      skimage.feature.corner_harris(method='k') on that tensor, k = 0.04) times (1 / (4 * 7 * 255))^4 in double is what the oracle's
      float response must match to float rounding.
 (vi) skimage.feature.corner_harris itself (Gaussian window, sigma = 2 ~ the 7 x 7 box) at the same points: not the same
      number, but it must rank the points almost the same way (Spearman correlation over the FAST corners among them, stored; the test asks for > 0.8).

    /opt/conda/bin/python3.9 tests/golden/pin_against_skimage3.py
"""
from pathlib import Path

import numpy as np
import scipy.ndimage as ndi
import skimage
from scipy.stats import spearmanr
from skimage.feature import corner_harris
from skimage.feature.corner import _compute_derivatives

OUT = Path(__file__).resolve().parent


def main():
    g = np.load(OUT / "skimage_pin.npz")
    img = g["img"]
    h, w = img.shape
    f = img.astype(np.float64)
    dr, dc = _compute_derivatives(f, mode="constant", cval=0)  # d/d row = Iy, d/d col = Ix
    ones = np.ones((7, 7))
    a = ndi.convolve(dc * dc, ones, mode="constant")
    b = ndi.convolve(dr * dr, ones, mode="constant")
    c = ndi.convolve(dr * dc, ones, mode="constant")
    rng = np.random.default_rng(20260929)
    n = 400
    # half of the points anywhere, half on FAST corners of the pinned image (where the extractor evaluates the response)
    fc = g["fast_l0_t20"].astype(np.int64)
    fc = fc[(fc[:, 0] >= 19) & (fc[:, 0] < w - 19) & (fc[:, 1] >= 19) & (fc[:, 1] < h - 19)]
    pick = fc[rng.permutation(len(fc))[: n // 2]]
    xs = np.concatenate([rng.integers(19, w - 19, n - len(pick)), pick[:, 0]])
    ys = np.concatenate([rng.integers(19, h - 19, n - len(pick)), pick[:, 1]])
    A, B, Cc = a[ys, xs], b[ys, xs], c[ys, xs]
    assert np.all(A == np.round(A)) and np.all(B == np.round(B)) and np.all(Cc == np.round(Cc))
    scale = 1.0 / (4 * 7 * 255.0)
    resp = (A * B - Cc * Cc - 0.04 * (A + B) ** 2) * scale ** 4
    gauss = corner_harris(f, method="k", k=0.04, sigma=2.0)[ys, xs]
    on = slice(n - len(pick), n)
    rho = float(spearmanr(resp[on], gauss[on]).correlation)  # on the corners
    np.savez_compressed(OUT / "skimage_pin3.npz", xs=xs.astype(np.int32), ys=ys.astype(np.int32), a=A.astype(np.int64), b=B.astype(np.int64),
                        c=Cc.astype(np.int64), response=resp, gaussian_harris=gauss, spearman=np.array(rho),
                        skimage_version=np.array(skimage.__version__))
    print(f"wrote {OUT / 'skimage_pin3.npz'}: {n} points, Spearman(box Harris, skimage corner_harris sigma 2) = {rho:.4f}")


if __name__ == "__main__":
    main()
