"""Seeded synthetic inputs of the benchmark workloads: corner-rich stereo frames,
random descriptor sets and the 20-keyframe x 2000-point local-BA scene.  Pure numpy; used by
tests/ and bench.py (there is no network for datasets)."""
from __future__ import annotations

import numpy as np

SEED = 363456635  # the reference's randomSeed (reference configs/euroc.ini:3)


def _draw_rects(img, rects, shift=None):
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
        img[y0:y1, x0:x1][m] = g


def stereo_frame(index: int = 0, width: int = 752, height: int = 480, n_rects: int = 400, seed: int = SEED):
    """Returns (left, right) uint8 images: low-frequency gradient + random (rotated) rectangles +
    Gaussian noise sigma=2; the right image shifts every rectangle by its own disparity in [2,60]."""
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
    _draw_rects(left, rects)
    _draw_rects(right, rects, disp)
    left += rng.normal(0, 2.0, left.shape)
    right += rng.normal(0, 2.0, right.shape)
    return (np.clip(np.rint(left), 0, 255).astype(np.uint8), np.clip(np.rint(right), 0, 255).astype(np.uint8))


def random_descriptors(n: int, seed: int = SEED):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
