"""Seeded fuzz of the ORB extractor against the oracle (bit-exact keypoints and descriptors): random image sizes, contents
(rectangles / noise / smooth gradients / mixtures with dense and empty regions), feature counts, levels, scale factors and thresholds.
Not part of the test suite (minutes of GPU time); run after changes to orb.hip:

    python tools/fuzz_orb.py [--seconds 120] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from snake_slam_amd import _lib, synth  # noqa: E402
from snake_slam_amd.orb import ORBExtractor  # noqa: E402


def image(rng, w, h):
    kind = int(rng.integers(0, 5))
    if kind == 0:
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == 1:
        return synth.stereo_frame(int(rng.integers(0, 10000)), w, h, n_rects=int(rng.integers(5, 600)))[0]
    if kind == 2:  # smooth: few or no corners, min-threshold fallback
        yy, xx = np.mgrid[0:h, 0:w]
        return np.clip(128 + 60 * np.sin(xx / 17.0) * np.cos(yy / 23.0) + rng.normal(0, rng.uniform(0, 4), (h, w)), 0, 255).astype(np.uint8)
    if kind == 3:  # half noise, half flat: dense cells next to empty ones
        img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
        img[:, : w // 2] = rng.integers(0, 256, (h, w // 2), dtype=np.uint8)
        return img
    img = synth.stereo_frame(int(rng.integers(0, 10000)), w, h, n_rects=int(rng.integers(50, 400)))[0]
    y0, x0 = int(rng.integers(0, max(1, h - 40))), int(rng.integers(0, max(1, w - 40)))
    img[y0:y0 + 40, x0:x0 + 40] = rng.integers(0, 256, img[y0:y0 + 40, x0:x0 + 40].shape, dtype=np.uint8)  # a patch of noise
    return img


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    orc.build()
    rng = np.random.default_rng(a.seed)
    t0, n, kp_total = time.time(), 0, 0
    while time.time() - t0 < a.seconds:
        w, h = int(rng.integers(40, 1300)), int(rng.integers(40, 800))
        nfeat, levels = int(rng.integers(20, 3000)), int(rng.integers(1, 9))
        scale = float(rng.choice([1.05, 1.1, 1.2, 1.3, 1.5, 2.0, 2.5]))
        ini, mn = int(rng.integers(8, 40)), int(rng.integers(2, 9))
        img = np.ascontiguousarray(image(rng, w, h))
        resp = int(rng.integers(0, 2))  # "orb.response": FAST score / Harris response, both sides
        _lib.set_definition("orb.response", resp)
        orc.set_definition("orb.response", resp)
        ext = ORBExtractor(nfeat, scale, levels, ini, mn)
        try:
            kps, desc = ext.Detect(img)
        finally:
            ext.close()
        wk, wd = orc.orb_detect(orc.orb_params(nfeat, scale, levels, ini, mn), img)
        ok = len(kps) == len(wk) and all(np.array_equal(kps[f], wk[f]) for f in ("octave", "x", "y", "size", "response", "angle")) \
            and np.array_equal(desc, wd)
        if not ok:
            np.save("fuzz_orb_failure.npy", img)
            print(f"MISMATCH case {n}: {w}x{h} nfeat {nfeat} levels {levels} scale {scale} th {ini}/{mn} orb.response {resp}: {len(kps)} vs {len(wk)} keypoints "
                  "(image saved to fuzz_orb_failure.npy)")
            return 1
        n += 1
        kp_total += len(kps)
    print(f"fuzz_orb: {n} cases, {kp_total} keypoints, all bit-exact (seed {a.seed}, {time.time() - t0:.0f} s)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
