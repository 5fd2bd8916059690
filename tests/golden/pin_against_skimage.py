#!/opt/conda/bin/python3.9
"""Pins two pieces of the ORB oracle against an INDEPENDENT implementation (scikit-image 0.18.3, which is only
installed for /opt/conda's python3.9 in the build container -- it does not travel, the fixture it writes does):

 (i)  the 256 BRIEF test pairs (oracle/brief_pattern.inc == snake_slam_amd/csrc/brief_pattern_31.inc) against
      skimage/feature/orb_descriptor_positions.txt, scikit-image's copy of OpenCV's `bit_pattern_31`;
 (ii) the FAST-9/16 segment test: on every pyramid level of the golden image, the pixels where the oracle's corner
      score S(p) exceeds the threshold t (t = iniThFAST 20 and minThFAST 7) against the pixels where
      skimage.feature.corner_fast(level, n=9, threshold=(t + 0.5) / 255) responds (skimage works on the image scaled
      to [0, 1]; the half step keeps its float compare `I > c + t` away from ties).

It stores skimage's answers in tests/golden/skimage_pin.npz; tests/test_oracle_pin.py (CPU suite, no skimage needed)
compares the oracle with them on every run.  This does NOT pin the path against saiga (absent);
it removes the doubt that the pattern table or the segment test were mis-transcribed.

    /opt/conda/bin/python3.9 tests/golden/pin_against_skimage.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

import skimage  # noqa: E402
from skimage.feature import corner_fast  # noqa: E402

from oracle import oracle as orc  # noqa: E402

OUT = Path(__file__).resolve().parent


def main():
    orc.build()
    pos = np.loadtxt(Path(skimage.__file__).parent / "feature" / "orb_descriptor_positions.txt").astype(np.int8)
    assert pos.shape == (256, 4)
    ours = np.asarray(orc.brief_pattern()).reshape(256, 4).astype(np.int8)
    # same column order as OpenCV's table: x0 y0 x1 y1
    assert np.array_equal(ours, pos), "brief pattern differs from scikit-image's bit_pattern_31"
    g = np.load(OUT / "orb_small.npz")
    img = g["img"]
    nfeat, sf, nl, ini, mn = g["params"]
    levels, _ = orc.pyramid(orc.orb_params(int(nfeat), float(sf), int(nl), int(ini), int(mn)), img)
    out = {"pattern": pos, "img": img, "params": g["params"], "skimage_version": np.array(skimage.__version__)}
    for l, lv in enumerate(levels):
        for t in (int(ini), int(mn)):
            resp = corner_fast(lv, n=9, threshold=(t + 0.5) / 255.0)
            ys, xs = np.nonzero(resp > 0)
            out[f"fast_l{l}_t{t}"] = np.stack([xs, ys], 1).astype(np.int16)
            # check right here as well
            h, w = lv.shape
            S = np.zeros((h, w), np.int32)
            for y in range(3, h - 3):
                for x in range(3, w - 3):
                    S[y, x] = orc.fast_score(lv, x, y)
            mine = S > t
            theirs = resp > 0
            assert np.array_equal(mine, theirs), f"FAST-9 set differs from skimage at level {l}, threshold {t}"
            print(f"level {l} ({w}x{h}) t={t}: {int(mine.sum())} corners, identical to skimage.feature.corner_fast")
    np.savez_compressed(OUT / "skimage_pin.npz", **out)
    print("brief pattern: 256 x 4 entries identical to skimage's orb_descriptor_positions.txt")
    print("wrote", OUT / "skimage_pin.npz")


if __name__ == "__main__":
    main()
