"""CPU: the ORB oracle against answers produced by an independent implementation (scikit-image 0.18.3), stored in
tests/golden/skimage_pin.npz by tests/golden/pin_against_skimage.py (run in the build container, where skimage exists).
Pins the BRIEF pattern table (skimage's copy of OpenCV's bit_pattern_31) and the FAST-9/16 segment test; it cannot pin
the saiga-side choices (pyramid, distribution, blur), see DESIGN.md section 2."""
import re
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
PIN = np.load(ROOT / "tests" / "golden" / "skimage_pin.npz")


def _inc_table(path):
    txt = re.sub(r"/\*.*?\*/", "", Path(path).read_text(), flags=re.S)
    return np.array([int(v) for v in re.findall(r"-?\d+", txt)], np.int8).reshape(256, 4)


def test_brief_pattern_is_opencv_bit_pattern_31(orc):
    want = PIN["pattern"]
    assert want.shape == (256, 4)
    assert np.array_equal(np.asarray(orc.brief_pattern()).reshape(256, 4).astype(np.int8), want)
    # the oracle's table and the kernels' table are separate files: both must be the pinned one
    assert np.array_equal(_inc_table(ROOT / "oracle" / "brief_pattern.inc"), want)
    assert np.array_equal(_inc_table(ROOT / "snake_slam_amd" / "csrc" / "brief_pattern_31.inc"), want)


def test_fast9_segment_test_matches_skimage_corner_fast(orc):
    nfeat, sf, nl, ini, mn = PIN["params"]
    levels, _ = orc.pyramid(orc.orb_params(int(nfeat), float(sf), int(nl), int(ini), int(mn)), PIN["img"])
    total = 0
    for l, lv in enumerate(levels):
        h, w = lv.shape
        S = np.zeros((h, w), np.int32)
        for y in range(3, h - 3):
            for x in range(3, w - 3):
                S[y, x] = orc.fast_score(lv, x, y)
        for t in (int(ini), int(mn)):
            ys, xs = np.nonzero(S > t)
            mine = np.stack([xs, ys], 1).astype(np.int16)
            assert np.array_equal(mine, PIN[f"fast_l{l}_t{t}"]), (l, t)
            total += len(mine)
    assert total > 1000
