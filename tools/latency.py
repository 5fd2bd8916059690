"""Single-call latencies of the host (synchronous) entry points, the way the reference calls them."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snake_slam_amd import synth  # noqa: E402
from snake_slam_amd.matcher import BruteForceMatcher, Preprocess, Rectification  # noqa: E402
from snake_slam_amd.orb import ORBExtractor  # noqa: E402


def timeit(f, n=50):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    left, right = synth.stereo_frame(0)
    ext = ORBExtractor(1000, 1.2, 4, 20, 7)
    kl, dl = ext.Detect(left)
    kr, dr = ext.Detect(right)
    print("ORBExtractor.Detect 752x480 (H2D + kernels + D2H, sync): %.3f ms, %d keypoints" % (timeit(lambda: ext.Detect(left)), len(kl)))
    pre = Preprocess()
    rect = Rectification.make((1.0, 1.0, 0.0, 0.0))
    rl, rr = pre.rectify(rect, kl)[0], pre.rectify(rect, kr)[0]
    ls = (np.float32(1.2) ** np.arange(4)).astype(np.float32)
    print("Preprocess.Rectify 1000 keypoints: %.3f ms" % timeit(lambda: pre.rectify(rect, kl)))
    print("Preprocess.StereoMatching 1000 x 1000: %.3f ms" % timeit(lambda: pre.StereoMatching(rl, dl, rr, dr, 47.9 * 2.5, ls, True)))
    bf = BruteForceMatcher()
    print("BruteForceMatcher.matchKnn2 1000 x 1000 + filter: %.3f ms" % timeit(lambda: (bf.matchKnn2(dl, dr), bf.filterMatches(60, 0.8))))


if __name__ == "__main__":
    main()
