"""Per-frame latency of the host (synchronous) projection matchers, the reference's actual call pattern (TrackingCoarse.cpp:234,
TrackingFine.cpp:149): SearchByProjectionFrameFrame2 with 1500 points and SearchByProjection2 with 10 000 points on one frame.
SNK_TRACK_PPW forces the points-per-wavefront choice (track.hip points_per_wave)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import track_helpers as T  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (test infrastructure: builds the synthetic tracking case)
from snake_slam_amd.tracking import SnakeORBMatcher  # noqa: E402


def timeit(f, n=200):
    for _ in range(5):
        f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    rng = np.random.default_rng(5)
    frame, cam, pose, ls, world, _ = T.make_tracking_case(orc, rng, n_clutter=400, m_pts=1500)
    pc = T.lm_coarse(orc, world)
    m = SnakeORBMatcher(0)
    m.bind_frame(frame)
    res = {}
    if pc is not None:
        res["coarse_1500_bound_ms"] = timeit(lambda: m.SearchByProjectionFrameFrame2(None, cam, pose, pc, 10.0, 75, 0, ls))
    frame2, cam2, pose2, ls2, world2, _ = T.make_tracking_case(orc, rng, n_clutter=400, m_pts=10000)
    pf = T.lm_fine(orc, rng, world2, pose2, ls2)
    m.bind_frame(frame2)
    if pf is not None:
        res["fine_10000_bound_ms"] = timeit(lambda: m.SearchByProjection2(None, cam2, pose2, pf.copy(), 4.0, 0.8, ls2))
    print("ppw=%s %s" % (os.environ.get("SNK_TRACK_PPW", "auto"), res))


if __name__ == "__main__":
    main()
