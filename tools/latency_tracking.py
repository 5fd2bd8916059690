"""Per-frame latency of the Tracking thread's chain through the HOST entry points, the way a Snake-SLAM build that swaps the seams one
by one calls it (TrackingCoarse.cpp:234-270, TrackingFine.cpp:149-158): bind the frame once, SearchByProjectionFrameFrame2 (1500
points), refinePose, SearchByProjection2 (10 000 points), refinePose -- four synchronous calls, each with its own upload / download.
The device-resident batch of the same chain is bench.py's `tracking` leg.  (Inputs are built with the test helpers and the oracle;
nothing of the oracle is timed.)

    python tools/latency_tracking.py [--frames 100]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pose_helpers as PH  # noqa: E402
import track_helpers as T  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from snake_slam_amd.tracking import PoseRefinement, SnakeORBMatcher, pose_observations  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=100)
    a = ap.parse_args()
    orc.build()
    rng = np.random.default_rng(99)
    cases = []
    for _ in range(4):
        frame, cam, pose_true, ls, world, _ = T.make_tracking_case(orc, rng, n_clutter=300, m_pts=1500, taken_frac=0.0)
        frame["taken"][:] = 0
        pred = PH.perturb(rng, pose_true, rot=0.006, trans=0.03)
        coarse = T.lm_coarse(orc, world)
        fine = T.lm_fine(orc, rng, world, pred, ls)
        reps = -(-10000 // len(fine))
        fine = np.concatenate([fine] * reps)[:10000].copy()  # 10 000 local-map points (the visible ones repeated: same arithmetic per point)
        cases.append((frame, cam, pred, ls, world, coarse, fine))
    m, ref = SnakeORBMatcher(), PoseRefinement()

    def chain(c):
        frame, cam, pred, ls, world, coarse, fine = c
        t = [time.perf_counter()]
        m.bind_frame(frame)
        t.append(time.perf_counter())
        n, idx = m.SearchByProjectionFrameFrame2(None, cam, pred, coarse, 10.0, 75, 0, ls)
        t.append(time.perf_counter())
        sel = np.nonzero(idx >= 0)[0]
        feat = idx[sel]
        depth = np.where(frame["right_points"][feat] > 0, cam[4] / np.maximum(frame["kps"]["x"][feat] - frame["right_points"][feat], 1e-3), -1.0)
        obs = pose_observations(frame["kps"][feat], depth, ls)
        t.append(time.perf_counter())
        pose, outl, inl = ref.refinePose(cam, pred, world["pos"][sel], obs)
        t.append(time.perf_counter())
        taken = np.zeros(len(frame["kps"]), np.uint8)
        taken[feat] = 1
        m.bound_taken(taken)
        f2 = fine.copy()
        t.append(time.perf_counter())
        n2, idx2, vis, valid = m.SearchByProjection2(None, cam, pose, f2, 4.0, 0.8, ls)
        t.append(time.perf_counter())
        return np.diff(t), n, n2

    for k in range(6):
        chain(cases[k % 4])
    rows = []
    for k in range(a.frames):
        d, n, n2 = chain(cases[k % 4])
        rows.append(d)
    med = np.median(np.array(rows), axis=0) * 1e3
    names = ["bind_frame", "SearchByProjectionFrameFrame2 (1500)", "host glue (observations)", "refinePose", "host glue (taken, copy)", "SearchByProjection2 (10000)"]
    for nm, v in zip(names, med):
        print(f"{nm}: {v:.3f} ms")
    print(f"tracking chain through the host entry points: {med.sum():.3f} ms per frame ({med[[0, 1, 3, 5]].sum():.3f} ms in library calls); "
          f"last frame: {n} coarse / {n2} fine matches ({a.frames} frames)")
    m.close()
    ref.close()


if __name__ == "__main__":
    main()
