"""Per-frame latency of the front-end the way the reference calls it (one stereo frame at a time): the one-call form
(snk_frontend_process: one upload, one launch chain / hipGraph, one download, one synchronisation) against the call-by-call path
(Detect x 2, rectify x 2, feature grid, StereoMatching: five round trips).  Median of N frames, distinct images.

    python tools/latency_frontend.py [--frames 200]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snake_slam_amd import synth  # noqa: E402
from snake_slam_amd.frontend import Frontend  # noqa: E402
from snake_slam_amd.matcher import Preprocess, Rectification  # noqa: E402
from snake_slam_amd.orb import ORBExtractor  # noqa: E402
from snake_slam_amd.tracking import FeatureGrid  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--kitti", action="store_true")
    a = ap.parse_args()
    w, h, orb = (1241, 376, (2000, 1.2, 7, 20, 7)) if a.kitti else (752, 480, (1000, 1.2, 4, 20, 7))
    pairs = [synth.stereo_frame(s, w, h) for s in range(16)]
    bounds = (0.0, 0.0, float(w), float(h))
    rect = Rectification.make((458.654, 457.296, 367.215, 248.375))
    fe = Frontend(orb, rect, rect, bounds, 47.9)
    ext, pre, grid = ORBExtractor(*orb), Preprocess(0), FeatureGrid(0)
    ls = fe.level_scale

    def one_call(l, r):
        return fe.Process(l, r)["n_stereo"]

    def by_calls(l, r):
        kl, dl = ext.Detect(l)
        kr, dr = ext.Detect(r)
        ul, _ = pre.rectify(rect, kl)
        ur, _ = pre.rectify(rect, kr)
        perm = np.asarray(grid.create(bounds, ul)[0])
        g, gd = np.zeros_like(ul), np.zeros_like(dl)
        g[perm], gd[perm] = ul, dl
        return pre.StereoMatching(g, gd, ur, dr, 47.9, ls, True)[0]

    import ctypes as C
    from snake_slam_amd import _lib
    lib = _lib.load()
    fr = fe._frame if fe._arrays is not None else None

    def raw_call(l, r):  # the C entry point alone (no numpy result copies): what a C++ caller pays
        return lib.snk_frontend_process(fe._h, l.ctypes.data, w, r.ctypes.data, w, w, h, C.byref(fe._frame))

    for name, fn in (("call by call (host entry points, 6 round trips)", by_calls), ("snk_frontend_process via the Python mirror", one_call),
                     ("snk_frontend_process, C entry point only", raw_call)):
        for k in range(8):
            fn(*pairs[k % 16])
        ts = []
        for k in range(a.frames):
            l, r = pairs[k % 16]
            t0 = time.perf_counter()
            fn(l, r)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e3
        print(f"{name}: median {np.median(ts):.3f} ms, p10 {np.percentile(ts, 10):.3f}, p90 {np.percentile(ts, 90):.3f} ({w}x{h}, {a.frames} frames)")
    for hnd in (fe, ext, pre, grid):
        hnd.close()


if __name__ == "__main__":
    main()
