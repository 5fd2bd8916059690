"""Latency of the reference's real local-BA call (LocalBundleAdjustment.cpp:353-413): for a NEW scene every keyframe
create(scene) -> initAndSolve() [3 LM iterations] -> chi-square pass -> mark outliers -> solve() [1 more iteration]
-> read the state back.  Run once per launch policy (the policy is read from the environment at the first solve):

    python tools/lba_call_latency.py                       # default: plain launches first, graph on a repeat
    SNK_BA_GRAPH_FIRST=1 python tools/lba_call_latency.py  # build + instantiate a graph for every new scene
    SNK_BA_NO_GRAPH=1 python tools/lba_call_latency.py     # never a graph

Prints one JSON line; stage times are host wall-clock per call (every stage ends with a stream synchronisation)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snake_slam_amd import synth  # noqa: E402
from snake_slam_amd.ba import BARec, lba_options  # noqa: E402


def main():
    n_scenes = int(os.environ.get("LBA_SCENES", "12"))
    scenes = [synth.ba_scene(seed=1000 + i, outlier_frac=float(os.environ.get("LBA_OUTLIER_FRAC", "0.02")))[0] for i in range(n_scenes)]  # 20 x 2000 x 8, a new one per call
    ba = BARec(lba_options())
    stages = {k: [] for k in ("create", "initAndSolve", "residuals", "set_outliers", "solve1", "state", "total")}
    marked_steps, marked_fused = [], []
    for rep in range(3):  # first pass = warm-up (buffers grow, kernels load)
        for s in scenes:
            t = [time.perf_counter()]
            ba.create(s)
            t.append(time.perf_counter())
            ba.initAndSolve()
            t.append(time.perf_counter())
            chi = ba.residuals(0)
            t.append(time.perf_counter())
            thr = np.where(np.asarray(s["obs_depth"]) > 0, 5.29, 4.41)
            mask = (chi > thr).astype(np.uint8)
            marked_steps.append(int(mask.sum()))
            if mask.any():  # LocalBundleAdjustment.cpp:399
                ba.set_outliers(0, mask)
            t.append(time.perf_counter())
            if mask.any():
                ba.solve(1)
            t.append(time.perf_counter())
            ba.state(0)
            t.append(time.perf_counter())
            if rep == 0:
                continue
            for k, (a, b) in zip(list(stages)[:-1], zip(t[:-1], t[1:])):
                stages[k].append((b - a) * 1e3)
            stages["total"].append((t[-1] - t[0]) * 1e3)
    # the same call through snk_ba_solve_local_scene: create -> one fused call (solve, chi-square pass on the device, the extra
    # iteration, results through the pinned buffer)
    fused = {k: [] for k in ("create", "solve_local_scene", "total")}
    for rep in range(3):
        for s in scenes:
            t0 = time.perf_counter()
            ba.create(s)
            t1 = time.perf_counter()
            marked_fused.append(ba.solve_local_scene(4.41, 5.29)[0])
            t2 = time.perf_counter()
            if rep:
                fused["create"].append((t1 - t0) * 1e3)
                fused["solve_local_scene"].append((t2 - t1) * 1e3)
                fused["total"].append((t2 - t0) * 1e3)
    mode = "graph_first" if os.environ.get("SNK_BA_GRAPH_FIRST") else ("no_graph" if os.environ.get("SNK_BA_NO_GRAPH") else "default")
    print(json.dumps({"tool": "lba_call_latency", "mode": mode, "scene": "20 KF x 2000 pts x 8 obs", "calls": len(stages["total"]),
                      "median_ms": {k: round(float(np.median(v)), 4) for k, v in stages.items()},
                      "min_ms": {k: round(float(np.min(v)), 4) for k, v in stages.items()},
                      "marked_per_call": {"steps": [min(marked_steps), max(marked_steps)], "fused": [min(marked_fused), max(marked_fused)],
                                          "same": marked_steps == marked_fused},
                      "fused_median_ms": {k: round(float(np.median(v)), 4) for k, v in fused.items()},
                      "fused_min_ms": {k: round(float(np.min(v)), 4) for k, v in fused.items()}}))
    ba.close()


if __name__ == "__main__":
    main()
