#!/usr/bin/env python3
"""The pipelined per-frame front-end driven from Python / ctypes in a process of its own: snk_frontend_submit (or _submit_pinned) +
snk_frontend_collect, one frame per call, depth 3, one thread.  bench.py measures the same loop inside ITS process, where a dozen
other handles (extractor, matchers, BA, torch's own streams) hold HIP streams: a process has four hardware queues, streams that
share one run one behind the other, and the three slots of the front-end then overlap less (profiles/NOTES.md, round 5: the same
effect cost 14.4 -> 9.8 k frames/s inside the C++ tool until the slots bound their queues first).  This script is the ctypes host
without that company.   python tools/frontend_pipelined_py.py [frames, default 2000]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from snake_slam_amd import _lib, synth  # noqa: E402
from snake_slam_amd.frontend import Frontend  # noqa: E402
from snake_slam_amd.matcher import Rectification  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    W, H, depth = 752, 480, 3
    pairs = [synth.stereo_frame(900 + k, W, H) for k in range(8)]
    rect = Rectification.make((458.654, 457.296, 367.215, 248.375))
    fe = Frontend((1000, 1.2, 4, 20, 7), rect, rect, (0.0, 0.0, float(W), float(H)), 47.9)
    lib = _lib.load()
    want = [fe.Process(*p) for p in pairs]
    fe.set_depth(depth)
    pin = [fe.pinned_images(W, H, 2) for _ in range(8)]
    for k in range(8):
        pin[k][0], pin[k][1] = pairs[k]
    adr_staged = [(int(l.ctypes.data), int(r.ctypes.data)) for l, r in pairs]
    adr_pinned = [(int(q[0].ctypes.data), int(q[1].ctypes.data)) for q in pin]
    frame = C.byref(fe._frame)
    out = {"tool": "tools/frontend_pipelined_py.py", "image": f"{W}x{H} stereo", "depth": depth, "frames": n}
    for name, fn, adr in (("staged", lib.snk_frontend_submit, adr_staged), ("pinned", lib.snk_frontend_submit_pinned, adr_pinned)):
        same = True
        for k in range(8 + depth - 1):  # checked pass (every slot past its captured frame)
            if k < 8:
                a, b = adr[k]
                _lib.check(fn(fe._h, a, W, b, W, W, H), name)
            if k >= depth - 1:
                g = fe.Collect()
                w = want[k - depth + 1]
                same = same and all(np.array_equal(g[key], w[key]) for key in w)

        def run(count):
            for k in range(count + depth - 1):
                if k < count:
                    a, b = adr[k % 8]
                    fn(fe._h, a, W, b, W, W, H)
                if k >= depth - 1:
                    lib.snk_frontend_collect(fe._h, frame, -1)

        run(4 * depth)
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            run(n)
            best = max(best, n / (time.perf_counter() - t0))
        out[name] = {"frames_per_s": round(best, 1), "identical_to_process": bool(same)}
    fe.close()
    print(json.dumps(out))
    return 0 if out["staged"]["identical_to_process"] and out["pinned"]["identical_to_process"] else 3


if __name__ == "__main__":
    sys.exit(main())
