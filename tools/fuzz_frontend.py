"""Seeded fuzz of the pipelined per-frame front-end (snk_frontend_submit / snk_frontend_collect) against the synchronous call
(snk_frontend_process) and, for a sample, against the oracle chain: random image sizes (changed only while the pipeline is empty), random
depth 1..4, random interleavings of submits and collects, mono and stereo handles, the "orb.response" switch flipped between bursts.
Not part of the test suite; run after changes to frontend.hip:

    python tools/fuzz_frontend.py [--seconds 60] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from snake_slam_amd import _lib, synth  # noqa: E402
from snake_slam_amd.frontend import Frontend  # noqa: E402

KEYS = ("keypoints", "descriptors", "undistorted_keypoints", "normalized_points", "permutation", "cell_start", "right_points", "depth",
        "keypoints_right", "descriptors_right")


def same(a, b):
    return a["N"] == b["N"] and a["n_right"] == b["n_right"] and a["n_stereo"] == b["n_stereo"] and \
        all(np.array_equal(np.asarray(a[k]).view(np.uint8), np.asarray(b[k]).view(np.uint8)) for k in KEYS)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    orc.build()
    rng = np.random.default_rng(a.seed)
    t0, bursts, frames, oracle_checks = time.time(), 0, 0, 0
    while time.time() - t0 < a.seconds:
        stereo = bool(rng.integers(0, 4))
        nfeat, levels = int(rng.integers(50, 1200)), int(rng.integers(1, 6))
        orb = (nfeat, float(rng.choice([1.2, 1.3, 1.5])), levels, int(rng.integers(10, 30)), int(rng.integers(3, 9)))
        fe = Frontend(orb, bounds=(0.0, 0.0, 800.0, 520.0), stereo=stereo)
        try:
            for _ in range(int(rng.integers(1, 4))):  # bursts on one handle: each with its own size, depth and definition
                w, h = int(rng.integers(12, 100)) * 8, int(rng.integers(12, 64)) * 8
                resp = int(rng.integers(0, 2))
                _lib.set_definition("orb.response", resp)
                orc.set_definition("orb.response", resp)
                pairs = [synth.stereo_frame(int(rng.integers(0, 10000)), w, h, n_rects=int(rng.integers(5, 200))) for _ in range(3)]
                want = [fe.Process(l, r if stereo else None) for l, r in pairs]
                if rng.integers(0, 3) == 0:  # the synchronous call itself against the oracle's extractor
                    wk, _ = orc.orb_detect(orc.orb_params(*orb), pairs[0][0])
                    g = sorted(zip(want[0]["keypoints"]["x"], want[0]["keypoints"]["y"], want[0]["keypoints"]["octave"], want[0]["keypoints"]["response"]))
                    if g != sorted(zip(wk["x"], wk["y"], wk["octave"], wk["response"])):
                        print(f"MISMATCH against the oracle: {w}x{h} orb {orb} stereo {stereo} response {resp}")
                        return 1
                    oracle_checks += 1
                depth = int(rng.integers(1, 5))
                fe.set_depth(depth)
                queue = []
                for _ in range(int(rng.integers(3, 30))):
                    if queue and (len(queue) == depth or rng.integers(0, 2)):
                        k = queue.pop(0)
                        if not same(fe.Collect(), want[k]):
                            print(f"MISMATCH: {w}x{h} orb {orb} stereo {stereo} depth {depth} response {resp} frame {k}")
                            return 1
                        frames += 1
                    else:
                        k = int(rng.integers(0, 3))
                        fe.Submit(pairs[k][0], pairs[k][1] if stereo else None)
                        queue.append(k)
                while queue:
                    k = queue.pop(0)
                    if not same(fe.Collect(), want[k]):
                        print(f"MISMATCH (drain): {w}x{h} orb {orb} stereo {stereo} depth {depth} response {resp} frame {k}")
                        return 1
                    frames += 1
                bursts += 1
        finally:
            fe.close()
            _lib.set_definition("orb.response", 0)
            orc.set_definition("orb.response", 0)
    print(f"fuzz_frontend: {bursts} bursts, {frames} pipelined frames identical to the synchronous call, {oracle_checks} oracle checks (seed {a.seed}, {time.time() - t0:.0f} s)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
