"""Seeded fuzz of the device-built BA lists alone (no solve): random scenes with 2 .. 530 keyframes handed over with
SNK_BA_CHECK_LISTS=1, under which snk_ba_set_problems compares the camera records, the work-item records and the camera-pair
block entries with the host builder / the caller's arrays and fails on the first difference.
    SNK_BA_CHECK_LISTS=1 python tools/probes/fuzz_ba_lists.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from snake_slam_amd import synth  # noqa: E402
from snake_slam_amd.ba import BARec, lba_options  # noqa: E402


def main():
    assert os.environ.get("SNK_BA_CHECK_LISTS"), "run with SNK_BA_CHECK_LISTS=1"
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ba = BARec(lba_options())
    t0, n, cams = time.time(), 0, []
    while time.time() - t0 < seconds:
        n_kf = int(rng.choice([2, 5, 20, 63, 64, 65, 66, 127, 129, 200, 300, 511, 512, 513, 530, int(rng.integers(2, 530))]))
        opp = int(rng.integers(2, min(n_kf, 14) + 1))
        n_pt = int(rng.integers(max(4, n_kf // 2), 4 * n_kf + 40))
        scs = []
        for _ in range(int(rng.choice([1, 1, 1, 3]))):
            sc, _ = synth.ba_scene(n_kf=n_kf, n_pt=n_pt, obs_per_pt=opp, seed=int(rng.integers(0, 1 << 30)), n_fixed=int(rng.integers(1, max(2, n_kf // 3))))
            if rng.random() < 0.3:
                sc["pt_const"] = (rng.random(n_pt) < 0.2).astype(np.uint8)
            if rng.random() < 0.2:  # a few invalid indices
                sc["obs_img"] = sc["obs_img"].copy()
                sc["obs_img"][rng.integers(0, len(sc["obs_img"]), 3)] = -1
            scs.append(sc)
        ba.create(scs)  # raises RuntimeError on a list mismatch
        ba.sync()
        n += 1
        cams.append(n_kf)
    ba.close()
    print(f"fuzz_ba_lists: {n} hand-overs, 2..{max(cams)} keyframes, every device-built list equal to the host builder's ({seconds:.0f} s)")


if __name__ == "__main__":
    main()
