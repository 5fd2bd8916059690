"""Where the hand-over of a batch of windows goes (309-398 ms for 1024 windows against 6 ms per solve step):
python-side packing, the C call (host lists, allocation, upload), first time and again on the same handle.
    SNK_BA_PROFILE_CREATE=1 python tools/probes/ba_handover_profile.py [--windows 1024]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from snake_slam_amd import synth  # noqa: E402
from snake_slam_amd import ba as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=1024)
    a = ap.parse_args()
    distinct = [synth.ba_scene(seed=synth.SEED + k)[0] for k in range(256)]
    scenes = [distinct[k % 256] for k in range(a.windows)]
    t0 = time.perf_counter()
    packed = [B._pack(s) for s in scenes]
    t1 = time.perf_counter()
    print(f"python _pack of {a.windows} scenes: {(t1 - t0) * 1e3:.1f} ms")
    ba = B.BARec(B.lba_options())
    for rep in range(3):
        t0 = time.perf_counter()
        ba.create(scenes)
        ba.sync()
        t1 = time.perf_counter()
        print(f"create #{rep} ({a.windows} windows, incl. python packing): {(t1 - t0) * 1e3:.1f} ms", flush=True)
    ba.close()


if __name__ == "__main__":
    main()
