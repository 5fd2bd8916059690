"""Only the batched BA leg: B windows of the 20 x 2000 x 8 scene, `--solves` solves of 3 LM iterations.  For profiling:
every kernel launch of the run is a B-window launch (bench.py's BA leg mixes in single-window and global-BA launches, which
makes per-kernel averages of a trace unusable).   python tools/ba_batch_only.py [--windows 256] [--solves 6]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snake_slam_amd import synth  # noqa: E402
from snake_slam_amd.ba import BARec, lba_options  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=256)
    ap.add_argument("--solves", type=int, default=6)
    a = ap.parse_args()
    distinct = [synth.ba_scene(seed=synth.SEED + k)[0] for k in range(4)]
    ba = BARec(lba_options())
    tc = time.perf_counter()
    ba.create([distinct[k % 4] for k in range(a.windows)])
    tc = time.perf_counter() - tc
    ba.solve_async(3)
    ba.sync()
    t0 = time.perf_counter()
    for _ in range(a.solves):
        ba.reset()
        ba.solve_async(3)
    ba.sync()
    dt = time.perf_counter() - t0
    print(f"{a.windows} windows x {a.solves} solves x 3 LM iterations: {a.windows * a.solves * 3 / dt:.0f} LM iterations/s, "
          f"{dt / a.solves / 3 * 1e3:.4f} ms per LM iteration of the batch; create (host lists + upload of the batch) {tc * 1e3:.1f} ms")
    ba.close()


if __name__ == "__main__":
    main()
