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
    ap.add_argument("--creates", type=int, default=1, help="hand-overs of the batch in front of the solves (profiling the hand-over's kernels: --creates 4 --solves 0)")
    a = ap.parse_args()
    distinct = [synth.ba_scene(seed=synth.SEED + k)[0] for k in range(4)]
    ba = BARec(lba_options())
    tc = time.perf_counter()
    ba.create([distinct[k % 4] for k in range(a.windows)])
    tc = time.perf_counter() - tc
    for _ in range(a.creates - 1):  # warm hand-overs on the handle
        tw = time.perf_counter()
        ba.create([distinct[k % 4] for k in range(a.windows)])
        ba.sync()
        print(f"warm hand-over {(time.perf_counter() - tw) * 1e3 - ba.last_pack_ms:.1f} ms (without the binding's packing)")
    ba.solve_async(3)
    ba.sync()
    t0 = time.perf_counter()
    for _ in range(a.solves):
        ba.reset()
        ba.solve_async(3)
    ba.sync()
    dt = max(time.perf_counter() - t0, 1e-9)
    a.solves = max(a.solves, 1)
    print(f"{a.windows} windows x {a.solves} solves x 3 LM iterations: {a.windows * a.solves * 3 / dt:.0f} LM iterations/s, "
          f"{dt / a.solves / 3 * 1e3:.4f} ms per LM iteration of the batch; create (host lists + upload of the batch) {tc * 1e3:.1f} ms")
    ba.close()


if __name__ == "__main__":
    main()
