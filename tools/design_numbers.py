"""Prints the "Current numbers" table of DESIGN.md section 5 from a checkpoint's files:  python tools/design_numbers.py r05s"""
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def main(tag):
    rnd = tag[:3]
    d = json.loads((ROOT / "profiles" / rnd / f"{tag}_bench_default.json").read_text().splitlines()[0])
    ba, tr, k, cb = d["ba"], d["tracking"], d["kitti"], d["cpu_baseline"]
    g = ba["global_ba"]
    lat = (ROOT / "profiles" / rnd / f"{tag}_latencies.log").read_text()
    m = re.search(r'"depth3": \{"one_thread_fps": (\d+), "two_threads_fps": (\d+)', lat)
    fps1, fps2 = int(m.group(1)), int(m.group(2))
    m = re.search(r'"one_call_ms": ([0-9.]+), "six_calls_ms": ([0-9.]+)', lat)
    one, six = m.group(1), m.group(2)
    pt = json.loads((ROOT / "profiles" / "pipeline_traffic.json").read_text())
    rows = [
        ("front-end, batch (BASELINE config 2)", f"**{d['value'] / 1e3:.1f} k stereo frames/s**, {d['ms_per_step']:.2f} ms per 1024 frames",
         f"`fast_kernel` {d['roofline']['frac']:.3f} of HBM on algorithmic bytes ({d['roofline']['avg_launch_ms']:.2f} ms per 2048 images); pipeline {d['pipeline_roofline']['frac']:.3f}; "
         f"counter traffic {pt['frontend']['hbm_bytes_per_frame'] / 1e6:.2f} MB per frame = {pt['frontend']['hbm_bytes_per_frame'] / d['pipeline_roofline']['algorithmic_bytes_per_frame']:.2f} × algorithmic",
         f"oracle {cb['value']:.0f} frames/s on {cb['cores']} cores, {cb['frames_checked']} frames identical"),
        ("front-end, KITTI (config 3)", f"{k['value'] / 1e3:.1f} k stereo frames/s (512 frames per step)", f"`fast_kernel` {k['roofline']['frac']:.3f}",
         f"oracle {k['cpu_baseline']['value']:.0f} frames/s; {k['checked_against_oracle']['frames_checked']} frames identical"),
        ("front-end, one frame per call", f"synchronous {one} ms (C++ adaptor; six calls {six} ms); **pipelined {fps1 / 1e3:.1f} k frames/s** (depth 3, one thread; {fps2 / 1e3:.1f} k with two threads)",
         "12 launches per stereo frame; a submit costs ≈ 26 µs of host time", "bit-identical to the synchronous call"),
        ("local BA (config 4), 1024 windows", f"**{ba['value'] / 1e3:.0f} k LM iterations/s**, {ba['ms_per_step']:.2f} ms per step; single window {ba['single_window_ms_per_solve']:.2f} ms per solve",
         f"{ba['roofline']['frac']:.2f} of HBM on model bytes, {pt['ba']['hbm_bytes_per_window_iteration'] / 1e6:.2f} MB counter bytes per window-iteration = "
         f"{pt['ba']['hbm_bytes_per_window_iteration'] * ba['value'] / 8e12:.2f} of HBM; f64 {ba['roofline']['flops']['frac']:.2f} of peak",
         f"oracle {cb['ba']['value']:.0f} LM it/s on 1 core; windows 0 / 511 / 1023 within 1e-13"),
        ("global BA, FullBA(4), 300 KF × 150 k obs", f"**{g['ms_per_solve']:.2f} ms** per solve (round 4: 14.7)",
         f"{g['roofline']['frac']:.2f} of HBM on algorithmic bytes (S re-reads are served by the Infinity Cache); counter bytes {pt['gba']['hbm_bytes_per_solve'] / 1e6:.0f} MB per solve; barrier-bound",
         f"oracle {g['cpu_baseline']['value']:.0f} ms on 1 core; RMSE {g['cpu_baseline']['pose_rmse_vs_gpu']:.1e}"),
        ("tracking chain (coarse → pose → fine)", f"{tr['value'] / 1e3:.0f} k frames/s", f"{tr['roofline']['frac']:.3f} of HBM on algorithmic bytes",
         f"oracle {tr['cpu_baseline']['value']:.0f} frames/s on 4 cores, identical"),
        ("pose refinement, host API", f"{d['pose_refine']['value'] / 1e3:.0f} k frames/s", "—", f"oracle {d['pose_refine']['cpu_baseline']['value']:.0f} frames/s"),
    ]
    print("| leg | measured | fraction of the roofline | CPU oracle beside it (checked) |\n|---|---|---|---|")
    for r in rows:
        print("| " + " | ".join(r) + " |")


if __name__ == "__main__":
    main(sys.argv[1])
