#!/usr/bin/env python3
"""Counter bytes (rocprofv3 FETCH_SIZE / WRITE_SIZE passes) of (a) every kernel of a front-end step and (b) one LM iteration of the
1024-window BA, written to profiles/pipeline_traffic.json -- what bench.py prints as `pipeline_roofline.traffic` and
`ba.roofline.traffic` (the model fraction of the BA leg is a fraction of bytes the fused kernels do not move; the
counters say what they do move).

usage: collect_pipeline_traffic.py <gpurun_out/prof_TAG> <gpurun_out/prof_ba_TAG> <frames per step> <windows per launch>

Corrections as MI355X_MICROARCH.md (section HBM) prescribes: FETCH_SIZE on gfx950 reports 1/2 of the bytes of wide coalesced
reads -- the value is doubled ("fetch_x2"; exact for the 16-byte-per-lane streams of level_kernel / fast_kernel / schur_fused's
records, an upper estimate for narrower gathers, so both raw and doubled figures are stored); WRITE_SIZE as reported.  Units: KB.
The file records the sha256 of orb.hip / matcher.hip / track.hip / ba.hip; bench.py ignores a section whose sources changed."""
import csv
import hashlib
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "snake_slam_amd" / "csrc"


def per_kernel(d: Path, counter: str):
    """kernel -> (mean bytes per dispatch, dispatches per run)"""
    vals = defaultdict(lambda: defaultdict(float))
    for f in (d / f"pmc_{counter}").rglob("*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter:
                    n = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
                    n = re.sub(r"^void ", "", n).split("(")[0]
                    if n.startswith("snk::"):
                        vals[n][row["Dispatch_Id"]] += float(row["Counter_Value"]) * 1024.0
    # only the batch-sized dispatches of a kernel (a run may also hold per-frame launches of the same kernel, orders of magnitude smaller)
    out = {}
    for k, v in vals.items():
        d = list(v.values())
        top = max(d)
        big = [x for x in d if x >= 0.25 * top] if top > 0 else d
        out[k] = (sum(big) / len(big), len(big))
    return out


def sha(*names):
    h = hashlib.sha256()
    for n in names:
        h.update((CSRC / n).read_bytes())
    return h.hexdigest()


def main():
    dfe, dba, frames, windows = Path(sys.argv[1]), Path(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    out = {}
    f, w = per_kernel(dfe, "FETCH_SIZE"), per_kernel(dfe, "WRITE_SIZE")
    if f:
        # tools/profile_gpu.sh runs bench.py --steps 3 --warmup 1: 4 steps; a kernel launched k times per step shows 4 k dispatches
        steps = 4
        ks = {}
        fe_kernels = ("level_kernel", "fast_kernel", "distribute", "describe_kernel", "rectify_kernel", "grid_kernel", "reorder_kernel",
                      "stereo_", "bf_knn2", "bf_filter", "resize_kernel")
        tot_raw = tot_x2 = tot_w = 0.0
        for k in sorted(set(f) | set(w)):
            if not any(s in k for s in fe_kernels):
                continue
            fb, nd = f.get(k, (0.0, 0))
            wb, _ = w.get(k, (0.0, 0))
            per_step = nd / steps
            ks[k] = {"launches_per_step": round(per_step, 2), "fetch_bytes_per_step": int(fb * per_step), "write_bytes_per_step": int(wb * per_step)}
            tot_raw += fb * per_step
            tot_x2 += 2 * fb * per_step
            tot_w += wb * per_step
        out["frontend"] = {"frames_per_step": frames, "kernels": ks, "fetch_bytes_per_frame_raw": int(tot_raw / frames),
                           "fetch_bytes_per_frame_x2": int(tot_x2 / frames), "write_bytes_per_frame": int(tot_w / frames),
                           "hbm_bytes_per_frame": int((tot_x2 + tot_w) / frames), "run": dfe.name,
                           "source_sha256": sha("orb.hip", "matcher.hip", "track.hip", "preprocess.hip")}
    f, w = per_kernel(dba, "FETCH_SIZE"), per_kernel(dba, "WRITE_SIZE")
    if f:
        it_kernels = ("schur_fused", "update_cost", "cam_pass", "pcg_small", "pcg_solve", "schur_sum", "accept_pass", "update_pass", "schur_pass",
                      "point_wave", "update_wave", "cost_wave")
        ks = {}
        tot_raw = tot_w = 0.0
        for k in sorted(set(f) | set(w)):
            if not any(s in k for s in it_kernels):
                continue
            fb, nd = f.get(k, (0.0, 0))
            wb, _ = w.get(k, (0.0, 0))
            ks[k] = {"dispatches": nd, "fetch_bytes_per_launch": int(fb), "write_bytes_per_launch": int(wb)}
            tot_raw += fb  # every kernel of the list runs once per LM iteration
            tot_w += wb
        out["ba"] = {"windows_per_launch": windows, "kernels": ks, "fetch_bytes_per_window_iteration_raw": int(tot_raw / windows),
                     "fetch_bytes_per_window_iteration_x2": int(2 * tot_raw / windows), "write_bytes_per_window_iteration": int(tot_w / windows),
                     "hbm_bytes_per_window_iteration": int((2 * tot_raw + tot_w) / windows), "run": dba.name, "source_sha256": sha("ba.hip")}
    out["note"] = ("MI355X_MICROARCH.md section HBM: FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads; an upper estimate "
                   "for narrow gathers), WRITE_SIZE as reported; separate --pmc passes per counter")
    (ROOT / "profiles" / "pipeline_traffic.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "kernels"} for k, v in out.items() if isinstance(v, dict)}, indent=1))


if __name__ == "__main__":
    main()
