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


def per_kernel_total(d: Path, counter: str):
    """kernel -> (bytes summed over ALL dispatches of the run, dispatches).  Round 5: the front-end section is total / steps -- the round-4
    form (mean over the "big" dispatches x count) dropped level_kernel's smallest level from the mean and over-counted it by 1.2."""
    vals = defaultdict(lambda: defaultdict(float))
    for f in (d / f"pmc_{counter}").rglob("*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter:
                    n = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
                    n = re.sub(r"^void ", "", n).split("(")[0]
                    if n.startswith("snk::"):
                        vals[n][row["Dispatch_Id"]] += float(row["Counter_Value"]) * 1024.0
    return {k: (sum(v.values()), len(v)) for k, v in vals.items()}


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
    f, w = per_kernel_total(dfe, "FETCH_SIZE"), per_kernel_total(dfe, "WRITE_SIZE")
    if f:
        # tools/profile_gpu.sh runs bench.py --steps 3 --warmup 1 (no per-frame leg, no KITTI leg): 4 steps; every front-end dispatch of the run
        # belongs to one of them
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
            ks[k] = {"launches_per_step": round(nd / steps, 2), "fetch_bytes_per_step": int(fb / steps), "write_bytes_per_step": int(wb / steps)}
            tot_raw += fb / steps
            tot_x2 += 2 * fb / steps
            tot_w += wb / steps
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
    # global BA (tools/profile_gba.sh <tag> pmc): counter bytes of one FullBA(4) = everything the run's solves moved / solves
    dg = dfe.parent / dfe.name.replace("prof_", "prof_gba_")
    if (dg / "pmc_FETCH_SIZE").exists():
        f, w = per_kernel_total(dg, "FETCH_SIZE"), per_kernel_total(dg, "WRITE_SIZE")
        solves = 4  # tools/gba_trace.py: the first solve + three timed ones
        ks = {k: {"dispatches": f.get(k, (0, 0))[1], "fetch_bytes_per_solve": int(f.get(k, (0.0, 0))[0] / solves), "write_bytes_per_solve": int(w.get(k, (0.0, 0))[0] / solves)}
              for k in sorted(set(f) | set(w))}
        tf, tw = sum(v[0] for v in f.values()) / solves, sum(v[0] for v in w.values()) / solves
        out["gba"] = {"scene": "300 keyframes x 15 000 points x 10 observations, FullBA(4)", "kernels": ks, "fetch_bytes_per_solve_raw": int(tf),
                      "fetch_bytes_per_solve_x2": int(2 * tf), "write_bytes_per_solve": int(tw), "hbm_bytes_per_solve": int(2 * tf + tw), "run": dg.name,
                      "source_sha256": sha("ba.hip")}
    out["note"] = ("MI355X_MICROARCH.md section HBM: FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads; an upper estimate "
                   "for narrow gathers), WRITE_SIZE as reported; separate --pmc passes per counter")
    (ROOT / "profiles" / "pipeline_traffic.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "kernels"} for k, v in out.items() if isinstance(v, dict)}, indent=1))


if __name__ == "__main__":
    main()
