#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE passes of tools/profile_gpu.sh into profiles/fast_kernel_traffic.json, the file
bench.py reads for `roofline.traffic`.

usage: collect_traffic.py <gpurun_out/prof_TAG> <images_per_launch> [kernel name prefix, default snk::fast_kernel] [workload, default euroc]

workload = kitti (a pass of tools/profile_gpu.sh run with `--workload kitti --batch 512`) writes profiles/fast_kernel_traffic_kitti.json:
the KITTI leg of bench.py reads ITS OWN counters (a KITTI image has 1.55 x the pixels, 7 levels, other cell shapes); it no longer
scales the EuRoC pass by the image count (round-5 review).

Corrections (MI355X_MICROARCH.md, section HBM): on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide
coalesced read (16 B per lane) -- fast_kernel's tile loader is `global_load_dwordx4`, so FETCH_SIZE is doubled;
WRITE_SIZE is uncalibrated and taken as is (it is 3 % of the total).  Units: the counters are in KB.
The file records the sha256 of the kernel's source; bench.py refuses a file whose hash is not the current source's."""
import csv
import hashlib
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SOURCE = ROOT / "snake_slam_amd" / "csrc" / "orb.hip"


def mean_counter(d: Path, counter: str, prefix: str):
    vals = defaultdict(list)
    for f in d.rglob("*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter:
                    name = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
                    name = re.sub(r"^void ", "", name)
                    if name.startswith(prefix):
                        vals[row.get("Dispatch_Id", len(vals))].append(float(row["Counter_Value"]))
    per_dispatch = [sum(v) for v in vals.values()]  # a dispatch's rows (one per XCD / instance) add up
    # only the batch-sized launches: a run may contain per-frame launches of the same kernel (bench.py's frontend_frame leg), orders of
    # magnitude smaller -- they must not enter the mean of "one launch over 2048 images"
    if per_dispatch:
        top = max(per_dispatch)
        per_dispatch = [v for v in per_dispatch if v >= 0.5 * top]
    return (sum(per_dispatch) / len(per_dispatch), len(per_dispatch)) if per_dispatch else (None, 0)


def main():
    d, images = Path(sys.argv[1]), int(sys.argv[2])
    prefix = sys.argv[3] if len(sys.argv) > 3 else "snk::fast_kernel"
    workload = sys.argv[4] if len(sys.argv) > 4 else "euroc"
    fetch_kb, nf = mean_counter(d / "pmc_FETCH_SIZE", "FETCH_SIZE", prefix)
    write_kb, nw = mean_counter(d / "pmc_WRITE_SIZE", "WRITE_SIZE", prefix)
    if fetch_kb is None or write_kb is None:
        raise SystemExit(f"no {prefix} dispatches with FETCH_SIZE / WRITE_SIZE under {d}")
    FETCH_CORRECTION = 2.0
    hbm = int(fetch_kb * 1024 * FETCH_CORRECTION + write_kb * 1024)
    # instruction counters of the same kernel (tools/profile_gpu.sh passes pmc_sq / pmc_sq2): what bench.py's `roofline.valu`
    # is computed from.  The counters of a dispatch add up over the XCDs; GRBM_GUI_ACTIVE / 8 = busy cycles of the launch.
    valu, _ = mean_counter(d / "pmc_sq", "SQ_INSTS_VALU", prefix)
    salu, _ = mean_counter(d / "pmc_sq", "SQ_INSTS_SALU", prefix)
    waves, _ = mean_counter(d / "pmc_sq", "SQ_WAVES", prefix)
    gui, _ = mean_counter(d / "pmc_sq2", "GRBM_GUI_ACTIVE", prefix)
    out = {"kernel": "fast_kernel", "workload": workload, "images_per_launch": images, "hbm_bytes_per_launch": hbm,
           "valu_insts_per_launch": None if valu is None else int(valu), "salu_insts_per_launch": None if salu is None else int(salu),
           "waves_per_launch": None if waves is None else int(waves), "busy_cycles_per_launch": None if gui is None else int(gui / 8),
           "raw": {"FETCH_SIZE_KB": round(fetch_kb, 1), "WRITE_SIZE_KB": round(write_kb, 1), "dispatches": [nf, nw], "run": d.name},
           "fetch_correction": FETCH_CORRECTION,
           "note": "MI355X_MICROARCH.md section HBM: gfx950 FETCH_SIZE = 1/2 of the bytes of a 16-B-per-lane coalesced read "
                   "(fast_kernel's tile loader is global_load_dwordx4) -> x2; WRITE_SIZE as reported; separate --pmc passes",
           "source": str(SOURCE.relative_to(ROOT)), "source_sha256": hashlib.sha256(SOURCE.read_bytes()).hexdigest()}
    (ROOT / "profiles" / ("fast_kernel_traffic.json" if workload == "euroc" else f"fast_kernel_traffic_{workload}.json")).write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
