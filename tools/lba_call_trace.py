"""Summarise a `rocprofv3 --kernel-trace --output-format csv` run of tools/lba_call_latency.py: per kernel the calls, the
average duration and the share; and for the LAST fused call (create -> snk_ba_solve_local_scene) the timeline: start offset,
duration and the idle gap before each kernel.  Usage: python tools/lba_call_trace.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("snk::", "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:60]


def main(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    agg = {}
    for r in rows:
        a = agg.setdefault(short(r["Kernel_Name"]), [0, 0])
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = sum(v[1] for v in agg.values())
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:70s} calls {v[0]:6d}  avg {v[1] / v[0] / 1e3:8.2f} us  {100.0 * v[1] / tot:5.1f} %")
    # last call: from the last scene upload to the end
    last = max(i for i, r in enumerate(rows) if "gather_cam_records" in r["Kernel_Name"]) - 1  # the upload before the last list build
    t0 = int(rows[last]["Start_Timestamp"])
    prev_end = t0
    busy = 0
    print("\nlast call (us from the scene upload):")
    for r in rows[last:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += e - s
        print(f"  {((s - t0) / 1e3):8.1f}  dur {((e - s) / 1e3):7.2f}  gap {((s - prev_end) / 1e3):6.2f}  {short(r['Kernel_Name'])}")
        prev_end = e
    print(f"  span {(prev_end - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
