"""Summarise a `rocprofv3 --kernel-trace --output-format csv` run of tools/frontend_latency_cpp.py: the steady state of the pipelined
per-frame front-end (the last ~400 frames): kernels per frame, GPU busy time (union of kernel intervals) per frame, the sum of kernel
durations per frame (more than the union when frames in different slots overlap on the device), and per kernel its average duration.
Usage: python tools/frontend_pipeline_trace.py <dir with *_kernel_trace.csv> [frames_in_tail]"""
import csv
import glob
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("snk::", "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:50]


def main(d, tail_frames=400):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # a frame = one describe_kernel launch
    desc = [i for i, r in enumerate(rows) if "describe_kernel" in r["Kernel_Name"]]
    first = desc[-tail_frames - 1] + 1
    tail = rows[first:desc[-1] + 1]
    t0, t1 = int(tail[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in tail)
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in tail)
    union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    total = sum(e - s for s, e in iv)
    print(f"tail: {tail_frames} frames in {(t1 - t0) / 1e6:.2f} ms = {tail_frames / ((t1 - t0) / 1e9):.0f} frames/s; per frame: span {(t1 - t0) / 1e3 / tail_frames:.1f} us, "
          f"GPU busy (union) {union / 1e3 / tail_frames:.1f} us, sum of kernel durations {total / 1e3 / tail_frames:.1f} us, kernels {len(tail) / tail_frames:.1f}")
    agg = {}
    for r in tail:
        a = agg.setdefault(short(r["Kernel_Name"]), [0, 0])
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:52s} per frame {v[0] / tail_frames:5.2f} x {v[1] / v[0] / 1e3:7.2f} us = {v[1] / 1e3 / tail_frames:7.2f} us")
    # one frame's timeline (the last)
    last = rows[desc[-2] + 1:desc[-1] + 1]
    b = int(last[0]["Start_Timestamp"])
    prev = b
    print("last frame (us):")
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  {(s - b) / 1e3:8.1f}  dur {(e - s) / 1e3:7.2f}  gap {(s - prev) / 1e3:6.2f}  {short(r['Kernel_Name'])}")
        prev = e


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 400)
