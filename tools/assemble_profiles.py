#!/usr/bin/env python3
"""Copies the summaries of one measurement round from gpurun_out/ (scratch) into profiles/r03/ (tracked; SNK_PROFILE_ROUND selects the directory).

usage: assemble_profiles.py <tag>      e.g. r03h -> profiles/r03/r03h_*  (+ profiles/fast_kernel_traffic.json)

Expects what tools/profile_gpu.sh, profile_track.sh, profile_ba.sh and the bench / pytest commands of a round leave under
gpurun_out/prof_<tag>, prof_track_<tag>, prof_ba_<tag> and gpurun_out/<tag>/; whatever is absent is skipped."""
import csv
import os
import re
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
G = ROOT / "gpurun_out"
P = ROOT / "profiles" / os.environ.get("SNK_PROFILE_ROUND", "r03")


def kstats(src: Path, dst: Path):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w") as f:
        f.write("kernel,calls,avg_us,min_us,max_us,total_us,pct\n")
        for r in rows:
            n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
            n = re.sub(r"^void ", "", n).split("(")[0]
            if n.startswith("snk::"):
                f.write(f'"{n}",{r["Calls"]},{float(r["AverageNs"]) / 1e3:.1f},{float(r["MinNs"]) / 1e3:.1f},'
                        f'{float(r["MaxNs"]) / 1e3:.1f},{float(r["TotalDurationNs"]) / 1e3:.0f},{r["Percentage"]}\n')


def main(tag):
    P.mkdir(parents=True, exist_ok=True)
    for d, name in ((f"prof_{tag}", "bench_b1024"), (f"prof_track_{tag}", "bench_with_tracking_chain"), (f"prof_ba_{tag}", "ba_b1024_only")):
        src = G / d / "trace" / "t_kernel_stats.csv"
        if src.exists():
            kstats(src, P / f"{tag}_kernel_stats_{name}.csv")
            subprocess.run([sys.executable, str(ROOT / "tools" / "pmc_summary.py"), str(G / d), str(P / f"{tag}_pmc_{name}.csv")],
                           stdout=subprocess.DEVNULL, check=False)
    for f in ("bench_default.json", "bench_sequence.json", "bench_kitti.json", "lba_call_latency.json", "pytest_gpu.log"):
        if (G / tag / f).exists():
            shutil.copy(G / tag / f, P / f"{tag}_{f}")
    t = G / tag / "fast_kernel_traffic.json"
    if t.exists():
        shutil.copy(t, ROOT / "profiles" / "fast_kernel_traffic.json")
    t = G / tag / "pipeline_traffic.json"
    if t.exists():
        shutil.copy(t, ROOT / "profiles" / "pipeline_traffic.json")
    for f in ("latencies.log",):
        if (G / tag / f).exists():
            shutil.copy(G / tag / f, P / f"{tag}_{f}")
    print("\n".join(sorted(p.name for p in P.glob(f"{tag}_*"))))


if __name__ == "__main__":
    main(sys.argv[1])
