#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: mean counter value per kernel (per dispatch).
usage: pmc_summary.py <prof_dir> [out.csv]"""
import csv
import re
import sys
from collections import defaultdict
from pathlib import Path


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([\w:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:60]


def main(d, out=None):
    acc = defaultdict(lambda: defaultdict(list))
    for f in Path(d).rglob("*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    counters = sorted({c for k in acc.values() for c in k})
    rows = [["kernel", "dispatches"] + counters]
    for k, cs in sorted(acc.items()):
        if not k.startswith("snk::"):
            continue
        n = max(len(v) for v in cs.values())
        rows.append([k, n] + [f"{sum(cs[c]) / len(cs[c]):.4g}" if c in cs else "" for c in counters])
    # kernel names carry commas (pose_kernel<4, 0, true>): quoted, so that the columns stay aligned for csv readers
    text = "\n".join(",".join(f'"{v}"' if isinstance(v, str) and "," in v else str(v) for v in r) for r in rows)
    print(text)
    if out:
        Path(out).write_text(text + "\n")


if __name__ == "__main__":
    main(*sys.argv[1:])
