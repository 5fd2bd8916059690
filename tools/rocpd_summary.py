#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace as CSV: per-kernel calls, total / average /
min / max duration (us) and share of GPU kernel time.  usage: rocpd_summary.py results.db out.csv"""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([\w:<>, ]+?)\(", name)
    return (m.group(1) if m else name)[:80]


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                      "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct_of_kernel_time"])
        for name, calls, tot, avg, mn, mx in rows:
            w.writerow([short(name), calls, f"{tot / 1e3:.3f}", f"{avg / 1e3:.3f}", f"{mn / 1e3:.3f}", f"{mx / 1e3:.3f}",
                        f"{100.0 * tot / total:.2f}"])
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
