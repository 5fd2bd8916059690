#!/bin/bash
# Runs on the GPU box: kernel-trace stats of the global-BA leg (FullBA(4) on 300 keyframes x 15 000 points x 10 observations, tools/gba_trace.py).
# usage: tools/profile_gba.sh <tag>   -> gpurun_out/<tag>/gba_kernel_stats.csv, gba_timeline.txt
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gba_trace -o t -- python $REPO/tools/gba_trace.py > $OUT/gba_trace.log 2>&1
python - <<PY
import csv, re
rows = list(csv.DictReader(open("$OUT/gba_trace/t_kernel_stats.csv")))
with open("$OUT/gba_kernel_stats.csv", "w") as f:
    f.write("kernel,calls,avg_us,total_us,pct\n")
    for r in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"^void ", "", n).split("(")[0][:60]
        f.write("%s,%s,%.1f,%.0f,%s\n" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
print(open("$OUT/gba_kernel_stats.csv").read())
# the LAST solve: span, busy time, gaps
tr = sorted(csv.DictReader(open("$OUT/gba_trace/t_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
last_init = max(i for i, r in enumerate(tr) if "pcgl_init" in r["Kernel_Name"] or "pcgl_persist" in r["Kernel_Name"])
# walk back to the start of that solve: 4 LM iterations -> the 4th pcg start from the end
starts = [i for i, r in enumerate(tr) if "pcgl_init" in r["Kernel_Name"] or "pcgl_persist" in r["Kernel_Name"]]
first = starts[-4] if len(starts) >= 4 else starts[0]
# include the linearisation kernels before the first PCG of the solve
while first > 0 and int(tr[first]["Start_Timestamp"]) - int(tr[first - 1]["End_Timestamp"]) < 200000:
    first -= 1
seg = tr[first:]
t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"last solve: {len(seg)} launches, span {(t1 - t0) / 1e6:.2f} ms, kernels busy {busy / 1e6:.2f} ms, idle between launches {(t1 - t0 - busy) / 1e6:.2f} ms")
PY
cat $OUT/gba_trace.log | grep -v "^W2026\|^E2026" | tail -5
rm -rf $OUT/gba_trace
# counter bytes of the same run (second argument "pmc"): separate passes per counter, read by tools/collect_pipeline_traffic.py
if [ "${2:-}" = "pmc" ]; then
  P=$REPO/gpurun_out/prof_gba_$TAG
  mkdir -p $P
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/pmc_$C -o p -- python $REPO/tools/gba_trace.py > $P/pmc_$C.log 2>&1
  done
fi
