#!/bin/bash
# Kernel trace of the batch hand-over alone (snk_ba_set_problems on 1024 windows, four times on one handle).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_ba_handover
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $REPO/tools/ba_batch_only.py --windows 1024 --creates 4 --solves 0 > $OUT/trace.log 2>&1
grep "hand-over" $OUT/trace.log
python - <<PY
import csv,re
rows=list(csv.DictReader(open("$OUT/trace/t_kernel_stats.csv")))
print("kernel,calls,avg_us,total_us,pct")
for r in rows[:24]:
    n=re.sub(r"\(anonymous namespace\)::","",r["Name"]); n=re.sub(r"^void ","",n); n=n.split("(")[0][:60]
    print(f'{n},{r["Calls"]},{float(r["AverageNs"])/1e3:.1f},{float(r["TotalDurationNs"])/1e3:.0f},{r["Percentage"]}')
PY
