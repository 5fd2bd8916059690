#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the reference's per-keyframe local-BA call (tools/lba_call_latency.py:
# single-window launches).   usage: tools/profile_lba_call.sh <tag>  -> gpurun_out/prof_lba_<tag>/
set -u
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_lba_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LBA_SCENES=6 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $REPO/tools/lba_call_latency.py > $OUT/trace.log 2>&1
python - <<PY
import csv,re
rows=list(csv.DictReader(open("$OUT/trace/t_kernel_stats.csv")))
print("kernel,calls,avg_us,total_us,pct")
for r in rows:
    n=re.sub(r"\(anonymous namespace\)::","",r["Name"]); n=re.sub(r"^void ","",n).split("(")[0][:60]
    print("%s,%s,%.1f,%.0f,%s" % (n, r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3, r["Percentage"]))
PY
