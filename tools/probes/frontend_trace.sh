#!/bin/bash
# rocprofv3 kernel trace of snk_frontend_process (one 752x480 stereo frame per call): per-kernel durations of the call's launch chain
# and the GPU-busy share of a call.   usage (GPU box): tools/probes/frontend_trace.sh <tag>
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r04/fe_trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $REPO/tools/latency_frontend.py --frames 100 > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel,calls,avg_us,total_ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print(f'{r["Name"][:70]},{r["Calls"]},{float(r["AverageNs"]) / 1e3:.1f},{float(r["TotalDurationNs"]) / 1e6:.2f}')
PY
grep -v amdgpu.ids $OUT/run.log | tail -4
