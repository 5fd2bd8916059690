#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC of bench.py's tracking leg (device-resident coarse + fine matchers).
# usage: tools/profile_track.sh <tag> [pmc]   -> gpurun_out/prof_track_<tag>/
set -u
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_track_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-overlap --distinct 16 --ba-windows 0 --pose-frames 0 --frame-calls 0 --gba-keyframes 0 --kitti-steps 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
if [ "${2:-}" = "pmc" ]; then
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- $CMD > $OUT/pmc_sq.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o p -- $CMD > $OUT/pmc_sq2.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o p -- $CMD > $OUT/pmc_$C.log 2>&1
  done
fi
python - <<PY
import csv,re
rows=list(csv.DictReader(open("$OUT/trace/t_kernel_stats.csv")))
print("kernel,calls,avg_us,total_us,pct")
for r in rows:
    n=re.sub(r"\(anonymous namespace\)::","",r["Name"]); n=re.sub(r"^void ","",n); n=n.split("(")[0][:60]
    if n.startswith("snk::"): print(f'{n},{r["Calls"]},{float(r["AverageNs"])/1e3:.1f},{float(r["TotalDurationNs"])/1e3:.0f},{r["Percentage"]}')
PY
