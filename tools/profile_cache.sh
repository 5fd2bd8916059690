#!/bin/bash
# Runs on the GPU box (via gpurun): L1 (TCP) / L2 (TCC) request counters of the ORB kernels, one rocprofv3 pass per counter group,
# every pass under its own timeout (a pass with TA_* counters hung on this pool -- not collected).
# usage: tools/profile_cache.sh <tag>   -> gpurun_out/prof_cache_<tag>/   (summarise with tools/cache_summary.py)
set -u
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_cache_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-overlap --ba-windows 0 --gba-keyframes 0 --pose-frames 0 --track-frames 0 --distinct 32"
i=0
for G in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT/g$i -o p -- python $REPO/bench.py $ARGS > $OUT/g$i.log 2>&1
done
python $REPO/tools/cache_summary.py $OUT
