#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the BA leg of bench.py.
# usage: tools/profile_ba.sh <tag>   -> gpurun_out/prof_ba_<tag>/
set -u
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_ba_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $REPO/bench.py --steps 3 --warmup 1 --batch 8 --no-cpu-baseline --pose-frames 0 --gba-keyframes 0 > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log
