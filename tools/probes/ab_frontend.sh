#!/bin/bash
# Same-box A/B of two builds of the library on the front-end batch leg: gpurun_tmp/libsnake_hip_A.so against the tree's library, alternating.
# usage (GPU box): tools/probes/ab_frontend.sh [pairs, default 3]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-3}
cd $REPO
run() {
  timeout 300 python bench.py --steps 40 --ba-windows 0 --gba-keyframes 0 --pose-frames 0 --track-frames 0 --frame-calls 0 --kitti-steps 0 --harris-steps 0 --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']
print('$1', round(d['value']), s['pyramid'], s['fast'], s['distribute'], s['describe'])"
}
for i in $(seq $N); do
  SNK_HIP_LIB=$REPO/gpurun_tmp/libsnake_hip_A.so run A
  run B
done
