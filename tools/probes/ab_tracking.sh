#!/bin/bash
# Same-box A/B of two builds of the library on the tracking-chain and pose-refinement legs: gpurun_tmp/libsnake_hip_A.so against the tree's
# library, alternating.   usage (GPU box): tools/probes/ab_tracking.sh [pairs, default 3]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-3}
cd $REPO
run() {
  timeout 300 python bench.py --steps 10 --ba-windows 0 --gba-keyframes 0 --frame-calls 0 --kitti-steps 0 --harris-steps 0 --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['tracking']; p=d.get('pose_refine',{})
print('$1', 'chain', round(t['value']), 'frames/s', {k: v for k, v in t.items() if k.startswith('ms_') or k.endswith('_ms')}, 'pose_refine', p.get('value'))"
}
for i in $(seq $N); do
  SNK_HIP_LIB=$REPO/gpurun_tmp/libsnake_hip_A.so run A
  run B
done
