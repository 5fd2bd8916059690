#!/bin/bash
# Same-box A/B of the Harris leg ("orb.response" = 1 on the EuRoC batch): harris_dense_kernel (default) against the per-cell kernel.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-2}
cd $REPO
run() {
  timeout 300 env $2 python bench.py --steps 20 --ba-windows 0 --gba-keyframes 0 --pose-frames 0 --track-frames 0 --frame-calls 0 --kitti-steps 0 --harris-steps 20 --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['harris']
print('$1', {k: h[k] for k in h if k in ('value','ms_per_step','identical_to_cpu','stage_ms_per_step')}, round(d['value']))"
}
for i in $(seq $N); do
  run "dense(default)" "SNK_AB_NONE=1"
  run "per-cell" "SNK_ORB_HARRIS_PER_CELL=1"
done
