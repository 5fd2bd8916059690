#!/bin/bash
# Same-box A/B of ONE library under an environment switch on the front-end batch leg, alternating runs:
#   tools/probes/ab_env.sh SNK_ORB_DESC_DMA=1 [pairs, default 3]      (A = switch set, B = default)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
SW=$1
N=${2:-3}
cd $REPO
run() {
  timeout 300 env $2 python bench.py --steps 40 --ba-windows 0 --gba-keyframes 0 --pose-frames 0 --track-frames 0 --frame-calls 0 --kitti-steps 0 --harris-steps 0 --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']
print('$1', round(d['value']), s['pyramid'], s['fast'], s['distribute'], s['describe'])"
}
for i in $(seq $N); do
  run "A($SW)" "$SW"
  run "B(default)" "SNK_AB_NONE=1"
done
