#!/bin/bash
# Same-box A/B of the batch hand-over (snk_ba_set_problems on 1024 windows): the handle's parked host threads (default) against
# threads created and joined per pass (SNK_BA_NO_HOST_POOL=1).   usage (GPU box): tools/probes/ab_ba_handover_pool.sh [rounds]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-3}
cd $REPO
run() {
  timeout 600 env $2 SNK_BA_PROFILE_CREATE=1 python bench.py --steps 5 --warmup 1 --batch 64 --gba-keyframes 0 --pose-frames 0 --track-frames 0 --frame-calls 0 --kitti-steps 0 --harris-steps 0 --no-cpu-baseline 2> /tmp/ho.err | grep '"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read())['ba']; print('$1', {k: d[k] for k in d if 'hand' in k})"
  grep "snk_ba_set_problems\] lists [0-9]* us, uploads" /tmp/ho.err | tail -1
}
for i in $(seq $N); do
  run "pool(default)" "SNK_AB_NONE=1"
  run "threads-per-pass" "SNK_BA_NO_HOST_POOL=1"
done
