#!/bin/bash
# The batch hand-over of BA (snk_ba_set_problems on 1024 windows) against the host threads that build its lists, with the builder's own
# section times (SNK_BA_PROFILE_CREATE=1).   usage (GPU box): tools/probes/ba_handover_threads.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
echo "host: $(nproc) logical cores, $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2)"
for T in 0 8 16 32 64; do
  echo "== SNK_BA_HOST_THREADS=$T (0 = default: min(cores, 16))"
  if [ "$T" = 0 ]; then E=""; else E="SNK_BA_HOST_THREADS=$T"; fi
  env $E SNK_BA_PROFILE_CREATE=1 timeout 600 python bench.py --steps 5 --warmup 1 --batch 64 --gba-keyframes 0 --pose-frames 0 --track-frames 0 --frame-calls 0 --kitti-steps 0 --harris-steps 0 --no-cpu-baseline 2> /tmp/ho.err | grep '"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read())['ba']; print({k: d[k] for k in d if 'hand' in k or k in ('value',)})"
  grep "snk_ba_set_problems" /tmp/ho.err | tail -4
done
