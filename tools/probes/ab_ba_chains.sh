#!/bin/bash
# Same-box scan of SNK_BA_GRAPH_CHAINS (chains a recorded 1024-window LM sequence is split into) on bench.py's BA leg only.
#   tools/probes/ab_ba_chains.sh "1 2 4" [rounds, default 2]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
VALS=${1:-"1 2 4"}
N=${2:-2}
cd $REPO
run() {
  timeout 300 env SNK_BA_GRAPH_CHAINS=$1 python bench.py --steps 20 --warmup 2 --gba-keyframes 0 --pose-frames 0 --track-frames 0 --frame-calls 0 --kitti-steps 0 --harris-steps 0 --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['ba']
print('chains=$1', round(b['value']), b['ms_per_step'], b['cost_final'], round(d['value']))"
}
for i in $(seq $N); do for v in $VALS; do run $v; done; done
