#!/bin/bash
# Every fuzzer against the oracle on the current sources.   usage (GPU box): tools/fuzz_all.sh [seconds each, default 90] [seed, default 7] > gpurun_out/<tag>/fuzz_all.log
S=${1:-90}
SEED=${2:-7}
cd "$(dirname "$0")/.."
rc=0
for f in fuzz_orb fuzz_orb_batch fuzz_match fuzz_match_batch fuzz_track_batch fuzz_ba_pose fuzz_frontend; do
  timeout $((S * 4 + 200)) python tools/$f.py --seconds "$S" --seed "$SEED" 2>&1 | grep -v amdgpu.ids | tail -2 || rc=1
done
SNK_ORB_LEVEL_BH=22 timeout $((S * 4 + 200)) python tools/fuzz_orb.py --seconds $((S / 2)) --seed $((SEED + 1)) 2>&1 | grep -v amdgpu.ids | tail -1
SNK_ORB_LEVEL_BH=64 timeout $((S * 4 + 200)) python tools/fuzz_orb_batch.py --seconds $((S / 2)) --seed $((SEED + 2)) 2>&1 | grep -v amdgpu.ids | tail -1
timeout $((S * 8 + 300)) python tools/fuzz_ba_pose.py --ba-only --big-batches --seconds "$S" --seed $((SEED + 3)) 2>&1 | grep -v amdgpu.ids | tail -1
exit $rc
