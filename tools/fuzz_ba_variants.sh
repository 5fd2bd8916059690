#!/bin/bash
# The BA fuzzer (batches of 1 .. 300 unequal scenes) under every path switch of tests/test_zy_ba_variants_gpu.py.
#   tools/fuzz_ba_variants.sh [seconds per switch, default 25] > gpurun_out/r04/fuzz_ba_variants.log
S=${1:-25}
cd "$(dirname "$0")/.."
rc=0
for e in "SNK_NONE=0" "SNK_BA_SCHUR_SET_MIN_ITEMS=1" "SNK_BA_NO_SCHUR_SET=1" "SNK_BA_NO_POINT_WAVE=1" "SNK_BA_SCHUR_SET_MIN_ITEMS=1 SNK_BA_NO_GRAPH=1" \
    "SNK_BA_GRAPH_FIRST=1" "SNK_BA_PCG_GENERAL=1" "SNK_BA_SCHUR_SET_MIN_ITEMS=1 SNK_BA_FUSED_K10=1" "SNK_BA_SCHUR_SET_MIN_ITEMS=1 SNK_BA_NO_SCHUR_FUSED=1" \
    "SNK_BA_SCHUR_SET_MIN_ITEMS=1 SNK_BA_NO_SCHUR_MFMA=1" "SNK_BA_CHECK_LISTS=1 SNK_BA_NO_SCHUR_SET=1" "SNK_BA_CHECK_LISTS=1" "SNK_BA_HOST_ENTRIES=1 SNK_BA_NO_SCHUR_SET=1" \
    "SNK_BA_SCHUR_SET_MIN_ITEMS=1 SNK_BA_CAM_SUMS=1" "SNK_BA_PCG_LDS=1" "SNK_BA_NO_BIG_ITEMS=1" "SNK_BA_NO_SCHUR_WIDE=1 SNK_BA_NO_SCHUR_SET=1"; do
  env $e timeout $((S * 6 + 120)) python tools/fuzz_ba_pose.py --ba-only --big-batches --seconds "$S" --seed 4 2>&1 | grep -v amdgpu.ids || rc=1
done
exit $rc
