#!/bin/bash
# Register / LDS / occupancy of every kernel of one source, from the compiler's resource-usage remarks (no GPU needed):
#   tools/kernel_resources.sh ba.hip [extra hipcc flags] | grep -A9 schur_fused
cd "$(dirname "$0")/.."
src=$1; shift
extra=""
[ "$src" = "ba.hip" ] && extra="-ffp-contract=fast"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -Isnake_slam_amd/csrc $extra "$@" \
  --cuda-device-only -c snake_slam_amd/csrc/$src -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "remark:" | sed -E 's/.*remark: [^ ]+ //; s/^.*\[-Rpass-analysis.*$//' | grep -v "^$"
