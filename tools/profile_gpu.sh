#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes of bench.py.
# usage: tools/profile_gpu.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-x}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-overlap --distinct 16 --ba-windows 0 --frame-calls 0 --gba-keyframes 0 --kitti-steps 0 --harris-steps 0 $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o p -- python $REPO/bench.py $ARGS > $OUT/pmc_$C.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- python $REPO/bench.py $ARGS > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o p -- python $REPO/bench.py $ARGS > $OUT/pmc_sq2.log 2>&1
find $OUT -name "*.csv" | head -30
tail -2 $OUT/trace.log
