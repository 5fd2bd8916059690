#!/bin/bash
# Runs on the GPU box (via gpurun): PMC passes over the BA leg of bench.py (256 windows).
# usage: tools/profile_ba_pmc.sh <tag>   -> gpurun_out/prof_bapmc_<tag>/
set -u
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_bapmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --batch 8 --no-cpu-baseline --pose-frames 0 --gba-keyframes 0"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- python $REPO/bench.py $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o p -- python $REPO/bench.py $ARGS > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $OUT/pmc_ta -o p -- python $REPO/bench.py $ARGS > $OUT/pmc_ta.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -o p -- python $REPO/bench.py $ARGS > $OUT/pmc_tcc.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o p -- python $REPO/bench.py $ARGS > $OUT/pmc_$C.log 2>&1
done
for d in $OUT/pmc_*; do [ -d $d ] && ls $d | head -3; done
tail -2 $OUT/pmc_ta.log
