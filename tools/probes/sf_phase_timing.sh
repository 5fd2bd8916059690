#!/bin/bash
# Runs on the GPU box: schur_fused without one of its phases at a time (build variants -DSNK_SF_SKIP=1/2/4/8/15 copied to gpurun_tmp/,
# results meaningless), kernel-trace average per variant.   usage: tools/probes/sf_phase_timing.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 4 8 15; do
  if [ $v = 0 ]; then unset SNK_HIP_LIB; else export SNK_HIP_LIB=$REPO/gpurun_tmp/libsnake_hip_sfskip$v.so; fi
  rm -rf /tmp/sfp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sfp -o t -- python $REPO/tools/ba_batch_only.py --windows 1024 --solves 4 > /tmp/sfp.log 2>&1
  python - <<PY
import csv
for r in csv.DictReader(open("/tmp/sfp/t_kernel_stats.csv")):
    if "schur_fused" in r["Name"]: print("skip=$v schur_fused avg_us %.1f" % (float(r["AverageNs"])/1e3))
PY
done
