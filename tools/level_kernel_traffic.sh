#!/bin/bash
# Runs on the GPU box: FETCH_SIZE / WRITE_SIZE of every level_kernel dispatch of a short bench run, dispatch by dispatch (round 5: the
# round-4 summary averaged over "big" dispatches and dropped the smallest level from the mean).   usage: tools/level_kernel_traffic.sh <tag>
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-overlap --distinct 16 --ba-windows 0 --frame-calls 0 --gba-keyframes 0 --pose-frames 0 --track-frames 0 --kitti-steps 0"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/lk_$C -o p -- python $REPO/bench.py $ARGS > $OUT/lk_$C.log 2>&1
done
python - <<PY
import csv, glob, re
from collections import defaultdict
B = 2048
# planes per image: blurred level l (w*h) + level l+1; E-stereo pyramid
dims = [(752, 480), (627, 400), (522, 333), (435, 278)]
model_w = [dims[l][0] * dims[l][1] + (dims[l + 1][0] * dims[l + 1][1] if l < 3 else 0) for l in range(4)]
model_r = [dims[l][0] * dims[l][1] for l in range(4)]
for C, model, corr in (("FETCH_SIZE", model_r, 2.0), ("WRITE_SIZE", model_w, 1.0)):
    per = defaultdict(float)
    for f in glob.glob("$OUT/lk_%s/**/*counter_collection.csv" % C, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == C and "level_kernel" in r["Kernel_Name"]:
                per[int(r["Dispatch_Id"])] += float(r["Counter_Value"]) * 1024 * corr
    ids = sorted(per)
    print(C, "(x%.0f)" % corr, len(ids), "level_kernel dispatches")
    for i, d in enumerate(ids):
        l = i % 4
        print(f"  dispatch {d:5d} level {l}: {per[d] / 1e6:9.1f} MB counted, model {model[l] * B / 1e6:9.1f} MB, ratio {per[d] / (model[l] * B):.3f}")
PY
rm -rf $OUT/lk_FETCH_SIZE $OUT/lk_WRITE_SIZE
