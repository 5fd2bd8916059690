#!/bin/bash
# Runs on the GPU box: FETCH_SIZE / WRITE_SIZE of tools/probes/hbm_counter_probe (known byte counts) -> gpurun_out/<tag>/hbm_counter_probe.txt
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/hbmprobe_$C -o p -- $REPO/tools/probes/hbm_counter_probe > $OUT/hbmprobe_$C.log 2>&1
done
python - <<PY
import csv, glob, json, re
from collections import defaultdict
want = json.loads([l for l in open("$OUT/hbmprobe_WRITE_SIZE.log") if l.startswith("{")][0])["bytes"]
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    per = defaultdict(lambda: defaultdict(float))
    for f in glob.glob("$OUT/hbmprobe_%s/**/*counter_collection.csv" % C, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == C:
                per[re.sub(r"^void |\(.*$", "", r["Kernel_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, d in sorted(per.items()):
        v = sorted(d.values())[-1] * 1024  # KB -> bytes, the warm repetition
        w = [b for n, b in want.items() if n.split("<")[0] in k and (("<" not in n) or n.split("<")[1].split(">")[0].replace(" ", "") in k.replace(" ", ""))]
        print(f"{C:11s} {k:40s} counted {v / 1e6:10.1f} MB   moved {w[0] / 1e6 if w else float('nan'):10.1f} MB   ratio {v / w[0] if w else float('nan'):.3f}")
PY
rm -rf $OUT/hbmprobe_FETCH_SIZE $OUT/hbmprobe_WRITE_SIZE
