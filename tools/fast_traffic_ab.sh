#!/bin/bash
# Runs on the GPU box: FETCH_SIZE of fast_kernel and its launch time by cells per wavefront (SNK_ORB_FAST_CPW).  The run that is
# kept in profiles/r03/r03q_fast_traffic_ab.log also had a column-major processing order of the cells (an experiment that was
# reverted: 3.1 x the algorithmic bytes); its "row_*" lines are what this script measures.
# usage: tools/fast_traffic_ab.sh   -> prints "variant fetch_KB_per_launch avg_us"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-overlap --distinct 16 --ba-windows 0 --gba-keyframes 0 --pose-frames 0 --track-frames 0"
run() {  # name, env assignments...
  name=$1; shift
  rm -rf /tmp/ft_$name
  env "$@" timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/ft_$name -o p -- python $REPO/bench.py $ARGS > /tmp/ft_$name.log 2>&1
  python - <<PY
import csv,glob,collections
v=collections.defaultdict(float); t=collections.defaultdict(list)
for f in glob.glob("/tmp/ft_$name/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fast_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE": v[r["Dispatch_Id"]]+=float(r["Counter_Value"])
for f in glob.glob("/tmp/ft_$name/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fast_kernel" in r["Kernel_Name"]: t[0].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
vals=list(v.values())
print("$name", "FETCH_KB %.0f" % (sum(vals)/max(1,len(vals))), "x2/alg %.3f" % (2*1024*sum(vals)/max(1,len(vals))/1856544768), "avg_us %.1f" % (sum(t[0])/max(1,len(t[0]))))
PY
}
run row_cpw8 SNK_ORB_FAST_CPW=8
run row_cpw1 SNK_ORB_FAST_CPW=1
run row_cpw4 SNK_ORB_FAST_CPW=4
