# kernel-trace stats of the front-end + post-extraction kernels (one stream), for quick A/B of a single kernel
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sk -o t -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-overlap --distinct 16 --ba-windows 0 --gba-keyframes 0 --pose-frames 0 --track-frames 0 > /tmp/sk.log 2>&1
python - <<PY
import csv,re
for r in csv.DictReader(open("/tmp/sk/t_kernel_stats.csv")):
    n=re.sub(r"\(anonymous namespace\)::","",r["Name"]); n=re.sub(r"^void ","",n).split("(")[0][:50]
    if n.startswith("snk::"): print("%s,%s,%.1f" % (n, r["Calls"], float(r["AverageNs"])/1e3))
PY
