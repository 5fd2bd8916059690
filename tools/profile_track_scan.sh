#!/bin/bash
# GPU box: instruction counters and durations of the frame-resident matchers (coarse_frame_kernel / fine_frame_kernel) on a 256-frame tracking
# leg under the current environment (A/B switches: SNK_TRACK_*), and a dump of the leg's inputs for tools/track_scan_stats.py.
#   usage: tools/profile_track_scan.sh <outdir>
OUT=$(realpath -m "${1:-gpurun_out/track_scan}"); mkdir -p "$OUT"
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp; export TMPDIR=/tmp
B="python $ROOT/bench.py --ba-windows 0 --gba-keyframes 0 --pose-frames 0 --frame-calls 0 --kitti-steps 0 --no-cpu-baseline --steps 3 --warmup 1 --batch 256 --track-frames 256"
timeout 200 $B --dump-track-inputs "$OUT/track_inputs.npz" > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ts -o t -- $B > /dev/null 2>&1
grep -E "Name|frame_kernel|pose_kernel|resolve|gather" /tmp/ts/t_kernel_stats.csv > "$OUT/stats.csv"
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/tp_$tag -o t -- $B > /dev/null 2>&1
  python - /tmp/tp_$tag/t_counter_collection.csv >> "$OUT/pmc.txt" <<'PY'
import collections, csv, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(\w+_frame_kernel(<[^>]*>)?)", r["Kernel_Name"])
    if not m: continue
    acc[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(m.group(1), r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: round(v / cnt[(k, c)]) for c, v in acc[k].items()})
PY
done
cat "$OUT/pmc.txt"
