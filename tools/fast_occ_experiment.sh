# FAST occupancy experiment: survivor-list capacity (entries) vs stage time
F="--ba-windows 0 --gba-keyframes 0 --pose-frames 0 --track-frames 0 --no-cpu-baseline --distinct 32"
for cap in ${CAPS:-default 320 384 448 512 640 1300}; do
  if [ $cap = default ]; then unset SNK_ORB_FAST_SURV_CAP; else export SNK_ORB_FAST_SURV_CAP=$cap; fi
  python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cap', d['value'], d['stage_ms_per_step'])"
done
