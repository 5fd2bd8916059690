# tracking-chain throughput with / without the pose kernel's LDS-resident matches
F="--no-cpu-baseline --ba-windows 0 --gba-keyframes 0 --pose-frames 0 --distinct 16"
for v in lds nolds; do
  if [ $v = nolds ]; then export SNK_POSE_NO_LDS=1; else unset SNK_POSE_NO_LDS; fi
  python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['tracking']['value'], d['tracking'].get('ms_per_step'), d['tracking'].get('identical_to_gpu'))"
done
