mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/d_smoke.log
B="python bench.py --no-cpu-baseline --ba-windows 0 --gba-keyframes 0 --pose-frames 0 --track-frames 0 --check-frames 0 --steps 20 --warmup 5"
for cfg in "0 0" "0 2" "27648 2" "27648 4" "27648 8" "33792 4" "41984 4" "33792 8"; do set -- $cfg; echo "== lds_wg $1 stagger $2"; SNK_ORB_FAST_LDS_WG=$1 $B --orb-stagger $2 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step'))"; done > gpurun_out/r04/e_stagger_lds.log 2>&1
cat gpurun_out/r04/d_smoke.log gpurun_out/r04/e_stagger_lds.log
