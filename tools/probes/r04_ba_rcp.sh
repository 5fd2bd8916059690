# BA with rcp_nr / rsqrt_nr in the linearisation (default) against the IEEE divisions / roots (variant ba_ieee): bench.py's BA leg, alternating.
B="python bench.py --no-cpu-baseline --gba-keyframes 0 --pose-frames 0 --check-frames 0 --frame-calls 0 --track-frames 0 --steps 10 --warmup 3"
run() { echo "== $1"; SNK_HIP_LIB=$2 $B 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['ba']; print(t['value'], t['ms_per_step'], t['cost_final'], t['single_window_ms_per_solve'])"; }
for rep in 1 2; do run default ""; run ba_ieee snake_slam_amd/lib/variants/libsnake_hip_ba_ieee.so; done
