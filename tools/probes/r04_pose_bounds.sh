# Where pose_kernel<4, 0, true> spends a Gauss-Newton step (round 4): variant libraries of pose.hip, tracking leg of bench.py.
# stub1 / stub2 are TIMING-ONLY (the solve replaced by a diagonal one / nothing: results meaningless).
# Build first (CPU): python tools/probes/r04_pose_bounds.py
B="python bench.py --no-cpu-baseline --ba-windows 0 --gba-keyframes 0 --pose-frames 0 --check-frames 0 --frame-calls 0 --steps 10 --warmup 3"
run() { echo "== $1"; SNK_HIP_LIB=$2 $B 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['tracking']; print(t['value'], t['ms_per_step'], t['pose_inliers_per_frame'])"; }
V=snake_slam_amd/lib/variants
for rep in 1 2; do
run default ""
for v in "$@"; do run $v $V/libsnake_hip_$v.so; done
done
