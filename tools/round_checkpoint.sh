#!/bin/bash
# One measurement checkpoint of a round on the GPU box: full GPU test suite, default bench line (+ KITTI), kernel stats and PMC
# passes of the front-end, the tracking chain and the 1024-window BA, the per-frame latencies.   usage: tools/round_checkpoint.sh <tag> [notest]
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/$TAG
if [ "${2:-}" != "notest" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q -rf --tb=short 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/pytest_gpu.log
  tail -3 gpurun_out/$TAG/pytest_gpu.log
fi
timeout 600 python bench.py 2>&1 | grep '"metric"' > gpurun_out/$TAG/bench_default.json
timeout 600 python bench.py --workload kitti --ba-windows 0 --gba-keyframes 0 --pose-frames 0 --track-frames 0 --kitti-steps 0 2>&1 | grep '"metric"' > gpurun_out/$TAG/bench_kitti.json
bash tools/profile_gpu.sh $TAG > gpurun_out/$TAG/profile_gpu.log 2>&1
python tools/collect_traffic.py gpurun_out/prof_$TAG 2048 > gpurun_out/$TAG/collect_traffic.log 2>&1 && cp profiles/fast_kernel_traffic.json gpurun_out/$TAG/fast_kernel_traffic.json
# the KITTI leg's own counters (round 6: no longer the EuRoC pass scaled by the image count): 512 stereo frames = 1024 images per launch
bash tools/profile_gpu.sh ${TAG}_kitti --workload kitti --batch 512 --pose-frames 0 --track-frames 0 > gpurun_out/$TAG/profile_gpu_kitti.log 2>&1
python tools/collect_traffic.py gpurun_out/prof_${TAG}_kitti 1024 snk::fast_kernel kitti > gpurun_out/$TAG/collect_traffic_kitti.log 2>&1 && cp profiles/fast_kernel_traffic_kitti.json gpurun_out/$TAG/fast_kernel_traffic_kitti.json
bash tools/profile_track.sh $TAG pmc > gpurun_out/$TAG/profile_track.log 2>&1
bash tools/profile_ba.sh $TAG pmc > gpurun_out/$TAG/profile_ba.log 2>&1
bash tools/profile_gba.sh $TAG pmc > gpurun_out/$TAG/profile_gba.log 2>&1
python tools/collect_pipeline_traffic.py gpurun_out/prof_$TAG gpurun_out/prof_ba_$TAG 1024 1024 > gpurun_out/$TAG/collect_pipeline_traffic.log 2>&1 && cp profiles/pipeline_traffic.json gpurun_out/$TAG/pipeline_traffic.json
(python tools/latency_frontend.py; python tools/frontend_latency_cpp.py; python tools/latency.py; python tools/lba_call_latency_cpp.py) 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/latencies.log
python -c "
import json
d=json.load(open('gpurun_out/$TAG/bench_default.json'))
print('front-end', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('stage_ms_per_step'))
print('ba', d['ba']['value'], 'tracking', d['tracking']['value'])
"
tail -5 gpurun_out/$TAG/latencies.log
# BASELINE config 5 on every GPU of this node (one here): native RCCL gather + bench.py --gpus N --mode sequence
timeout 900 python tools/run_all_gpus.py --out gpurun_out/$TAG/all_gpus --skip-batch > gpurun_out/$TAG/all_gpus.log 2>&1; tail -1 gpurun_out/$TAG/all_gpus.log
