#!/bin/bash
# Runs on the GPU box (round 5, review item 1c): blur inside describe_kernel (SNK_ORB_BLUR_IN_DESCRIBE=1, an experiment: level_kernel skips the
# store of the blurred level, the descriptor wavefront blurs each keypoint's raw 43 x 48 window in LDS) against the default, end to end.
# First the check that the experiment computes the right thing where it claims to (keypoints >= 21 px inside their level), then bench.py's
# front-end leg, alternating, one lease.
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<'PY'
import os, subprocess, sys, json
import numpy as np
code = r"""
import numpy as np, sys
from snake_slam_amd import synth
from snake_slam_amd.orb import ORBExtractor
img, _ = synth.stereo_frame(3, 752, 480)
ext = ORBExtractor(1000, 1.2, 4, 20, 7)
k, d = ext.Detect(img)
np.save(sys.argv[1], np.concatenate([k['x'][:, None].astype(np.float64), k['y'][:, None], k['octave'][:, None], d.view(np.uint8).reshape(len(k), 32)], 1))
"""
outs = []
for env in ({}, {"SNK_ORB_BLUR_IN_DESCRIBE": "1"}):
    f = "/tmp/bid_%d.npy" % len(outs)
    subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **env), check=True, stderr=subprocess.DEVNULL)
    outs.append(np.load(f))
a, b = outs
assert a.shape == b.shape and np.array_equal(a[:, :3], b[:, :3]), "the keypoints themselves must not change"
scale = 1.2 ** a[:, 2]
w, h = np.round(752 / scale), np.round(480 / scale)
x, y = a[:, 0] / scale, a[:, 1] / scale
interior = (x >= 21.5) & (y >= 21.5) & (x <= w - 22.5) & (y <= h - 22.5)
same = (a[:, 3:] == b[:, 3:]).all(1)
print(f"keypoints {len(a)}, interior {int(interior.sum())}, identical descriptors: interior {int((same & interior).sum())}, border {int((same & ~interior).sum())} of {int((~interior).sum())}")
PY
ARGS="--ba-windows 0 --gba-keyframes 0 --pose-frames 0 --track-frames 0 --frame-calls 0 --kitti-steps 0 --no-cpu-baseline"
for rep in 1 2; do
  for V in 0 1; do
    if [ $V = 1 ]; then export SNK_ORB_BLUR_IN_DESCRIBE=1; else unset SNK_ORB_BLUR_IN_DESCRIBE; fi
    python bench.py $ARGS 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('blur_in_describe', $V, d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
  done
done
