"""distribute_kernel phase by phase (SNK_ORB_DIST_TIMING=1: cycle sums of thread 0 per phase and level, printed by the library after a
synchronisation) on a batch of the bench's textured frames.   usage (GPU box): python tools/probes/dist_phase_timing.py [batch]"""
import os
import sys

os.environ["SNK_ORB_DIST_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from snake_slam_amd import synth  # noqa: E402
from snake_slam_amd.orb import ORBExtractor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ex = ORBExtractor(1000, 1.2, 4, 20, 7)
ex.configure(752, 480, B)
cap = ex.max_keypoints()
frames = synth.stereo_frames(list(range(16)), 752, 480)
left = np.stack([f[0] if isinstance(f, (tuple, list)) else f["left"] for f in frames])
pitch = 768
buf = np.zeros((B, 480, pitch), np.uint8)
for i in range(B):
    buf[i, :, :752] = left[i % len(left)]
img = torch.from_numpy(buf).cuda()
kps = torch.zeros((B, cap, 24), dtype=torch.uint8, device="cuda")
desc = torch.zeros((B, cap, 4), dtype=torch.int64, device="cuda")
n = torch.zeros((B,), dtype=torch.int32, device="cuda")
for _ in range(2):
    ex.detect_batch_dev(img, kps, desc, n)
    ex.sync()
print("keypoints per image", float(n.float().mean()))
