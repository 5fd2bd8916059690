"""Seeded fuzz of the BATCHED, device-resident extractor (snk_orb_detect_batch_dev) against the oracle: random batch sizes around the
XCD-mapping threshold (1 ... 40), image sizes, row pitches and base offsets (aligned and not), parameters, launch chains.  Every image
of a batch must come out bit-exactly as the oracle extracts it alone.

    python tools/fuzz_orb_batch.py [--seconds 120] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc  # noqa: E402
from snake_slam_amd import synth  # noqa: E402
from snake_slam_amd.orb import KEYPOINT_DTYPE, ORBExtractor  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    orc.build()
    rng = np.random.default_rng(a.seed)
    dev = torch.device("cuda:0")
    t0, n_b, n_img = time.time(), 0, 0
    while time.time() - t0 < a.seconds:
        B = int(rng.choice([1, 2, 7, 8, 15, 16, 17, 24, 33, 40]))
        W, H = int(rng.integers(48, 700)), int(rng.integers(48, 500))
        aligned = rng.random() < 0.6
        P = (W + int(rng.integers(0, 9)) * 4 + 3) & ~3 if aligned else W + int(rng.integers(0, 13))
        off = 0 if aligned else int(rng.integers(1, 4))
        nfeat, levels = int(rng.integers(30, 1500)), int(rng.integers(1, 8))
        scale = float(rng.choice([1.1, 1.2, 1.3, 1.5, 2.0]))
        ini, mn = int(rng.integers(10, 40)), int(rng.integers(3, 9))
        imgs = []
        for i in range(B):
            k = int(rng.integers(0, 4))
            if k == 0:
                imgs.append(rng.integers(0, 256, (H, W), dtype=np.uint8))
            elif k == 1:
                imgs.append(np.full((H, W), int(rng.integers(0, 256)), np.uint8))
            else:
                imgs.append(synth.stereo_frame(int(rng.integers(0, 100000)), W, H, n_rects=int(rng.integers(10, 300)))[0])
        host = np.zeros((B, H, P), np.uint8)
        for i, im in enumerate(imgs):
            host[i, :, :W] = im
        flat = torch.zeros(B * H * P + 16, dtype=torch.uint8, device=dev)
        view = flat[off:off + B * H * P].view(B, H, P)
        view.copy_(torch.from_numpy(host))
        ext = ORBExtractor(nfeat, scale, levels, ini, mn)
        try:
            cap = ext.configure(W, H, B)
            if B >= 8 and rng.random() < 0.3:
                ext.set_chains(2)
            d_kps = torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev)
            d_desc = torch.zeros((B, cap, 4), dtype=torch.int64, device=dev)
            d_n = torch.zeros(B, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            ext.detect_batch_dev(view, d_kps, d_desc, d_n)
            ext.sync()
        finally:
            ext.close()
        n = d_n.cpu().numpy()
        kps = d_kps.cpu().numpy().view(KEYPOINT_DTYPE).reshape(B, cap)
        desc = d_desc.cpu().numpy().view(np.uint64)
        p = orc.orb_params(nfeat, scale, levels, ini, mn)
        for i in range(B):
            wk, wd = orc.orb_detect(p, imgs[i])
            if not (n[i] == len(wk) and np.array_equal(kps[i, : n[i]], wk.astype(KEYPOINT_DTYPE)) and np.array_equal(desc[i, : n[i]], wd)):
                print(f"MISMATCH batch {n_b} image {i}: B {B} {W}x{H} pitch {P} offset {off} nfeat {nfeat} levels {levels} scale {scale} "
                      f"th {ini}/{mn}: {n[i]} vs {len(wk)} keypoints")
                return 1
        n_b += 1
        n_img += B
    print(f"fuzz_orb_batch: {n_b} batches, {n_img} images, all bit-exact (seed {a.seed}, {time.time() - t0:.0f} s)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
