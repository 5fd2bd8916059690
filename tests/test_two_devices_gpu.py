"""GPU (needs two devices; skipped on a one-GPU box): handles on different devices in ONE process.  The opt-in to more than 64 KB
of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize) belongs to a kernel ON a device; round 2 cached it per kernel only,
so handles on the second device launched the large-carve kernels (stereo_frame_kernel, the frame-resident projection
matchers, pose_kernel, fast_kernel, distribute) without it.  set_max_lds_once now keys by (device, kernel)."""
import numpy as np
import pytest

from helpers import SEED, make_stereo_case

pytestmark = pytest.mark.gpu


def test_large_lds_kernels_on_two_devices(orc):
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU")
    from snake_slam_amd import synth
    from snake_slam_amd.matcher import KP64_DTYPE, Preprocess
    from snake_slam_amd.orb import ORBExtractor

    rng = np.random.default_rng(SEED + 77)
    img, _ = synth.stereo_frame(3, 320, 240, n_rects=100)
    wk, wd = orc.orb_detect(orc.orb_params(300, 1.2, 3, 20, 7), img)
    B, cap = 8, 1024  # >= 8 frames: stereo_frame_kernel (one workgroup per frame, the right side in a > 64 KB LDS carve)
    cases = [make_stereo_case(rng, 900, 880) for _ in range(B)]
    for dev in (1, 0, 1):  # the second device first: its attribute must not be skipped because device 0's was set, and vice versa
        ext = ORBExtractor(300, 1.2, 3, 20, 7, device=dev)       # fast_kernel / distribute_kernel carves
        k, d = ext.Detect(img)
        ext.close()
        assert np.array_equal(k, wk) and np.array_equal(d, wd), dev
        t = torch.device("cuda", dev)
        pre = Preprocess(dev)

        kl, kr = np.zeros((B, cap), KP64_DTYPE), np.zeros((B, cap), KP64_DTYPE)
        dl, dr = np.zeros((B, cap, 4), np.uint64), np.zeros((B, cap, 4), np.uint64)
        for b, (l, a, r, c, _, _) in enumerate(cases):
            kl[b, :len(l)], dl[b, :len(l)], kr[b, :len(r)], dr[b, :len(r)] = l, a, r, c
        to = lambda a: torch.from_numpy(a.view(np.uint8).reshape(a.shape[0], a.shape[1], -1)).to(t)
        nl = torch.full((B,), 900, dtype=torch.int32, device=t)
        nr = torch.full((B,), 880, dtype=torch.int32, device=t)
        rp = torch.full((B, cap), -1000.0, dtype=torch.float32, device=t)
        dp = torch.full((B, cap), -1000.0, dtype=torch.float32, device=t)
        ns = torch.zeros(B, dtype=torch.int32, device=t)
        ls = cases[0][5]
        with torch.cuda.device(t):
            torch.cuda.synchronize(t)  # the inputs were produced on torch's stream, the matcher has its own
            pre.match_batch_dev(to(kl), torch.from_numpy(dl.view(np.int64)).to(t), nl, to(kr), torch.from_numpy(dr.view(np.int64)).to(t), nr,
                                cases[0][4], ls, True, rp, dp, ns)
            torch.cuda.synchronize(t)
        for b, (l, a, r, c, bfv, _) in enumerate(cases):
            wn, wrp, wdp = orc.stereo_match(l, a, r, c, bfv, ls, True)
            assert int(ns[b].item()) == wn and np.array_equal(rp[b, :900].cpu().numpy(), wrp) and np.array_equal(dp[b, :900].cpu().numpy(), wdp), (dev, b)
        pre.close()
