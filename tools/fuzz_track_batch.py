"""Seeded fuzz of the BATCHED, device-resident projection matchers (snk_match_project_coarse_batch_dev -> mark_taken -> fine) against
the oracle frame by frame: random batch sizes, features per frame (0 ... 3000: with and without the frame staged in LDS), points per
frame (few ... thousands: every points-per-wavefront choice), radii, thresholds, directions.

    python tools/fuzz_track_batch.py [--seconds 120] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import track_helpers as T  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from snake_slam_amd.tracking import KP64_DTYPE, LM_COARSE_DTYPE, LM_FINE_DTYPE, SnakeORBMatcher, frames_dev  # noqa: E402


def pack(frames, cap):
    B = len(frames)
    ncell = frames[0]["cols"] * frames[0]["rows"] + 1
    kps, desc = np.zeros((B, cap), KP64_DTYPE), np.zeros((B, cap, 4), np.uint64)
    rp, taken = np.full((B, cap), -1.0, np.float32), np.zeros((B, cap), np.uint8)
    cs, n = np.zeros((B, ncell), np.int32), np.zeros(B, np.int32)
    for b, f in enumerate(frames):
        k = len(f["kps"])
        n[b] = k
        kps[b, :k], desc[b, :k], rp[b, :k], taken[b, :k], cs[b] = f["kps"], f["desc"], f["right_points"], f["taken"], f["cell_start"]
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(a.shape[0], -1)).to(dev)  # noqa: E731
    return dict(n=torch.from_numpy(n).to(dev), kps=t(kps).view(B, cap, 24), desc=torch.from_numpy(desc.view(np.int64)).to(dev),
                right_points=torch.from_numpy(rp).to(dev), taken=torch.from_numpy(taken).to(dev), cell_start=torch.from_numpy(cs).to(dev))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    orc.build()
    rng = np.random.default_rng(a.seed)
    dev = torch.device("cuda", 0)
    pm = SnakeORBMatcher(0)
    t0, n_b, n_f = time.time(), 0, 0
    while time.time() - t0 < a.seconds:
        B = int(rng.choice([1, 2, 3, 8, 17, 40]))
        big = rng.random() < 0.3
        n_levels = int(rng.integers(2, 8))
        cases = [T.make_tracking_case(orc, rng, n_clutter=int(rng.choice([0, 5, 200, 900, 2600 if big else 700])),
                                      m_pts=int(rng.choice([20, 64, 65, 400, 1500 if big else 300])), n_levels=n_levels) for _ in range(B)]
        frames = [c[0] for c in cases]
        cam, ls = cases[0][1], cases[0][3]
        cap = max(len(f["kps"]) for f in frames) + int(rng.integers(1, 9))
        coarse = [T.lm_coarse(orc, c[4]) for c in cases]
        fine = [T.lm_fine(orc, rng, c[4], c[2], ls) for c in cases]
        mc_cap, mf_cap = max(len(c) for c in coarse) + 3, max(len(f) for f in fine) + 5
        pc, pf = np.zeros((B, mc_cap), LM_COARSE_DTYPE), np.zeros((B, mf_cap), LM_FINE_DTYPE)
        nc, nf = np.zeros(B, np.int32), np.zeros(B, np.int32)
        for b in range(B):
            nc[b], nf[b] = len(coarse[b]), len(fine[b])
            pc[b, : nc[b]], pf[b, : nf[b]] = coarse[b], fine[b]
        D = pack(frames, cap)
        poses = torch.from_numpy(np.stack([c[2] for c in cases])).to(dev)
        d_pc = torch.from_numpy(pc.view(np.uint8).reshape(B, mc_cap, 88)).to(dev)
        d_pf = torch.from_numpy(pf.view(np.uint8).reshape(B, mf_cap, 96)).to(dev)
        d_nc, d_nf = torch.from_numpy(nc).to(dev), torch.from_numpy(nf).to(dev)
        mi_c = torch.full((B, mc_cap), -7, dtype=torch.int32, device=dev)
        mi_f = torch.full((B, mf_cap), -7, dtype=torch.int32, device=dev)
        vis = torch.full((B, mf_cap), 9, dtype=torch.uint8, device=dev)
        n_c, n_ff = torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
        fd = frames_dev(T.BOUNDS, D["n"], D["kps"], D["desc"], D["right_points"], D["taken"], D["cell_start"])
        thc, fe, direction = float(rng.uniform(3, 35)), int(rng.integers(30, 120)), int(rng.integers(0, 3))
        thf, ratio = float(rng.uniform(0.8, 7)), float(rng.choice([0.6, 0.8, 0.9]))
        torch.cuda.synchronize()
        pm.coarse_batch_dev(fd, cam, poses, d_pc, d_nc, thc, fe, direction, ls, mi_c, n_c)
        pm.mark_taken_batch_dev(mi_c, d_nc, D["taken"])
        ro = bool(rng.integers(0, 2))  # every other batch through the read-only form (snk_match_project_fine_batch_ro_dev)
        pf_in = d_pf.clone() if ro else None
        pm.fine_batch_dev(fd, cam, poses, d_pf, d_nf, thf, ratio, ls, mi_f, vis, n_ff, write_valid=not ro)
        pm.sync()
        if ro and not torch.equal(d_pf, pf_in):
            print("FAIL: the read-only fine matcher wrote into the local-map records")
            sys.exit(1)
        mi_c, mi_f, vis, n_c, n_ff = mi_c.cpu().numpy(), mi_f.cpu().numpy(), vis.cpu().numpy(), n_c.cpu().numpy(), n_ff.cpu().numpy()
        pf_after = d_pf.cpu().numpy().view(LM_FINE_DTYPE).reshape(B, mf_cap)
        for b, (frame, _, pose, _, _, _) in enumerate(cases):
            wn, widx = orc.match_coarse(frame, cam, pose, coarse[b], thc, fe, direction, ls)
            ok = n_c[b] == wn and np.array_equal(mi_c[b, : nc[b]], widx) and (mi_c[b, nc[b]:] == -1).all()
            f2 = dict(frame)
            f2["taken"] = frame["taken"].copy()
            f2["taken"][widx[widx >= 0]] = 1
            wn, widx, wvis, wvalid = orc.match_fine(f2, cam, pose, fine[b], thf, ratio, ls)
            ok = ok and n_ff[b] == wn and np.array_equal(mi_f[b, : nf[b]], widx) and np.array_equal(vis[b, : nf[b]], wvis) and \
                (ro or np.array_equal(pf_after[b, : nf[b]]["valid"], wvalid)) and np.array_equal(np.asarray(wvis) != 0, np.asarray(wvalid) != 0)
            if not ok:
                print(f"MISMATCH batch {n_b} frame {b}: B {B} features {len(frame['kps'])} cap {cap} coarse {nc[b]} fine {nf[b]} "
                      f"th {thc}/{thf} fe {fe} dir {direction} ratio {ratio} levels {n_levels}")
                return 1
        n_b += 1
        n_f += B
    pm.close()
    print(f"fuzz_track_batch: {n_b} batches, {n_f} frames, all bit-exact (seed {a.seed}, {time.time() - t0:.0f} s)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
