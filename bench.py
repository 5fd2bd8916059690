#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X.

Metric (BASELINE.json): frames/s of ORB extract + match at 752x480.  One "step" = one pass of
the per-frame feature pipeline over a batch of B synthetic EuRoC-shaped STEREO frames resident
in HBM: ORB extraction of the left and right image (2B images: pyramid, blur, FAST cells, quadtree
distribution, angle + BRIEF), undistort/rectify, feature grid, the Preprocess stereo row-band matcher,
and the brute-force kNN-2 Hamming matcher + ratio filter between the two descriptor sets.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0.  `value` is whole-job frames/s (all ranks; weak scaling: every
GPU processes its own batch, no data-path collective; one RCCL all_gather of a per-rank result
block after the timed region).  `roofline` is for the dominant kernel (the FAST cell kernel),
timed with HIP events recorded on the launch stream inside the timed region.  `cpu_baseline` is
the CPU oracle (a restatement of the reference path — NOT the reference binary, which cannot be
built here) timed on this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

W, H = 752, 480
ORB = dict(nfeatures=1000, scale_factor=1.2, n_levels=4, ini_th_fast=20, min_th_fast=7)  # reference configs/euroc.ini:32-36
BF_SYNTH = 47.9 * 2.5  # bf such that the synthetic disparities (2..60 px) fall inside [0, bf/2]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_CLOCK_HZ = 2.4e9  # MI355X_MICROARCH.md: max clock


def pyramid_pixels():
    s, tot = 1.0, 0
    f = np.float32(1.0)
    for _ in range(ORB["n_levels"]):
        inv = np.float32(1.0) / f
        tot += int(np.rint(np.float32(W) * inv)) * int(np.rint(np.float32(H) * inv))
        f = np.float32(f * np.float32(ORB["scale_factor"]))
    return tot


def cpu_baseline(frames, gpu=None, seconds_budget=12.0):
    """Oracle pipeline on host cores with the reference's thread configuration: extractor 2
    threads (fd_threads), matchers 4 threads (num_tracking_threads).

    The oracle's results are KEPT (first visit of every distinct frame) and, after the clock has stopped, compared bit for bit
    with what the timed steps left in HBM (`gpu`, host copies made by `snapshot_outputs`): keypoints / descriptors / counts of
    BOTH extractor output sets (the two-stream pipeline alternates between them; an ordering bug between the streams would show
    here), the grid-ordered left features, `right_points` / `depth` / match count of StereoMatching, the kNN-2 table and the
    filtered pairs.  The line reports `identical_to_gpu` and `frames_checked`."""
    from oracle import oracle as orc

    orc.build()
    p = orc.orb_params(ORB["nfeatures"], ORB["scale_factor"], ORB["n_levels"], ORB["ini_th_fast"], ORB["min_th_fast"])
    ls = (np.float32(ORB["scale_factor"]) ** np.arange(ORB["n_levels"])).astype(np.float32)
    rect = orc.rectification((1.0, 1.0, 0.0, 0.0))
    bounds = (0.0, 0.0, float(W), float(H))
    kept = {}
    done, t0 = 0, time.perf_counter()
    while True:
        i = done % len(frames)
        left, right = frames[i]
        kl, dl = orc.orb_detect(p, left, threads=2)
        kr, dr = orc.orb_detect(p, right, threads=2)
        rl, _ = orc.rectify(rect, kl)
        rr, _ = orc.rectify(rect, kr)
        perm, _, _, _ = orc.feature_grid(rl, bounds)          # computeFeatureGrid: left features into grid order
        g, gd = np.zeros_like(rl), np.zeros_like(dl)
        g[perm], gd[perm] = rl, dl
        ns, rp, dp = orc.stereo_match(g, gd, rr, dr, BF_SYNTH, ls, True)
        knn = orc.bf_knn2(gd, dr, threads=4)
        pairs = orc.bf_filter(knn, 60, 0.8)
        if i not in kept:
            kept[i] = (kl, dl, kr, dr, g, gd, rr, ns, rp, dp, knn, pairs)
        done += 1
        el = time.perf_counter() - t0
        if el >= seconds_budget or done >= 2000:
            break
    out = {"value": round(done / el, 3), "unit": "frames/s", "cores": 4, "kind": "port",
           "sample": f"{done} stereo frames {W}x{H} (extract L+R with 2 threads, rectify, feature grid, stereo match, "
                     f"BF kNN-2 with 4 threads + filter) in {el:.1f} s; CPU restatement of the reference path, not the reference binary"}
    if gpu is not None:
        bad = compare_with_oracle(gpu, kept)
        out["identical_to_gpu"] = not bad
        out["frames_checked"] = len(kept)
        out["checked"] = ("keypoints, descriptors and counts of both extractor output sets (L+R), grid-ordered left features, "
                          "right_points, depth, stereo count, kNN-2 table, filtered pairs -- bit for bit against the oracle")
        if bad:
            out["first_differences"] = bad[:8]
    return out


def snapshot_outputs(B, out_sets, kp64, kp64_g, desc_g, right_points, depth, n_stereo, knn, pairs, n_pairs, n_check):
    """Host copies of what the timed steps left in HBM for the first n_check frames of the batch (frame b holds distinct pair
    b): both extractor output sets and the single-buffered outputs of the post-extraction stage."""
    from snake_slam_amd.matcher import KNN2_DTYPE, KP64_DTYPE
    from snake_slam_amd.orb import KEYPOINT_DTYPE

    n = n_check
    cap = kp64_g.shape[1]
    sets = []
    for kps, desc, nkp in out_sets:
        sets.append({
            "kl": kps[:n].cpu().numpy().view(KEYPOINT_DTYPE).reshape(n, cap), "kr": kps[B:B + n].cpu().numpy().view(KEYPOINT_DTYPE).reshape(n, cap),
            "dl": desc[:n].cpu().numpy().view(np.uint64), "dr": desc[B:B + n].cpu().numpy().view(np.uint64),
            "nl": nkp[:n].cpu().numpy(), "nr": nkp[B:B + n].cpu().numpy()})
    return {"sets": sets, "rr": kp64[B:B + n].cpu().numpy().view(KP64_DTYPE).reshape(n, cap),
            "g": kp64_g[:n].cpu().numpy().view(KP64_DTYPE).reshape(n, cap), "gd": desc_g[:n].cpu().numpy().view(np.uint64),
            "rp": right_points[:n].cpu().numpy(), "dp": depth[:n].cpu().numpy(), "ns": n_stereo[:n].cpu().numpy(),
            "knn": knn[:n].cpu().numpy().view(KNN2_DTYPE).reshape(n, cap), "pairs": pairs[:n].cpu().numpy(), "np": n_pairs[:n].cpu().numpy()}


def compare_with_oracle(gpu, kept):
    """-> list of "frame b: what" strings (empty = identical).  Exact comparison of every field (floats bit for bit)."""
    bad = []

    def same(a, b):
        a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()

    for b, (kl, dl, kr, dr, g, gd, rr, ns, rp, dp, knn, pairs) in sorted(kept.items()):
        if b >= len(gpu["ns"]):
            continue
        nl, nr = len(kl), len(kr)
        for si, st in enumerate(gpu["sets"]):
            if int(st["nl"][b]) != nl or int(st["nr"][b]) != nr:
                bad.append(f"frame {b} set {si}: keypoint counts {int(st['nl'][b])}/{int(st['nr'][b])} vs oracle {nl}/{nr}")
                continue
            for name, got, want in (("left keypoints", st["kl"][b, :nl], kl), ("right keypoints", st["kr"][b, :nr], kr),
                                    ("left descriptors", st["dl"][b, :nl], dl), ("right descriptors", st["dr"][b, :nr], dr)):
                if not same(got, want):
                    bad.append(f"frame {b} set {si}: {name}")
        if len(bad) > 16:
            break
        checks = (("rectified right keypoints", gpu["rr"][b, :nr], rr), ("grid-ordered left keypoints", gpu["g"][b, :nl], g),
                  ("grid-ordered left descriptors", gpu["gd"][b, :nl], gd), ("right_points", gpu["rp"][b, :nl], rp),
                  ("depth", gpu["dp"][b, :nl], dp), ("kNN-2 table", gpu["knn"][b, :nl], knn),
                  ("filtered pairs", gpu["pairs"][b, :int(gpu["np"][b])], pairs))
        for name, got, want in checks:
            if not same(got, want):
                bad.append(f"frame {b}: {name}")
        if int(gpu["ns"][b]) != int(ns):
            bad.append(f"frame {b}: stereo match count {int(gpu['ns'][b])} vs {int(ns)}")
    return bad


def cpu_baseline_ba(ba_check=None, seconds_budget=8.0):
    """Oracle LBA solve (1 thread, like the reference's numThreads = 1, LocalBundleAdjustment.cpp:56).  `ba_check` holds, for
    three windows of the timed batch, (scene, cost_initial, cost_final, poses, points) as the LAST timed solve left them: the
    oracle solves the same scenes and the line reports the largest relative cost difference and pose / point RMSE
    (north_star: <= 1e-5 RMSE)."""
    from oracle import oracle as orc
    from snake_slam_amd import synth

    sc, _ = synth.ba_scene()
    done, t0 = 0, time.perf_counter()
    while True:
        orc.ba_solve(sc, orc.ba_options())
        done += 1
        el = time.perf_counter() - t0
        if el > seconds_budget:
            break
    out = {"value": round(done * 3 / el, 2), "unit": "LM iterations/s", "cores": 1, "kind": "port",
           "sample": f"{done} solves of the 20x2000x8 window (3 LM iterations each) in {el:.1f} s"}
    if ba_check:
        worst = {"cost_initial_rel": 0.0, "cost_final_rel": 0.0, "pose_rmse": 0.0, "point_rmse": 0.0}
        for k, (scene, ci, cf, pose, pt) in ba_check.items():
            wpose, wpt, wci, wcf, _ = orc.ba_solve(scene, orc.ba_options())
            worst["cost_initial_rel"] = max(worst["cost_initial_rel"], abs(ci - wci) / max(abs(wci), 1e-300))
            worst["cost_final_rel"] = max(worst["cost_final_rel"], abs(cf - wcf) / max(abs(wcf), 1e-300))
            worst["pose_rmse"] = max(worst["pose_rmse"], float(np.sqrt(np.mean((pose - wpose) ** 2))))
            worst["point_rmse"] = max(worst["point_rmse"], float(np.sqrt(np.mean((pt - wpt) ** 2))))
        ok = worst["cost_initial_rel"] <= 1e-9 and worst["cost_final_rel"] <= 1e-7 and worst["pose_rmse"] <= 1e-5 and worst["point_rmse"] <= 1e-5
        out["identical_to_gpu"] = bool(ok)
        out["windows_checked"] = sorted(ba_check)
        out["tolerance"] = "pose / point RMSE <= 1e-5 (north_star), cost_initial 1e-9 / cost_final 1e-7 relative"
        out["worst"] = {k: float(f"{v:.3e}") for k, v in worst.items()}
    return out


TRACK_M_COARSE = 1500   # maxFeatures: last-frame + last-keyframe points (reference SnakeGlobal.h:120, TrackingCoarse.cpp:94-127)
TRACK_M_FINE = 10000    # reserved size of the fine local map (reference Map/LocalMap.h:88)
TRACK_CAM = (458.654, 457.296, 367.215, 248.375, BF_SYNTH)


def tracking_points(kps, desc, n, depth, rng, m_coarse, m_fine, level_scale):
    """Synthetic local map of ONE frame built from the frame's own features (host numpy, untimed): every point is a feature
    back-projected to its stereo depth (or a seeded depth in [2, 12] m) with 1.5 px of reprojection noise, its descriptor the
    feature's with ~24 of 256 bits flipped.  The local map has more points than the frame has features (1 500 / 10 000 vs
    ~1 000), so features are drawn with repetition: several points compete for one feature, as in a real local map.
    Camera pose = identity.  Returns (snk_lm_coarse[m_coarse], snk_lm_fine[m_fine])."""
    from snake_slam_amd.tracking import LM_COARSE_DTYPE, LM_FINE_DTYPE

    fx, fy, cx, cy, _ = TRACK_CAM

    def make(m, dtype):
        out = np.zeros(m, dtype)
        if n == 0:
            return out, np.zeros(m, np.int64), np.zeros(m)
        j = rng.integers(0, n, m)
        z = np.where(depth[j] > 0, np.clip(depth[j].astype(np.float64), 0.5, 60.0), rng.uniform(2.0, 12.0, m))
        u = kps["x"][j] + rng.normal(0, 1.5, m)
        v = kps["y"][j] + rng.normal(0, 1.5, m)
        pos = np.stack([(u - cx) / fx * z, (v - cy) / fy * z, z], 1)
        out["pos"] = pos
        nrm = -pos / np.linalg.norm(pos, axis=1, keepdims=True)
        out["normal"] = nrm
        flip = rng.integers(0, 2**64, (m, 4), dtype=np.uint64) & rng.integers(0, 2**64, (m, 4), dtype=np.uint64) \
            & rng.integers(0, 2**64, (m, 4), dtype=np.uint64) & (rng.integers(0, 2**64, (m, 4), dtype=np.uint64) | rng.integers(0, 2**64, (m, 4), dtype=np.uint64))
        out["desc"] = desc[j] ^ flip
        return out, j, np.linalg.norm(pos, axis=1)

    if n == 0:
        return make(m_coarse, LM_COARSE_DTYPE)[0], make(m_fine, LM_FINE_DTYPE)[0]
    pc, j, _ = make(m_coarse, LM_COARSE_DTYPE)
    pc["octave"] = kps["octave"][j]
    pc["angle"] = np.mod(kps["angle"][j] + rng.normal(0, 6.0, m_coarse), 360.0).astype(np.float32)
    pf, j, dist = make(m_fine, LM_FINE_DTYPE)
    pf["reference_depth"] = dist.astype(np.float32)
    pf["reference_scale_level"] = kps["octave"][j]
    pf["valid"] = 1
    return pc, pf


def sequence_mode(args, rank, world, local, dev):
    """BASELINE.json config 5 ("8 sequences one-per-GPU, RCCL/xGMI gather"): rank r tracks sequence r, one frame per step,
    through the host (PCIe-inclusive, synchronous) entry points in the order the reference's threads call them; after the
    timed region ONE all_gather carries every rank's padded TUM trajectory block (SURVEY.md section 8e,
    reference Snake/System/System.cpp:552-563).  No collective on the data path."""
    import torch

    from snake_slam_amd import parallel, synth
    from snake_slam_amd.sequence import SequenceTracker, trajectory_block, trajectory_rows

    n_frames = args.warmup + args.steps
    if not args.host_api:
        return lockstep_sequence_mode(args, rank, world, local, dev, n_frames)
    frames = list(synth.sequence_frames(rank, n_frames, W, H))
    trk = SequenceTracker(TRACK_CAM, orb=ORB, device=local, width=W, height=H)
    for t in range(args.warmup):
        trk.process(*frames[t], float(t))
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.warmup, n_frames):
        trk.process(*frames[t], float(t))
    torch.cuda.synchronize()
    parallel.barrier()
    t1 = time.perf_counter()
    elapsed = parallel.max_over_ranks(t1 - t0, dev)
    block = torch.from_numpy(trajectory_block(trk.rows, n_frames)).to(dev)
    blocks = parallel.gather_blocks(block)  # one RCCL all_gather of (1 + 8 * frames) doubles per rank
    if rank == 0:
        baseline_m = TRACK_CAM[4] / TRACK_CAM[0]
        per_rank = []
        for r, b in enumerate(blocks):
            rows = trajectory_rows(b.cpu().numpy())
            gt = 0.05 * baseline_m * rows[-1, 0]
            per_rank.append({"rank": r, "frames": int(len(rows)), "final_position": [round(float(v), 5) for v in rows[-1, 1:4]],
                             "ground_truth_x": round(float(gt), 5)})
        st = trk.stats
        out = {"metric": f"frames/s, sequence mode (per-frame tracking chain through the host API) @{W}x{H}",
               "value": round(world * args.steps / elapsed, 2), "unit": "frames/s", "n_gpus": world, "dist": parallel.describe(), "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "u8", "data": f"synthetic: one seeded {W}x{H} stereo sequence per rank (rig moving 0.05 baselines per frame)",
               "config": {"workload": f"{world} independent stereo sequences, one per GPU: Detect L+R, rectify, feature grid, StereoMatching, "
                                      "matchKnn2 + filterMatches vs the previous frame, RefinePoseWithMatches; host API (PCIe inclusive)",
                          "parallelism": f"{world} x one sequence per GPU, one all_gather of the TUM trajectory blocks"},
               "keypoints_per_image": round(st["keypoints"] / max(1, 2 * st["frames"]), 1),
               "stereo_matches_per_frame": round(st["stereo"] / max(1, st["frames"]), 1),
               "bf_pairs_per_frame": round(st["bf_pairs"] / max(1, st["frames"] - 1), 1),
               "pose_inliers_per_frame": round(st["inliers"] / max(1, st["frames"] - 1), 1),
               "trajectory_block_bytes": int(block.numel() * 8), "trajectories": per_rank,
               "roofline": None,  # a latency path (one frame at a time); the roofline object belongs to the batch mode
               "cpu_baseline": None}
        print(json.dumps(out), flush=True)
    trk.close()


def lockstep_sequence_mode(args, rank, world, local, dev, n_frames):
    """Config 5 the MI355X way: S sequences per GPU in lockstep, device resident (snake_slam_amd.sequence.MultiSequenceTracker) --
    frame t of every sequence is one batch through the batched entry points, images are the only per-step PCIe traffic (double
    buffered beside the kernels), poses stay on the device until the end.  One step = frame t of all S sequences of a rank;
    value = sequences x steps / time over all ranks; one all_gather of the ranks' S trajectory blocks."""
    import torch

    from snake_slam_amd import parallel, synth
    from snake_slam_amd.sequence import MultiSequenceTracker, trajectory_block, trajectory_rows

    S = args.seqs_per_gpu
    seqs = args.sequences  # generated in main() before the HIP context existed (forked workers)
    trk = MultiSequenceTracker(TRACK_CAM, S, n_frames, orb=ORB, device=local, width=W, height=H)

    # every step's 2 S images staged in pinned host memory beforehand (where a camera driver / reader thread would put them):
    # the timed step is then one asynchronous DMA + the device-resident chain, not a Python memcpy loop
    staged = [trk.stage([seqs[s][t][0] for s in range(S)], [seqs[s][t][1] for s in range(S)]) for t in range(n_frames)]

    def step(t):
        trk.process_staged(staged[t], float(t))

    for t in range(args.warmup):
        step(t)
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.warmup, n_frames):
        step(t)
    torch.cuda.synchronize()
    parallel.barrier()
    t1 = time.perf_counter()
    elapsed = parallel.max_over_ranks(t1 - t0, dev)
    rows, st = trk.results()
    block = torch.from_numpy(np.concatenate([trajectory_block(r, n_frames) for r in rows])).to(dev)
    blocks = parallel.gather_blocks(block)  # one all_gather: S x (1 + 8 * frames) doubles per rank
    if rank == 0:
        baseline_m = TRACK_CAM[4] / TRACK_CAM[0]
        per = 1 + 8 * n_frames
        per_rank = []
        for r, b in enumerate(blocks):
            bb = b.cpu().numpy()
            finals = [trajectory_rows(bb[s * per:(s + 1) * per])[-1] for s in range(S)]
            per_rank.append({"rank": r, "sequences": S, "frames_per_sequence": n_frames,
                             "mean_final_x": round(float(np.mean([f[1] for f in finals])), 5),
                             "ground_truth_x": round(float(0.05 * baseline_m * finals[0][0]), 5)})
        out = {"metric": f"frames/s, sequence mode, {S} sequences per GPU in lockstep (device-resident tracking chain) @{W}x{H}",
               "value": round(world * S * args.steps / elapsed, 2), "unit": "frames/s", "n_gpus": world, "dist": parallel.describe(),
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u8",
               "data": f"synthetic: {S} seeded {W}x{H} stereo sequences per rank (rig moving 0.05 baselines per frame), images uploaded per step",
               "config": {"workload": f"{world} x {S} independent stereo sequences in lockstep: Detect L+R, rectify, feature grid, StereoMatching, "
                                      "matchKnn2 + filterMatches vs the previous frame, RefinePoseWithMatches, all device resident; images over PCIe "
                                      "(double buffered)",
                          "sequences_per_gpu": S,
                          "parallelism": f"{world} GPUs x {S} sequences per GPU, one all_gather of the TUM trajectory blocks"},
               "keypoints_per_image": round(st["keypoints"] / max(1, 2 * st["frames"]), 1),
               "stereo_matches_per_frame": round(st["stereo"] / max(1, st["frames"]), 1),
               "bf_pairs_per_frame": round(st["bf_pairs"] / max(1, st["frames"] - S), 1),
               "pose_inliers_per_frame": round(st["inliers"] / max(1, st["frames"] - S), 1),
               "image_bytes_per_step": int(2 * S * W * H), "trajectory_block_bytes": int(block.numel() * 8), "trajectories": per_rank,
               "roofline": None, "cpu_baseline": None}
        print(json.dumps(out), flush=True)
    trk.close()


def harris_leg(args, step, stream, out_sets, step_no, n_sets, B):
    """The headline batch once more with the Harris response as the corner measure (`"orb.response"` = 1: harris_kernel + the ranked
    distribution): frames/s of the same step.  Returns the line's "harris" object and host copies of the
    first frames' keypoints / descriptors of the last step (for cpu_baseline_harris).  Runs last: it overwrites the extractor's output sets."""
    import torch

    from snake_slam_amd import _lib as L_
    from snake_slam_amd.orb import KEYPOINT_DTYPE

    L_.set_definition("orb.response", 1)
    try:
        with torch.cuda.stream(stream):
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.harris_steps):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
        kps, desc, nkp = out_sets[(step_no[0] - 1) % n_sets]
        out = {"metric": "stereo frames/s, the headline step with the Harris response as corner measure (orb.response = 1)",
               "value": round(B * args.harris_steps / (t1 - t0), 1), "unit": "frames/s", "ms_per_step": round((t1 - t0) / args.harris_steps * 1e3, 4),
               "steps": args.harris_steps}
        hn, snap = nkp.cpu().numpy(), []
        for b in range(min(4, B)):
            for side in (0, 1):
                row = b + side * B
                n = int(hn[row])
                snap.append((b, side, kps[row, :n].cpu().numpy().view(KEYPOINT_DTYPE).reshape(n), desc[row, :n].cpu().numpy().view(np.uint64)))
        return out, snap
    finally:
        L_.set_definition("orb.response", 0)


def cpu_baseline_harris(frames, n_dpairs, snap):
    """The oracle under the same definition on the images `harris_leg` kept: bit for bit."""
    from oracle import oracle as orc

    orc.build()
    orc.set_definition("orb.response", 1)
    try:
        p = orc.orb_params(ORB["nfeatures"], ORB["scale_factor"], ORB["n_levels"], ORB["ini_th_fast"], ORB["min_th_fast"])
        same = True
        for b, side, gk, gd in snap:
            wk, wd = orc.orb_detect(p, frames[b % n_dpairs][side], threads=2)
            same = same and len(gk) == len(wk) and all(np.array_equal(gk[f], wk[f]) for f in ("x", "y", "angle", "response", "octave")) and np.array_equal(gd, wd)
        return {"identical_to_oracle": bool(same), "images_checked": len(snap)}
    finally:
        orc.set_definition("orb.response", 0)


def kitti_leg(args):
    """BASELINE.json configs[2] (KITTI seq 00 stereo 1241x376, 2000 features, 7 levels; reference configs/kitti.ini:30-34) as part of the
    default line: this script once more with --workload kitti (front-end only, 512 stereo frames per step, outputs of the last step
    checked bit for bit against the oracle on a few frames), its line cut down to the measured keys.  Runs after the EuRoC legs have
    released the GPU's memory; a failure is reported in the line, it does not take the headline down."""
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "kitti", "--steps", str(args.kitti_steps), "--warmup", "2", "--batch", "512",
           "--ba-windows", "0", "--gba-keyframes", "0", "--pose-frames", "0", "--track-frames", "0", "--frame-calls", "0", "--kitti-steps", "0", "--harris-steps", "0",
           "--cpu-seconds", "3", "--check-frames", "16", "--distinct", "64"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
        if r.returncode != 0 or not line:
            return {"error": f"kitti leg failed (status {r.returncode}): {r.stderr[-300:]}"}
        d = json.loads(line[-1])
        keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "data", "config", "roofline", "pipeline_roofline", "stage_ms_per_step",
                "keypoints_per_image", "stereo_matches_per_frame")
        k = {key: d[key] for key in keep if key in d}
        cb = d.get("cpu_baseline", {})
        k["checked_against_oracle"] = {key: cb[key] for key in ("identical_to_gpu", "frames_checked") if key in cb}
        k["cpu_baseline"] = {key: cb[key] for key in ("value", "unit", "cores", "kind", "sample") if key in cb}
        return k
    except Exception as e:  # noqa: BLE001
        return {"error": f"kitti leg: {e!r}"}


def spawn_ranks_if_needed(args):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset) starts its own N ranks: it re-executes this very command
    line under `torch.distributed.run --nproc-per-node N` on 127.0.0.1 (one process per GPU, rank r on GPU r) and exits with the
    launcher's status; rank 0 of the children prints the one JSON line.  Under a launcher (the driver's torch.distributed.run)
    WORLD_SIZE must equal --gpus.  Fails loudly when the box has fewer than N GPUs -- N ranks on fewer devices would be a
    different experiment (SNK_BENCH_DEVICE, the one-GPU rehearsal of the tests, is the explicit exception)."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world_env} ranks")
        return
    if args.gpus == 1:
        return
    import socket
    import subprocess

    import torch

    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and not os.environ.get("SNK_BENCH_DEVICE"):
        raise SystemExit(f"bench.py: --gpus {args.gpus} asked for but this box has {have} GPU(s) "
                         "(torch.cuda.device_count()); refusing to run N ranks on fewer devices")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def measured_copy_bandwidth(dev, gib=1.0, reps=5):
    """the achievable copy bandwidth of THIS device beside the 8 TB/s data-sheet peak -- a device-to-device copy of
    `gib` GiB (torch's copy kernel), read + write bytes over the best of `reps` timings with HIP events."""
    import torch

    n = int(gib * (1 << 30))
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    a.zero_(), b.zero_()
    best = 0.0
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        e1.synchronize()
        best = max(best, 2.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    return {"value": round(best, 1), "unit": "GB/s", "how": f"device-to-device copy of {gib:g} GiB (read + write bytes), best of {reps}, HIP events, this run"}


def _pipeline_traffic(section, sources):
    """profiles/pipeline_traffic.json[section] (tools/collect_pipeline_traffic.py) when it was collected on the current sources."""
    import hashlib

    tj = ROOT / "profiles" / "pipeline_traffic.json"
    try:
        t = json.loads(tj.read_text())[section]
        h = hashlib.sha256()
        for n in sources:
            h.update((ROOT / "snake_slam_amd" / "csrc" / n).read_bytes())
        return t if t.get("source_sha256") == h.hexdigest() else None
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="stereo frames per GPU per step (1024 = 2048 images = 0.74 GB of input resident in "
                    "HBM; every kernel of the chain ends in a tail of a few microseconds, measured 154 / 172 / 181 / 185 k frames/s at 128 / 256 / "
                    "512 / 1024 frames per step)")
    ap.add_argument("--ba-windows", type=int, default=1024, help="independent local-BA windows per GPU per step (0 = skip); measured "
                    "417.8 / 455.1 k LM iterations/s at 256 / 1024 windows per launch sequence (r03e): the per-iteration launch tails "
                    "are shared by more windows")
    ap.add_argument("--workload", choices=["euroc", "kitti"], default="euroc",
                    help="euroc = BASELINE.json's metric config (752x480, 1000 features, 4 levels); kitti = configs[2] "
                         "(1241x376, 2000 features, 7 levels), an extra measured case")
    ap.add_argument("--orb-chains", type=int, default=1, help="launch chains per ORB batch (1 = the library's default: the stages of a step do not overlap, "
                    "their HIP-event times add up to the step; 2 = two half batches on two streams, +1 to +3 %% from overlapped launch tails, per-kernel "
                    "timings then overlap)")
    ap.add_argument("--orb-stagger", type=int, default=0, help="staggered schedule of the extractor: the batch in this many parts, front "
                    "halves (level passes, FAST) back to back, the back half (distribution, descriptors) of part p on a second stream "
                    "beside the front half of part p + 1 (0 = off)")
    ap.add_argument("--gba-keyframes", type=int, default=300, help="keyframes of the global-BA leg (0 = skip; single GPU only)")
    ap.add_argument("--frame-calls", type=int, default=100, help="calls of the per-frame leg (snk_frontend_process against the six host "
                    "calls, one 752x480 stereo frame per call; 0 = skip; single GPU, euroc workload only)")
    ap.add_argument("--pose-frames", type=int, default=256, help="frames per pose-refinement call (0 = skip; single GPU only)")
    ap.add_argument("--track-frames", type=int, default=1024, help="frames of the tracking-matcher leg (device-resident coarse + fine "
                    "projection matchers on the frames the front-end left in HBM; 0 = skip)")
    ap.add_argument("--mode", choices=["batch", "sequence"], default="batch",
                    help="batch = the headline throughput benchmark (default); sequence = BASELINE.json config 5: every rank walks its "
                         "own synthetic stereo sequence frame by frame through the host entry points (one step = one frame per rank) "
                         "and the ranks' TUM trajectories are gathered with one all_gather")
    ap.add_argument("--seqs-per-gpu", type=int, default=1, help="--mode sequence: sequences per GPU (1 = BASELINE.json config 5 as written: "
                    "one sequence per GPU); S sequences run in lockstep, frame t of every sequence is one batch of the device-resident chain")
    ap.add_argument("--host-api", action="store_true", help="--mode sequence: ONE sequence per GPU through the host entry points, one "
                    "synchronous call per seam in the order the reference's threads make them (the round-2 form of the mode, ~800 frames/s); "
                    "default: the device-resident chain (snake_slam_amd.sequence.MultiSequenceTracker)")
    ap.add_argument("--distinct", type=int, default=256, help="distinct synthetic stereo pairs / BA scenes per rank, tiled over the batch / the "
                    "windows (0 = every frame of the batch and every BA window is its own seeded scene; 256 keeps the input generation of the "
                    "default batch at ~10 s)")
    ap.add_argument("--scene", choices=["textured", "flat"], default="textured",
                    help="textured (default since round 3) = a textured far wall + a few textured objects: ~40 %% of the left keypoints find "
                         "a stereo match, ~29 %% a BF pair; flat = the round-1/2 images (400 flat rectangles in draw order: ~12 %% / 6 %%)")
    ap.add_argument("--no-overlap", action="store_true", help="post-extraction stage on the extractor's stream (no overlap of batch i's "
                    "stage with batch i + 1's extraction)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump-track-inputs", default="", help="tools: write the first 16 frames of the tracking leg's inputs (grid-ordered features, "
                    "cell starts, local maps) to this .npz (tools/track_scan_stats.py reads it)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget of the CPU baseline of the front-end")
    ap.add_argument("--check-frames", type=int, default=256, help="frames of the last timed step whose outputs in HBM are compared "
                    "bit for bit with the oracle's (those the CPU baseline reaches in its time budget)")
    ap.add_argument("--no-stage-events", action="store_true", help="do not record per-stage HIP events")
    ap.add_argument("--harris-steps", type=int, default=20, help="steps of the Harris leg (the same batch under snk_set_definition(\"orb.response\", 1): "
                    "north_star lists the Harris score among the hot-path kernels; 0 = skip; single GPU only)")
    ap.add_argument("--kitti-steps", type=int, default=20, help="steps of the KITTI leg of the default line (BASELINE.json configs[2]: 1241x376 stereo, 2000 "
                    "features, 7 levels; this script re-run with --workload kitti on 512 frames per step once the EuRoC legs are done, its line "
                    "embedded under \"kitti\"; 0 = skip; single GPU, euroc workload only)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    spawn_ranks_if_needed(args)
    global W, H, ORB
    if args.workload == "kitti":  # reference configs/kitti.ini:30-34
        W, H = 1241, 376
        ORB = dict(nfeatures=2000, scale_factor=1.2, n_levels=7, ini_th_fast=20, min_th_fast=7)

    from snake_slam_amd import parallel, synth

    env_rank, _, local = parallel.env_rank_world()
    if os.environ.get("SNK_BENCH_DEVICE"):  # rehearsal of the N > 1 code on a one-GPU box (with SNK_DIST_BACKEND=gloo): every rank on this device
        local = int(os.environ["SNK_BENCH_DEVICE"])
    # Synthetic inputs first (worker processes are forked here, before this process owns a HIP context): EVERY frame of the batch
    # and EVERY BA window is its own seeded scene, so the data-dependent kernels (FAST survivors, quadtree, stereo bands, the
    # camera-set grouping of BA) see B / NW different workloads, not a handful tiled.
    n_dpairs = min(args.batch, args.distinct) if args.distinct > 0 else args.batch
    n_dscenes = min(args.ba_windows, args.distinct) if args.distinct > 0 else args.ba_windows
    frames, ba_distinct = [], []
    if args.mode == "batch":
        frames = synth.stereo_frames([env_rank * args.batch + i for i in range(n_dpairs)], W, H,
                                     texture=0.0 if args.scene == "flat" else None)
        ba_distinct = synth.ba_scenes([synth.SEED + 1000 * env_rank + k for k in range(n_dscenes)])
    elif not args.host_api:  # device-resident sequence mode: S sequences of this rank, generated by forked workers before HIP is touched
        args.sequences = synth.sequences([env_rank * args.seqs_per_gpu + s for s in range(args.seqs_per_gpu)], args.warmup + args.steps, W, H)

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rank, world = parallel.init_distributed(dev)  # "nccl" = RCCL over xGMI; used for barrier + result gather only

    if args.mode == "sequence":
        sequence_mode(args, rank, world, local, dev)
        parallel.shutdown()
        return

    from snake_slam_amd.matcher import BruteForceMatcher, Preprocess, Rectification
    from snake_slam_amd.orb import ORBExtractor
    from snake_slam_amd.tracking import FeatureGrid

    B = args.batch
    # ---- synthetic frames (seeded, generated above), resident in HBM ----
    pitch = (W + 63) & ~63
    host = np.zeros((2 * B, H, pitch), np.uint8)  # [0,B): left images, [B,2B): right images
    for b in range(B):
        l, r = frames[b % n_dpairs]
        host[b, :, :W] = l
        host[B + b, :, :W] = r
    images = torch.from_numpy(host).to(dev)
    del host

    stream = torch.cuda.Stream(device=dev)
    sh = stream.cuda_stream
    ext = ORBExtractor(**ORB, device=local, stream=sh)
    cap = ext.configure(W, H, 2 * B)
    if args.orb_chains != 1:
        ext.set_chains(args.orb_chains)
    if args.orb_stagger:
        ext.set_stagger(args.orb_stagger)
    # Two streams, two sets of extractor outputs: the post-extraction stage of batch i (rectify, grid, stereo, kNN-2 + filter --
    # small kernels that leave most of the chip idle) runs on stream B beside the extraction of batch i + 1 on stream A.  Batches
    # are independent frames; all K steps are complete inside the timed region (synchronize + barrier).  --no-overlap: one stream.
    overlap = not args.no_overlap
    stream_b = torch.cuda.Stream(device=dev) if overlap else stream
    shb = stream_b.cuda_stream
    pre = Preprocess(local, shb)
    bf = BruteForceMatcher(local, shb)
    grid = FeatureGrid(local, shb)
    GRID_BOUNDS = (0.0, 0.0, float(W), float(H))  # featureGridBounds of the undistorted image
    n_cells = int(np.ceil(W / 20.0)) * int(np.ceil(H / 20.0))
    rect = Rectification.make((1.0, 1.0, 0.0, 0.0))  # synthetic pairs are already rectified
    level_scale = (np.float32(ORB["scale_factor"]) ** np.arange(ORB["n_levels"])).astype(np.float32)

    n_sets = 2 if overlap else 1
    out_sets = [(torch.zeros((2 * B, cap, 24), dtype=torch.uint8, device=dev), torch.zeros((2 * B, cap, 4), dtype=torch.int64, device=dev),
                 torch.zeros(2 * B, dtype=torch.int32, device=dev)) for _ in range(n_sets)]
    ev_a = [torch.cuda.Event() for _ in range(n_sets)]  # extraction of the set done (stream A)
    ev_b = [torch.cuda.Event() for _ in range(n_sets)]  # post-extraction stage has finished reading the set (stream B)
    kps, desc, nkp = out_sets[0]
    kp64 = torch.zeros((2 * B, cap, 24), dtype=torch.uint8, device=dev)
    kp64_g = torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev)  # left keypoints / descriptors in feature-grid order
    desc_g = torch.zeros((B, cap, 4), dtype=torch.int64, device=dev)
    perm = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    cell_start = torch.zeros((B, n_cells + 1), dtype=torch.int32, device=dev)
    right_points = torch.full((B, cap), -1000.0, dtype=torch.float32, device=dev)
    depth = torch.full((B, cap), -1000.0, dtype=torch.float32, device=dev)
    n_stereo = torch.zeros(B, dtype=torch.int32, device=dev)
    knn = torch.zeros((B, cap, 4), dtype=torch.int32, device=dev)
    pairs = torch.zeros((B, cap, 2), dtype=torch.int32, device=dev)
    n_pairs = torch.zeros(B, dtype=torch.int32, device=dev)

    step_no = [0]

    def step():
        k = step_no[0] % n_sets
        step_no[0] += 1
        kps, desc, nkp = out_sets[k]
        if overlap:
            stream.wait_event(ev_b[k])  # the stage that read this set two steps ago is done (no-op for a fresh event)
        ext.detect_batch_dev(images, kps, desc, nkp)
        if overlap:
            ev_a[k].record(stream)
            stream_b.wait_event(ev_a[k])
        pre.rectify_batch_dev(rect, kps, nkp, kp64)                                   # undistortKeypoints / rect.Forward
        grid.create_batch_dev(GRID_BOUNDS, kp64[:B], desc[:B], nkp[:B], kp64_g, desc_g, perm, cell_start)  # computeFeatureGrid
        pre.match_batch_dev(kp64_g, desc_g, nkp[:B], kp64[B:], desc[B:], nkp[B:], BF_SYNTH, level_scale, True,
                            right_points, depth, n_stereo)                            # StereoMatching
        bf.knn2_batch_dev(desc_g, nkp[:B], desc[B:], nkp[B:], knn)                    # matchKnn2 + filterMatches
        bf.filter_batch_dev(knn, nkp[:B], 60, 0.8, pairs, n_pairs)
        if overlap:
            ev_b[k].record(stream_b)

    barrier = parallel.barrier

    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            torch.cuda.synchronize()
            right_points.fill_(-1000.0)
            depth.fill_(-1000.0)
            torch.cuda.synchronize()
            step()
        torch.cuda.synchronize()
        ext.set_profiling(not args.no_stage_events)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
    kps, desc, nkp = out_sets[(step_no[0] - 1) % n_sets]  # what the last step left in HBM (the legs below read it)
    stage_ms, n_calls = ext.stage_times() if not args.no_stage_events else ([0, 0, 0, 0, 0], 0)
    ext.set_profiling(False)

    elapsed = parallel.max_over_ranks(t1 - t0, dev)
    verify = not args.no_cpu_baseline and world == 1
    gpu_snapshot = None
    if verify:  # what the LAST timed steps left in HBM (both extractor sets), before any other leg runs
        gpu_snapshot = snapshot_outputs(B, out_sets, kp64, kp64_g, desc_g, right_points, depth, n_stereo, knn, pairs, n_pairs,
                                        min(B, n_dpairs, args.check_frames))

    # ---- second half of the metric: local-BA LM iterations/s (20 KF x 2000 pts x 8 obs/pt) ----
    ba_out = None
    if args.ba_windows > 0:
        from snake_slam_amd.ba import BARec, lba_options

        NW, LM_IT = args.ba_windows, 3
        distinct = ba_distinct
        ba = BARec(lba_options(), device=local, stream=sh)
        th0 = time.perf_counter()
        ba.create([distinct[k % n_dscenes] for k in range(NW)])
        ba.sync()
        th1 = time.perf_counter()  # the batch's scene hand-over (host lists on ONE core + upload), outside the timed region
        with torch.cuda.stream(stream):
            for _ in range(max(1, args.warmup)):
                ba.reset()
                ba.solve_async(LM_IT)
            torch.cuda.synchronize()
            barrier()
            tb0 = time.perf_counter()
            for _ in range(args.steps):
                ba.reset()
                ba.solve_async(LM_IT)
            torch.cuda.synchronize()
            barrier()
            tb1 = time.perf_counter()
        # The same windows WITH their hand-over on the clock (round-5 review: `value` is a resident-data number; the reference pays
        # `create(scene)` in every solve, LocalBundleAdjustment.cpp:357-365, and so does the CPU baseline beside it): snk_ba_set_problems
        # (host list building on up to 16 threads + upload of the lists) followed by the three LM iterations, warm handle, median of 3.
        ho_ms, ho_total_ms = [], []
        with torch.cuda.stream(stream):
            for _ in range(3):
                torch.cuda.synchronize()
                tw0 = time.perf_counter()
                ba.create([distinct[k % n_dscenes] for k in range(NW)])
                ba.sync()
                tw1 = time.perf_counter()
                ba.solve_async(LM_IT)
                torch.cuda.synchronize()
                tw2 = time.perf_counter()
                # without what the Python binding spends turning numpy arrays into snk_ba_problem structs (a C++ host holds them already)
                ho_ms.append((tw1 - tw0) * 1e3 - ba.last_pack_ms)
                ho_total_ms.append((tw2 - tw0) * 1e3 - ba.last_pack_ms)
        ho_warm = sorted(ho_ms)[1]
        ho_total = parallel.max_over_ranks(sorted(ho_total_ms)[1] * 1e-3, dev)
        ci, cf = ba.solve(0)
        ba_check = None
        if verify:  # state of three windows after the LAST timed solve (reset + 3 LM iterations), compared with the oracle below
            ba_check = {k: (distinct[k % n_dscenes], float(ci[k]), float(cf[k])) + ba.state(k)[:2] for k in sorted({0, NW // 2 - 1 if NW > 1 else 0, NW - 1})}
        # single-window latency (one problem per launch sequence)
        ba1 = BARec(lba_options(), device=local, stream=sh)
        ba1.create(distinct[0])
        with torch.cuda.stream(stream):
            ba1.solve_async(LM_IT)
            torch.cuda.synchronize()
            tl0 = time.perf_counter()
            for _ in range(10):
                ba1.reset()
                ba1.solve_async(LM_IT)
            torch.cuda.synchronize()
            tl1 = time.perf_counter()
        tba = torch.tensor([parallel.max_over_ranks(tb1 - tb0, dev)], dtype=torch.float64)
        ba_out = {"metric": "local-BA LM iterations/s (20 KF x 2000 pts x 8 obs/pt, 3 LM its, PCG<=30)",
                  "value": round(world * NW * LM_IT * args.steps / float(tba.item()), 1), "unit": "LM iterations/s",
                  "windows_per_gpu_per_step": NW, "ms_per_step": round(float(tba.item()) / args.steps * 1e3, 4),
                  "single_window_ms_per_solve": round((tl1 - tl0) / 10 * 1e3, 4),
                  "ms_batch_hand_over_first": round((th1 - th0) * 1e3, 1),  # not part of `value`: windows are resident when the timed region starts
                  "ms_batch_hand_over_warm": round(ho_warm, 1),
                  # hand-over (snk_ba_set_problems: host-built lists + upload, every step) + solve on the clock -- the figure to set beside
                  # cpu_baseline, whose oracle pays its set-up inside every solve too
                  "value_with_hand_over": round(world * NW * LM_IT / ho_total, 1),
                  "ms_per_step_with_hand_over": round(ho_total * 1e3, 2),
                  "cost_initial": round(float(ci[0]), 3), "cost_final": round(float(cf[0]), 3), "dtype": "f64",
                  "data": f"synthetic: {n_dscenes} distinct seeded scenes per rank over {NW} windows"}
        # ~5.65 MB algorithmic per LM iteration of the 20 x 2000 x 8 window
        ba_out["roofline"] = {"bound": "hbm", "algorithmic_bytes_per_lm_iteration": 5650000,
                              "achieved": round(5.65e6 * ba_out["value"] / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(5.65e6 * ba_out["value"] / 1e9 / HBM_PEAK_GBS, 5)}
        # What the kernels actually move and compute, beside the model fraction above (its 5.65 MB contain 4.6 MB of W written and read
        # back, which schur_fused keeps in LDS): HBM bytes per window and LM iteration from the FETCH_SIZE / WRITE_SIZE passes of
        # tools/profile_ba.sh (profiles/pipeline_traffic.json, bound to ba.hip by hash), and the f64 rate on SURVEY.md section 8d's
        # 46 MFLOP per LM iteration against the 78.6 TFLOP/s f64 peak (vector = matrix on this part).
        ba_out["roofline"]["traffic"] = None
        pt = _pipeline_traffic("ba", ["ba.hip"])
        if pt is not None:
            tb = pt["hbm_bytes_per_window_iteration"]
            ba_out["roofline"]["traffic"] = int(tb)
            ba_out["roofline"]["actual"] = {"hbm_bytes_per_lm_iteration": int(tb), "achieved": round(tb * ba_out["value"] / 1e9, 2), "unit": "GB/s",
                                            "frac": round(tb * ba_out["value"] / 1e9 / HBM_PEAK_GBS, 5),
                                            "source": "FETCH_SIZE x 2 + WRITE_SIZE of separate rocprofv3 --pmc passes (profiles/pipeline_traffic.json)"}
        ba_out["roofline"]["flops"] = {"model_flop_per_lm_iteration": 46.0e6, "achieved": round(46.0e6 * ba_out["value"] / 1e12, 3), "peak": 78.6,
                                       "unit": "TFLOP/s (f64)", "frac": round(46.0e6 * ba_out["value"] / 1e12 / 78.6, 5)}
        ba.close()
        ba1.close()
        # global BA (GlobalBundleAdjustment::FullBA(4), PCG <= 40): one big scene,
        # reduced system beyond one workgroup's LDS -> multi-workgroup PCG.  Rank 0 only, single GPU only.
        if world == 1 and args.gba_keyframes > 0:
            gsc = synth.ba_scene(n_kf=args.gba_keyframes, n_pt=50 * args.gba_keyframes, obs_per_pt=10, seed=31, n_fixed=1)[0]
            gba = BARec(lba_options(max_iterations=4, max_pcg_iterations=40), device=local, stream=sh)
            tc0 = time.perf_counter()
            gba.create(gsc)
            gba.sync()
            tc1 = time.perf_counter()
            gci, gcf = gba.initAndSolve()
            tc2 = time.perf_counter()
            gba.create(gsc)  # the hand-over again on the handle (device buffers and pinned lists keep their capacity)
            gba.sync()
            tc3 = time.perf_counter()
            gci, gcf = gba.initAndSolve()
            tg0 = time.perf_counter()
            for _ in range(3):
                gba.reset()
                gci, gcf = gba.initAndSolve()
            tg1 = time.perf_counter()
            g_pose, g_pt, g_pcg = gba.state(0)
            gba.close()
            ms_solve = (tg1 - tg0) / 3 * 1e3
            # what the algorithm has to move (SURVEY.md section 8d's per-observation / per-point figures; the reduced camera system S is dense,
            # n6 x n6 doubles, written once per LM iteration and read once per PCG iteration -- the reference's explicit Schur + PCG does the same)
            n6_ = 6 * (args.gba_keyframes - 1)
            n_obs_, n_pt_ = 500 * args.gba_keyframes, 50 * args.gba_keyframes
            g_bytes = 4 * (n_obs_ * 328 + n_pt_ * 144 + 8 * n6_ * n6_) + int(g_pcg) * 8 * n6_ * n6_
            g_roof = {"bound": "hbm", "algorithmic_bytes_per_solve": int(g_bytes), "pcg_iterations": int(g_pcg),
                      "achieved": round(g_bytes / (ms_solve * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(g_bytes / (ms_solve * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                      "note": "S (n6 x n6 doubles = %.1f MB) fits the 256 MB Infinity Cache: and, up to 2048 unknowns, the registers of the persistent PCG kernel, which reads it once per LM iteration (the figure above counts one read per PCG iteration, as the algorithm is written); the solve is bound by the grid barrier of a PCG iteration and the exchange of A p between XCDs, not by bandwidth" % (8 * n6_ * n6_ / 1e6)}
            gt_ = _pipeline_traffic("gba", ["ba.hip"]) if args.gba_keyframes == 300 else None
            if gt_ is not None:
                g_roof["traffic"] = int(gt_["hbm_bytes_per_solve"])
                g_roof["actual_GBs"] = round(gt_["hbm_bytes_per_solve"] / (ms_solve * 1e-3) / 1e9, 2)
            g_cpu = None
            if not args.no_cpu_baseline:
                from oracle import oracle as orc_

                tq0 = time.perf_counter()
                o_pose, o_pt, o_ci, o_cf, _ = orc_.ba_solve(gsc, orc_.ba_options(4, 40))
                tq1 = time.perf_counter()
                rm = lambda a, b: float(np.sqrt(((np.asarray(a) - np.asarray(b)) ** 2).sum(-1).mean()))  # noqa: E731
                g_cpu = {"value": round((tq1 - tq0) * 1e3, 1), "unit": "ms per FullBA(4)", "cores": 1, "kind": "port",
                         "sample": "the same scene, one solve", "pose_rmse_vs_gpu": rm(g_pose, o_pose), "point_rmse_vs_gpu": rm(g_pt, o_pt),
                         "identical_to_gpu": bool(rm(g_pose, o_pose) <= 1e-5 and rm(g_pt, o_pt) <= 1e-5), "tolerance": 1e-5}
            ba_out["global_ba"] = {"metric": "FullBA(4) wall time, host call to result", "keyframes": args.gba_keyframes,
                                   "points": 50 * args.gba_keyframes, "observations": 500 * args.gba_keyframes,
                                   "ms_per_solve": round(ms_solve, 3), "pcg": "one cooperative launch per LM iteration (pcgl_persist_reg up to 2048 unknowns: the rows of S in registers for the launch, one grid barrier per PCG iteration; pcgl_persist1 above)",
                                   "roofline": g_roof, "cpu_baseline": g_cpu,
                                   "ms_scene_hand_over": {"first": round((tc1 - tc0) * 1e3, 3), "same_handle_again": round((tc3 - tc2) * 1e3, 3)},
                                   "cost_initial": round(float(gci[0]), 3),
                                   "cost_final": round(float(gcf[0]), 3)}

    # ---- pose refinement after the matchers: 256 frames x 300 matches per call.
    # Host API (host pointers in, synchronous), so the figure includes staging and PCIe; single GPU only.
    pose_out = None
    if world == 1 and args.pose_frames > 0:
        from snake_slam_amd.tracking import PoseRefinement

        probs = [synth.pose_problem(7000 + k, 300, outlier_frac=0.2) for k in range(8)]
        batch = [dict(pose=probs[k % 8]["pose0"], wps=probs[k % 8]["wps"], obs=probs[k % 8]["obs"]) for k in range(args.pose_frames)]
        ref = PoseRefinement(device=local)
        res = ref.refine_batch(synth.POSE_CAM, batch)
        tp0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            res = ref.refine_batch(synth.POSE_CAM, batch)
        tp1 = time.perf_counter()
        ref.close()
        pose_out = {"metric": "pose refinements/s (300 matches, 20 % outliers, 4 x 10 GN iterations), host API incl. staging",
                    "value": round(args.pose_frames * reps / (tp1 - tp0), 1), "unit": "frames/s", "frames_per_call": args.pose_frames,
                    "ms_per_call": round((tp1 - tp0) / reps * 1e3, 3), "inliers_frame0": int(res[0][2]), "dtype": "f64"}
        if not args.no_cpu_baseline:
            from oracle import oracle as orc

            cam = orc.Camera(*synth.POSE_CAM)
            tc0, done = time.perf_counter(), 0
            while time.perf_counter() - tc0 < 2.0:
                pr = probs[done % 8]
                orc.pose_refine(pr["pose0"], cam, pr["wps"], pr["obs"])
                done += 1
            pose_out["cpu_baseline"] = {"value": round(done / (time.perf_counter() - tc0), 1), "unit": "frames/s", "cores": 1,
                                        "kind": "port", "sample": f"{done} refinements of the same problems"}

    # ---- the reference's per-frame call shape (FeatureDetector::Detect x 2 + Preprocess::Process, one stereo frame at a time): the
    # one-call form snk_frontend_process (one upload, one launch chain / hipGraph replay, one download, ONE synchronisation) beside the
    # same work as six synchronous host calls.  Host pointers in and out: PCIe inclusive.  Median over distinct frames; single GPU only.
    frame_out = None
    if world == 1 and args.workload == "euroc" and args.frame_calls > 0:
        import ctypes as C

        from snake_slam_amd import _lib as L_
        from snake_slam_amd.frontend import Frontend
        from snake_slam_amd.matcher import Preprocess, Rectification
        from snake_slam_amd.tracking import FeatureGrid

        fpairs = [synth.stereo_frame(900 + k, W, H) for k in range(8)]
        rect_ = Rectification.make((458.654, 457.296, 367.215, 248.375))
        orb_t = (ORB["nfeatures"], ORB["scale_factor"], ORB["n_levels"], ORB["ini_th_fast"], ORB["min_th_fast"])
        fe = Frontend(orb_t, rect_, rect_, (0.0, 0.0, float(W), float(H)), 47.9, device=local)
        ext1, pre1, grid1 = ORBExtractor(**ORB, device=local), Preprocess(local), FeatureGrid(local)
        fe.Process(*fpairs[0])
        lib_ = L_.load()

        # both sides at the same layer: the Python wrappers (each allocates / copies its result arrays); the bare C ABI call is timed
        # beside them, and without Python by tools/cpp/frontend_latency.cpp
        def one_call(l, r):
            return fe.Process(l, r)

        def one_call_abi(l, r):
            return lib_.snk_frontend_process(fe._h, l.ctypes.data, W, r.ctypes.data, W, W, H, C.byref(fe._frame))

        def six_calls(l, r):
            kl, dl = ext1.Detect(l)
            kr, dr = ext1.Detect(r)
            ul, _ = pre1.rectify(rect_, kl)
            ur, _ = pre1.rectify(rect_, kr)
            perm = np.asarray(grid1.create((0.0, 0.0, float(W), float(H)), ul)[0])
            g_, gd_ = np.zeros_like(ul), np.zeros_like(dl)
            g_[perm], gd_[perm] = ul, dl
            return pre1.StereoMatching(g_, gd_, ur, dr, 47.9, fe.level_scale, True)[0]

        med = {}
        for name, fn in (("one_call", one_call), ("one_call_abi", one_call_abi), ("six_calls", six_calls)):
            for k in range(6):
                fn(*fpairs[k % 8])
            ts = []
            for k in range(args.frame_calls):
                t0 = time.perf_counter()
                fn(*fpairs[k % 8])
                ts.append(time.perf_counter() - t0)
            med[name] = float(np.median(ts)) * 1e3
        n_st = int(fe._frame.n_stereo)
        # the PIPELINED form (snk_frontend_submit / snk_frontend_collect: the reference's FeatureDetection -> Preprocess stage queue,
        # Snake/Preprocess/FeatureDetector.h:39): still one frame per call, `depth` frames in flight; bare C ABI calls from one thread
        want = [fe.Process(*fpairs[k]) for k in range(8)]
        depth_ = 3
        fe.set_depth(depth_)
        identical = True
        for k in range(8 + depth_ - 1):  # checked pass through the wrappers (also takes every slot past its captured frame)
            if k < 8:
                fe.Submit(*fpairs[k])
            if k >= depth_ - 1:
                g_ = fe.Collect()
                w_ = want[k - depth_ + 1]
                identical = identical and all(np.array_equal(g_[key], w_[key]) for key in w_)
        for k in range(8):
            fe.Submit(*fpairs[k]), fe.Collect()
        n_pipe = max(args.frame_calls * 5, 200)

        def pipe_run(count):
            for k in range(count + depth_ - 1):
                if k < count:
                    l_, r_ = fpairs[k % 8]
                    lib_.snk_frontend_submit(fe._h, l_.ctypes.data, W, r_.ctypes.data, W, W, H)
                if k >= depth_ - 1:
                    lib_.snk_frontend_collect(fe._h, C.byref(fe._frame), -1)

        pipe_run(4 * depth_)
        pipe_fps = 0.0
        for _ in range(3):  # best of three: the loop is a Python loop around two C calls per frame and shares the host with whatever else runs
            tq0 = time.perf_counter()
            pipe_run(n_pipe)
            pipe_fps = max(pipe_fps, n_pipe / (time.perf_counter() - tq0))
        # round 6: the same loop with caller-owned page-locked images (snk_frontend_submit_pinned: no staging copy inside submit)
        pin_ = [fe.pinned_images(W, H, 2) for _ in range(8)]
        for k in range(8):
            pin_[k][0], pin_[k][1] = fpairs[k]
        pin_adr = [(int(q[0].ctypes.data), int(q[1].ctypes.data)) for q in pin_]

        def pipe_run_pinned(count):
            for k in range(count + depth_ - 1):
                if k < count:
                    a_l, a_r = pin_adr[k % 8]
                    lib_.snk_frontend_submit_pinned(fe._h, a_l, W, a_r, W, W, H)
                if k >= depth_ - 1:
                    lib_.snk_frontend_collect(fe._h, C.byref(fe._frame), -1)

        identical_pinned = True
        for k in range(8 + depth_ - 1):
            if k < 8:
                fe.SubmitPinned(pin_[k][0], pin_[k][1])
            if k >= depth_ - 1:
                g_ = fe.Collect()
                w_ = want[k - depth_ + 1]
                identical_pinned = identical_pinned and all(np.array_equal(g_[key], w_[key]) for key in w_)
        pipe_run_pinned(4 * depth_)
        pipe_fps_pinned = 0.0
        for _ in range(3):
            tq0 = time.perf_counter()
            pipe_run_pinned(n_pipe)
            pipe_fps_pinned = max(pipe_fps_pinned, n_pipe / (time.perf_counter() - tq0))
        for hnd in (fe, ext1, pre1, grid1):
            hnd.close()
        # the same measurement without Python in the timed region: tools/cpp/frontend_latency.cpp through the C++ adaptor (built here with g++,
        # a few seconds; skipped when no compiler is around)
        cpp_ = None
        try:
            import subprocess

            rr = subprocess.run([sys.executable, str(ROOT / "tools" / "frontend_latency_cpp.py"), "8", str(max(args.frame_calls, 50))],
                                capture_output=True, text=True, timeout=300)
            ln = [x for x in rr.stdout.splitlines() if x.startswith("{")]
            if rr.returncode == 0 and ln:
                cj = json.loads(ln[-1])
                cpp_ = {"tool": "tools/cpp/frontend_latency.cpp (snake_hip::Frontend, no Python in the timed region)",
                        "one_call_ms": cj["one_call_ms"], "six_calls_ms": cj["six_calls_ms"],
                        "pipelined_frames_per_s": {k: v for k, v in cj["pipelined"].items()},
                        "pipelined_pinned_frames_per_s": {k: v for k, v in cj.get("pipelined_pinned", {}).items()},
                        "identical_match_counts": cj["pipelined_identical_match_counts"]}
        except Exception as e:  # noqa: BLE001
            cpp_ = {"error": repr(e)[:200]}
        # ... and the Python / ctypes loop above once more in a process of its own (tools/frontend_pipelined_py.py): this process holds a
        # dozen other handles and torch's streams, a process has four hardware queues, and the front-end's three slots overlap less when
        # their streams share queues with that company -- the ctypes host itself costs under a microsecond per call
        py_ = None
        try:
            import subprocess

            rr = subprocess.run([sys.executable, str(ROOT / "tools" / "frontend_pipelined_py.py"), str(max(args.frame_calls * 10, 500))],
                                capture_output=True, text=True, timeout=300)
            ln = [x for x in rr.stdout.splitlines() if x.startswith("{")]
            if rr.returncode == 0 and ln:
                py_ = json.loads(ln[-1])
        except Exception as e:  # noqa: BLE001
            py_ = {"error": repr(e)[:200]}
        frame_out = {"metric": f"ms per {W}x{H} stereo frame through the host API, one frame per call (PCIe inclusive, median of {args.frame_calls} calls)",
                     "value": round(med["one_call"], 4), "unit": "ms", "higher_is_better": False,
                     "entry_point": "snk_frontend_process: Detect L + R, undistortKeypoints, computeFeatureGrid, StereoMatching in one call, one synchronisation",
                     "layer": "Python wrappers on both sides (Frontend.Process against ORBExtractor.Detect x 2, rectify x 2, FeatureGrid.create, StereoMatching)",
                     "six_host_calls_ms": round(med["six_calls"], 4), "frames_per_s_one_call": round(1e3 / med["one_call"], 1),
                     "one_call_c_abi_ms": round(med["one_call_abi"], 4),
                     "pipelined": {"entry_points": "snk_frontend_submit / snk_frontend_collect (bare C ABI calls, one thread, one frame per call)",
                                   "depth": depth_, "frames": n_pipe, "frames_per_s": round(pipe_fps, 1), "identical_to_process": bool(identical),
                                   "pinned": {"entry_points": "snk_frontend_submit_pinned / snk_frontend_collect (caller-owned page-locked images: no staging copy)",
                                              "frames_per_s": round(pipe_fps_pinned, 1), "identical_to_process": bool(identical_pinned)},
                                   "own_process": py_,
                                   "note": "frames_per_s above is measured inside this process beside a dozen other handles' streams (four hardware queues per "
                                           "process); own_process is the same ctypes loop alone; cpp_adaptor below is the host layer a Snake-SLAM build uses"},
                     "cpp_adaptor": cpp_,
                     "stereo_matches_last_frame": n_st}

    # ---- tracking matchers on the frames the front-end left in HBM: SearchByProjectionFrameFrame2 with
    # M = 1500 points, `mvpMapPoints[idx] = mp` on the device, SearchByProjection2 with M = 10 000 points -- the 1-2 coarse + 1
    # fine call the Tracking thread makes per frame (TrackingCoarse.cpp:234, TrackingFine.cpp:149), for a batch of frames, device
    # resident (kp64_g / desc_g / cell_start / right_points are consumed where the batched front-end wrote them).
    track_out = None
    if world == 1 and args.track_frames > 0 and args.workload == "euroc":
        from snake_slam_amd.tracking import KP64_DTYPE, PoseRefinement, SnakeORBMatcher, frames_dev, pose_observations

        TB = min(args.track_frames, B)
        h_kps = kp64_g[:TB].cpu().numpy().view(KP64_DTYPE).reshape(TB, cap)
        h_desc = desc_g[:TB].cpu().numpy().view(np.uint64)
        h_n = nkp[:TB].cpu().numpy()
        h_depth = depth[:TB].cpu().numpy()
        rng = np.random.default_rng(synth.SEED + 4711)
        lm = [tracking_points(h_kps[b], h_desc[b], int(h_n[b]), h_depth[b], rng, TRACK_M_COARSE, TRACK_M_FINE, level_scale)
              for b in range(TB)]
        if args.dump_track_inputs:
            nd = min(TB, 16)
            np.savez_compressed(args.dump_track_inputs, kps=h_kps[:nd], n=h_n[:nd], cell_start=cell_start[:nd].cpu().numpy(),
                                coarse=np.stack([x[0] for x in lm[:nd]]), fine=np.stack([x[1] for x in lm[:nd]]),
                                level_scale=np.asarray(level_scale), bounds=np.asarray(GRID_BOUNDS, np.float64), cam=np.asarray(TRACK_CAM))
        d_pc = torch.from_numpy(np.stack([x[0] for x in lm]).view(np.uint8).reshape(TB, TRACK_M_COARSE, 88)).to(dev)
        pf_host = np.stack([x[1] for x in lm])
        d_pf0 = torch.from_numpy(pf_host.view(np.uint8).reshape(TB, TRACK_M_FINE, 96)).to(dev)
        d_pf = d_pf0
        d_mc = torch.full((TB,), TRACK_M_COARSE, dtype=torch.int32, device=dev)
        d_mf = torch.full((TB,), TRACK_M_FINE, dtype=torch.int32, device=dev)
        ident = np.zeros((TB, 7))
        ident[:, 3] = 1.0
        d_pose = torch.from_numpy(ident).to(dev)
        taken = torch.zeros((TB, cap), dtype=torch.uint8, device=dev)
        mi_c = torch.zeros((TB, TRACK_M_COARSE), dtype=torch.int32, device=dev)
        mi_f = torch.zeros((TB, TRACK_M_FINE), dtype=torch.int32, device=dev)
        vis = torch.zeros((TB, TRACK_M_FINE), dtype=torch.uint8, device=dev)
        n_c = torch.zeros(TB, dtype=torch.int32, device=dev)
        n_f = torch.zeros(TB, dtype=torch.int32, device=dev)
        trk = SnakeORBMatcher(local, sh)
        refp = PoseRefinement(device=local, stream=sh)  # on the SAME stream as the matchers: one ordered chain
        fd = frames_dev(GRID_BOUNDS, nkp[:TB], kp64_g[:TB], desc_g[:TB], right_points[:TB], taken, cell_start[:TB])
        d_pose0 = d_pose.clone()
        outl_c = torch.zeros((TB, TRACK_M_COARSE), dtype=torch.uint8, device=dev)
        inl_c = torch.zeros(TB, dtype=torch.int32, device=dev)

        def track_step():
            trk.coarse_batch_dev(fd, TRACK_CAM, d_pose, d_pc, d_mc, 10.0, 75, 0, level_scale, mi_c, n_c)   # th 10: stereo, Tracking.h:184
            refp.refine_matches_batch_dev(fd, depth[:TB], TRACK_CAM, d_pc, mi_c, d_mc, level_scale, d_pose, outl_c, inl_c)  # TrackingCoarse.cpp:270
            trk.mark_taken_batch_dev(mi_c, d_mc, taken)
            trk.fine_batch_dev(fd, TRACK_CAM, d_pose, d_pf, d_mf, 4.0, 0.8, level_scale, mi_f, vis, n_f,    # th 4: stereo, Tracking.h:189
                               write_valid=False)  # local-map records read-only (lmp.valid after the search = vis): nothing to restore

        def restore_inputs():  # what a step consumes: the taken mask and the start poses (16 KB + 56 B per frame) -- input preparation
            taken.zero_()      # (until round 4 also the 983 MB of local-map records whose `valid` flags the fine matcher cleared in place)
            d_pose.copy_(d_pose0)

        ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
        with torch.cuda.stream(stream):
            for _ in range(max(1, args.warmup)):
                restore_inputs()
                track_step()
            torch.cuda.synchronize()
            tt0 = time.perf_counter()
            for k in range(args.steps):
                restore_inputs()
                ev0[k].record(stream)   # HIP events on the chain's stream: the chain alone, inputs resident
                track_step()
                ev1[k].record(stream)
            torch.cuda.synchronize()
            tt1 = time.perf_counter()
        chain_s = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) * 1e-3
        wall_s = tt1 - tt0
        # algorithmic bytes per frame (same accounting as the other matchers): the frame's features once
        # (24 + 32 + 4 + 1 B each) + every local-map point once (88 / 96 B) + 4 B (+1) of result per point
        n_feat = float(h_n.mean())
        a_track = n_feat * 61 + TRACK_M_COARSE * (88 + 4) + n_feat * 61 + TRACK_M_FINE * (96 + 5)
        fps = TB * args.steps / chain_s
        track_out = {"metric": "frames/s of the tracking chain (coarse M=1500 th=10 -> RefinePoseWithMatches -> fine M=10000 th=4), device resident",
                     "value": round(fps, 1), "unit": "frames/s", "frames_per_step": TB, "ms_per_step": round(chain_s / args.steps * 1e3, 4),
                     "timing": "HIP events around the chain on its stream, summed over the steps (inputs resident); "
                               "ms_per_step_with_input_restore is the host clock around the same steps including the reset of the taken "
                               "mask and of the start poses (the local-map records are read-only: snk_match_project_fine_batch_ro_dev)",
                     "ms_per_step_with_input_restore": round(wall_s / args.steps * 1e3, 4),
                     "coarse_matches_per_frame": round(float(n_c.float().mean().item()), 1),
                     "fine_matches_per_frame": round(float(n_f.float().mean().item()), 1),
                     "pose_inliers_per_frame": round(float(inl_c.float().mean().item()), 1), "dtype": "u8 descriptors, f64 geometry",
                     "roofline": {"bound": "hbm", "algorithmic_bytes_per_frame": int(a_track),
                                  "achieved": round(a_track * fps / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(a_track * fps / 1e9 / HBM_PEAK_GBS, 6)}}
        if not args.no_cpu_baseline:
            from oracle import oracle as orc

            orc.build()
            orc.set_match_threads(4)  # num_tracking_threads (reference Settings.h:88)
            h_rp = right_points[:TB].cpu().numpy()
            h_cs = cell_start[:TB].cpu().numpy()
            gcols, grows = int(np.ceil(W / 20.0)), int(np.ceil(H / 20.0))
            tc0, done, same = time.perf_counter(), 0, True
            h_mi_c, h_mi_f, h_pose = mi_c.cpu().numpy(), mi_f.cpu().numpy(), d_pose.cpu().numpy()
            ocam = orc.Camera(*TRACK_CAM)
            while time.perf_counter() - tc0 < 4.0 and done < TB:
                b = done
                nb = int(h_n[b])
                fr = dict(kps=h_kps[b, :nb], desc=h_desc[b, :nb], right_points=h_rp[b, :nb], taken=np.zeros(nb, np.uint8),
                          cell_start=h_cs[b], bounds=GRID_BOUNDS, cols=gcols, rows=grows)
                _, wi = orc.match_coarse(fr, TRACK_CAM, ident[0], lm[b][0], 10.0, 75, 0, level_scale)
                sel = np.nonzero(wi >= 0)[0]
                wpose = ident[0]
                if len(sel) >= 3:
                    wobs = pose_observations(fr["kps"][wi[sel]], h_depth[b][wi[sel]], level_scale)
                    wpose, _, _ = orc.pose_refine(ident[0], ocam, lm[b][0]["pos"][sel], wobs)
                fr["taken"][wi[sel]] = 1
                _, wf, _, _ = orc.match_fine(fr, TRACK_CAM, wpose, lm[b][1].copy(), 4.0, 0.8, level_scale)
                same = same and np.array_equal(wi, h_mi_c[b]) and np.array_equal(wf, h_mi_f[b]) and bool(np.allclose(wpose, h_pose[b], rtol=0, atol=1e-9))
                done += 1
            orc.set_match_threads(1)
            track_out["cpu_baseline"] = {"value": round(done / (time.perf_counter() - tc0), 2), "unit": "frames/s", "cores": 4, "kind": "port",
                                         "sample": f"{done} frames of the same inputs (coarse, pose refinement, fine), matchers with 4 OpenMP threads",
                                         "identical_to_gpu": bool(same)}
        trk.close()
        refp.close()

    # ---- result gather: one fixed-size block per rank (RCCL all_gather over xGMI) ----
    block = torch.tensor([float(B * args.steps), float(nkp.sum().item()), float(n_stereo.sum().item()),
                          float(n_pairs.sum().item()), t1 - t0, 0.0, 0.0, 0.0], dtype=torch.float64, device=dev)
    blocks = parallel.gather_result_blocks(block)

    if rank == 0:
        total_frames = sum(float(b[0].item()) for b in blocks)
        value = total_frames / elapsed
        P = pyramid_pixels()
        out = {
            "metric": f"frames/s ORB extract+match @{W}x{H}",
            "value": round(value, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "dist": parallel.describe(),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "timed_region_s": round(elapsed, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": (f"synthetic: seeded {W}x{H} stereo pairs (" + ("gradient + 400 flat rectangles + noise" if args.scene == "flat" else
                     "gradient + textured far wall + 20 textured objects + noise") + f"), {n_dpairs} distinct pairs per rank over a batch of {B}, resident in HBM"),
            "config": {"workload": ("EuRoC" if args.workload == "euroc" else "KITTI") + f" stereo {W}x{H}: ORB extract (L+R, {ORB['nfeatures']} feat, {ORB['n_levels']} levels) + stereo row-band match + BF kNN-2 Hamming match",
                       "scene": args.scene,
                       "frames_per_gpu_per_step": B, "images_per_frame": 2, "orb": ORB,
                       "parallelism": f"{world} x independent batches (one per GPU), RCCL all_gather of results only",
                       "streams": ("2: the post-extraction stage of batch i (rectify, grid, stereo, kNN-2) overlaps the extraction of batch i + 1"
                                   if overlap else "1")},
            "keypoints_per_image": round(float(blocks[0][1].item()) / (2 * B), 1),
            "stereo_matches_per_frame": round(float(blocks[0][2].item()) / B, 1),
            "bf_pairs_per_frame": round(float(blocks[0][3].item()) / B, 1),
            "stereo_yield": round(float(blocks[0][2].item()) / max(1.0, float(blocks[0][1].item()) / 2), 4),   # matches per left keypoint
            "bf_yield": round(float(blocks[0][3].item()) / max(1.0, float(blocks[0][1].item()) / 2), 4),
        }
        if n_calls > 0:
            # n_calls counts launch chains (1 per step unless --orb-chains > 1), each timed on its own stream
            fast_ms = stage_ms[2] / n_calls
            images_per_launch = 2 * B * args.steps // n_calls
            alg_bytes = P * images_per_launch  # read every pyramid pixel once (the FAST+score pass of A_orb)
            achieved = alg_bytes / (fast_ms * 1e-3) / 1e9
            traffic = None
            valu = None
            # the counters of THIS workload (a KITTI image has 1.55 x the pixels of a EuRoC one, 7 levels, other FAST cells: the EuRoC pass
            # scaled by the image count is not a KITTI measurement -- round-5 review; without its own pass the KITTI leg reports null)
            tj = ROOT / "profiles" / ("fast_kernel_traffic.json" if args.workload == "euroc" else f"fast_kernel_traffic_{args.workload}.json")
            if tj.exists():
                # PMC bytes of a separate rocprofv3 --pmc run (tools/profile_gpu.sh + tools/collect_traffic.py); only valid for the
                # kernel source it was collected on: a file whose recorded source hash is not the current one is refused
                try:
                    import hashlib

                    t = json.loads(tj.read_text())
                    cur = hashlib.sha256((ROOT / t.get("source", "snake_slam_amd/csrc/orb.hip")).read_bytes()).hexdigest()
                    if t.get("source_sha256") == cur and t.get("images_per_launch") and t.get("workload", "euroc") == args.workload:
                        scale = images_per_launch / t.get("images_per_launch")
                        traffic = int(t.get("hbm_bytes_per_launch") * scale)
                        if t.get("valu_insts_per_launch"):
                            # The kernel is instruction-bound, not HBM-bound (traffic ~ algorithmic bytes, nothing re-read): its
                            # vector-instruction roofline.  Peak: 256 CUs x 4 SIMDs, one wave64 integer / packed-16 VALU instruction per
                            # 4 cycles (SIMD16 issue; measured on level_kernel, profiles/NOTES.md) at the 2.4 GHz maximum clock.
                            n_valu = t["valu_insts_per_launch"] * scale
                            peak = 1024 * VALU_CLOCK_HZ / 4.0
                            valu = {"instr_per_launch": int(n_valu), "issue_peak": peak, "unit": "wave64 VALU instructions/s",
                                    "achieved": round(n_valu / (fast_ms * 1e-3), 1), "frac": round(n_valu / (fast_ms * 1e-3) / peak, 4),
                                    "instr_per_wave": round(t["valu_insts_per_launch"] / max(1, t.get("waves_per_launch") or 1), 1),
                                    "source": f"SQ_INSTS_VALU of a separate rocprofv3 --pmc pass (profiles/{tj.name})"}
                            if t.get("busy_cycles_per_launch"):  # the clock the chip actually held during the profiled launch
                                valu["frac_at_profiled_clock"] = round(t["valu_insts_per_launch"] * 4.0 / (1024 * t["busy_cycles_per_launch"]), 4)
                except Exception:
                    traffic, valu = None, None
            peak_meas = None
            if world == 1:
                try:
                    peak_meas = measured_copy_bandwidth(dev)
                except Exception as e:  # noqa: BLE001
                    peak_meas = {"error": repr(e)[:120]}
            out["roofline"] = {"bound": "hbm", "kernel": "fast_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "peak_measured": peak_meas,
                               "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                               "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(fast_ms, 4),
                               "images_per_launch": images_per_launch, "launches_per_step": n_calls // args.steps,
                               "limited_by": "valu" if valu else None, "valu": valu}
            # summed over the launch chains of a step (with --orb-chains 2 the chains overlap on the device and the sum exceeds the step time)
            out["stage_ms_per_step"] = {k: round(v / args.steps, 4) for k, v in zip(["pyramid", "blur", "fast", "distribute", "describe"], stage_ms)}
            out["stage_ms_per_step"]["launch_chains"] = n_calls // args.steps
            out["stage_ms_per_step"]["chains_overlap"] = bool(n_calls // args.steps > 1)
        # whole front-end against the HBM roofline: A_orb = 3 P + 56 N bytes per mono image,
        # BF / stereo matching (N1 + N2) * 32 + N1 * 16 bytes each
        n_kp = float(blocks[0][1].item()) / (2 * B)
        a_frame = 2 * (3 * P + 56 * n_kp) + 2 * (2 * n_kp * 32 + n_kp * 16)
        out["pipeline_roofline"] = {"bound": "hbm", "algorithmic_bytes_per_frame": int(a_frame),
                                    "achieved": round(a_frame * value / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(a_frame * value / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                                    "note": "extract (L+R) + stereo match + BF match, algorithmic bytes only"}
        pt = _pipeline_traffic("frontend", ["orb.hip", "matcher.hip", "track.hip", "preprocess.hip"]) if args.workload == "euroc" else None
        if pt is not None:
            # counter bytes of every kernel of a step (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes), per stereo frame
            out["pipeline_roofline"]["traffic"] = int(pt["hbm_bytes_per_frame"])
            out["pipeline_roofline"]["traffic_over_algorithmic"] = round(pt["hbm_bytes_per_frame"] / a_frame, 3)
            out["pipeline_roofline"]["actual_GBs"] = round(pt["hbm_bytes_per_frame"] * value / 1e9, 2)
        if ba_out is not None:
            out["ba"] = ba_out
        if pose_out is not None:
            out["pose_refine"] = pose_out
        if frame_out is not None:
            out["frontend_frame"] = frame_out
        if track_out is not None:
            out["tracking"] = track_out
        harris_snap = None
        if world == 1 and args.harris_steps > 0 and args.mode == "batch":
            out["harris"], harris_snap = harris_leg(args, step, stream, out_sets, step_no, n_sets, B)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames, gpu_snapshot, seconds_budget=args.cpu_seconds)
            if harris_snap:
                out["harris"].update(cpu_baseline_harris(frames, n_dpairs, harris_snap))
            if ba_out is not None:
                out["cpu_baseline"]["ba"] = cpu_baseline_ba(ba_check)
        if world == 1 and args.workload == "euroc" and args.mode == "batch" and args.kitti_steps > 0 and "WORLD_SIZE" not in os.environ:
            out["kitti"] = kitti_leg(args)
        print(json.dumps(out), flush=True)

    parallel.shutdown()


if __name__ == "__main__":
    main()
