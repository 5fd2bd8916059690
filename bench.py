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
N_DISTINCT = 8         # distinct synthetic stereo pairs, tiled to fill the batch


def pyramid_pixels():
    s, tot = 1.0, 0
    f = np.float32(1.0)
    for _ in range(ORB["n_levels"]):
        inv = np.float32(1.0) / f
        tot += int(np.rint(np.float32(W) * inv)) * int(np.rint(np.float32(H) * inv))
        f = np.float32(f * np.float32(ORB["scale_factor"]))
    return tot


def cpu_baseline(frames, seconds_budget=12.0):
    """Oracle pipeline on host cores with the reference's thread configuration: extractor 2
    threads (fd_threads), matchers 4 threads (num_tracking_threads)."""
    from oracle import oracle as orc

    orc.build()
    p = orc.orb_params(ORB["nfeatures"], ORB["scale_factor"], ORB["n_levels"], ORB["ini_th_fast"], ORB["min_th_fast"])
    ls = (np.float32(ORB["scale_factor"]) ** np.arange(ORB["n_levels"])).astype(np.float32)
    rect = orc.rectification((1.0, 1.0, 0.0, 0.0))
    done, t0 = 0, time.perf_counter()
    while True:
        left, right = frames[done % len(frames)]
        kl, dl = orc.orb_detect(p, left, threads=2)
        kr, dr = orc.orb_detect(p, right, threads=2)
        rl, _ = orc.rectify(rect, kl)
        rr, _ = orc.rectify(rect, kr)
        orc.stereo_match(rl, dl, rr, dr, BF_SYNTH, ls, True)
        knn = orc.bf_knn2(dl, dr, threads=4)
        orc.bf_filter(knn, 60, 0.8)
        done += 1
        el = time.perf_counter() - t0
        if el >= seconds_budget or done >= 2000:
            break
    return {"value": round(done / el, 3), "unit": "frames/s", "cores": 4, "kind": "port",
            "sample": f"{done} stereo frames {W}x{H} (extract L+R with 2 threads, rectify, stereo match, "
                      f"BF kNN-2 with 4 threads + filter) in {el:.1f} s; CPU restatement of the reference path, not the reference binary"}


def cpu_baseline_ba(seconds_budget=8.0):
    """Oracle LBA solve (1 thread, like the reference's numThreads = 1, LocalBundleAdjustment.cpp:56)."""
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
    return {"value": round(done * 3 / el, 2), "unit": "LM iterations/s", "cores": 1, "kind": "port",
            "sample": f"{done} solves of the 20x2000x8 window (3 LM iterations each) in {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="stereo frames per GPU per step")
    ap.add_argument("--ba-windows", type=int, default=256, help="independent local-BA windows per GPU per step (0 = skip)")
    ap.add_argument("--workload", choices=["euroc", "kitti"], default="euroc",
                    help="euroc = BASELINE.json's metric config (752x480, 1000 features, 4 levels); kitti = configs[2] "
                         "(1241x376, 2000 features, 7 levels), an extra measured case")
    ap.add_argument("--orb-chains", type=int, default=1, help="launch chains per ORB batch (2 = two half batches on two streams, +3 %%; "
                    "per-kernel timings then overlap)")
    ap.add_argument("--gba-keyframes", type=int, default=300, help="keyframes of the global-BA leg (0 = skip; single GPU only)")
    ap.add_argument("--pose-frames", type=int, default=256, help="frames per pose-refinement call (0 = skip; single GPU only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-events", action="store_true", help="do not record per-stage HIP events")
    args = ap.parse_args()
    global W, H, ORB
    if args.workload == "kitti":  # reference configs/kitti.ini:30-34
        W, H = 1241, 376
        ORB = dict(nfeatures=2000, scale_factor=1.2, n_levels=7, ini_th_fast=20, min_th_fast=7)

    import torch

    from snake_slam_amd import parallel

    _, _, local = parallel.env_rank_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rank, world = parallel.init_distributed(dev)  # "nccl" = RCCL over xGMI; used for barrier + result gather only

    from snake_slam_amd import synth
    from snake_slam_amd.matcher import BruteForceMatcher, Preprocess, Rectification
    from snake_slam_amd.orb import ORBExtractor
    from snake_slam_amd.tracking import FeatureGrid

    B = args.batch
    # ---- synthetic frames (seeded; a few distinct pairs tiled over the batch), resident in HBM ----
    frames = [synth.stereo_frame(rank * N_DISTINCT + i, W, H) for i in range(N_DISTINCT)]
    pitch = (W + 63) & ~63
    host = np.zeros((2 * B, H, pitch), np.uint8)  # [0,B): left images, [B,2B): right images
    for b in range(B):
        l, r = frames[b % N_DISTINCT]
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
    pre = Preprocess(local, sh)
    bf = BruteForceMatcher(local, sh)
    grid = FeatureGrid(local, sh)
    GRID_BOUNDS = (0.0, 0.0, float(W), float(H))  # featureGridBounds of the undistorted image
    n_cells = int(np.ceil(W / 20.0)) * int(np.ceil(H / 20.0))
    rect = Rectification.make((1.0, 1.0, 0.0, 0.0))  # synthetic pairs are already rectified
    level_scale = (np.float32(ORB["scale_factor"]) ** np.arange(ORB["n_levels"])).astype(np.float32)

    kps = torch.zeros((2 * B, cap, 24), dtype=torch.uint8, device=dev)
    desc = torch.zeros((2 * B, cap, 4), dtype=torch.int64, device=dev)
    nkp = torch.zeros(2 * B, dtype=torch.int32, device=dev)
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

    def step():
        ext.detect_batch_dev(images, kps, desc, nkp)
        pre.rectify_batch_dev(rect, kps, nkp, kp64)                                   # undistortKeypoints / rect.Forward
        grid.create_batch_dev(GRID_BOUNDS, kp64[:B], desc[:B], nkp[:B], kp64_g, desc_g, perm, cell_start)  # computeFeatureGrid
        pre.match_batch_dev(kp64_g, desc_g, nkp[:B], kp64[B:], desc[B:], nkp[B:], BF_SYNTH, level_scale, True,
                            right_points, depth, n_stereo)                            # StereoMatching
        bf.knn2_batch_dev(desc_g, nkp[:B], desc[B:], nkp[B:], knn)                    # matchKnn2 + filterMatches
        bf.filter_batch_dev(knn, nkp[:B], 60, 0.8, pairs, n_pairs)

    barrier = parallel.barrier

    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            right_points.fill_(-1000.0)
            depth.fill_(-1000.0)
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
    stage_ms, n_calls = ext.stage_times() if not args.no_stage_events else ([0, 0, 0, 0, 0], 0)
    ext.set_profiling(False)

    elapsed = parallel.max_over_ranks(t1 - t0, dev)

    # ---- second half of the metric: local-BA LM iterations/s (20 KF x 2000 pts x 8 obs/pt) ----
    ba_out = None
    if args.ba_windows > 0:
        from snake_slam_amd.ba import BARec, lba_options

        NW, LM_IT = args.ba_windows, 3
        distinct = [synth.ba_scene(seed=synth.SEED + 1000 * rank + k)[0] for k in range(4)]
        ba = BARec(lba_options(), device=local, stream=sh)
        ba.create([distinct[k % 4] for k in range(NW)])
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
        ci, cf = ba.solve(0)
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
                  "cost_initial": round(float(ci[0]), 3), "cost_final": round(float(cf[0]), 3), "dtype": "f64"}
        # ~5.65 MB algorithmic per LM iteration of the 20 x 2000 x 8 window
        ba_out["roofline"] = {"bound": "hbm", "algorithmic_bytes_per_lm_iteration": 5650000,
                              "achieved": round(5.65e6 * ba_out["value"] / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(5.65e6 * ba_out["value"] / 1e9 / HBM_PEAK_GBS, 5)}
        ba.close()
        ba1.close()
        # global BA (GlobalBundleAdjustment::FullBA(4), PCG <= 40): one big scene,
        # reduced system beyond one workgroup's LDS -> multi-workgroup PCG.  Rank 0 only, single GPU only.
        if world == 1 and args.gba_keyframes > 0:
            gsc = synth.ba_scene(n_kf=args.gba_keyframes, n_pt=50 * args.gba_keyframes, obs_per_pt=10, seed=31, n_fixed=1)[0]
            gba = BARec(lba_options(max_iterations=4, max_pcg_iterations=40), device=local, stream=sh)
            gba.create(gsc)
            gci, gcf = gba.initAndSolve()
            tg0 = time.perf_counter()
            for _ in range(3):
                gba.reset()
                gci, gcf = gba.initAndSolve()
            tg1 = time.perf_counter()
            gba.close()
            ba_out["global_ba"] = {"metric": "FullBA(4) wall time, host call to result", "keyframes": args.gba_keyframes,
                                   "points": 50 * args.gba_keyframes, "observations": 500 * args.gba_keyframes,
                                   "ms_per_solve": round((tg1 - tg0) / 3 * 1e3, 3), "cost_initial": round(float(gci[0]), 3),
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
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": f"synthetic: seeded {W}x{H} stereo pairs (gradient + 400 rectangles + noise), {N_DISTINCT} distinct pairs tiled over the batch, resident in HBM",
            "config": {"workload": ("EuRoC" if args.workload == "euroc" else "KITTI") + f" stereo {W}x{H}: ORB extract (L+R, {ORB['nfeatures']} feat, {ORB['n_levels']} levels) + stereo row-band match + BF kNN-2 Hamming match",
                       "frames_per_gpu_per_step": B, "images_per_frame": 2, "orb": ORB,
                       "parallelism": f"{world} x independent batches (one per GPU), RCCL all_gather of results only"},
            "keypoints_per_image": round(float(blocks[0][1].item()) / (2 * B), 1),
            "stereo_matches_per_frame": round(float(blocks[0][2].item()) / B, 1),
            "bf_pairs_per_frame": round(float(blocks[0][3].item()) / B, 1),
        }
        if n_calls > 0:
            # n_calls counts launch chains (1 per step unless --orb-chains > 1), each timed on its own stream
            fast_ms = stage_ms[2] / n_calls
            images_per_launch = 2 * B * args.steps // n_calls
            alg_bytes = P * images_per_launch  # read every pyramid pixel once (the FAST+score pass of A_orb)
            achieved = alg_bytes / (fast_ms * 1e-3) / 1e9
            traffic = None
            tj = ROOT / "profiles" / "fast_kernel_traffic.json"
            if tj.exists():
                try:
                    t = json.loads(tj.read_text())
                    if t.get("images_per_launch"):  # PMC bytes scale with the images of a launch
                        traffic = int(t.get("hbm_bytes_per_launch") * images_per_launch / t.get("images_per_launch"))
                except Exception:
                    traffic = None
            out["roofline"] = {"bound": "hbm", "kernel": "fast_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                               "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(fast_ms, 4),
                               "images_per_launch": images_per_launch, "launches_per_step": n_calls // args.steps}
            # summed over the launch chains of a step (with --orb-chains 2 the chains overlap and the sum exceeds the step time)
            out["stage_ms_per_step"] = {k: round(v / args.steps, 4) for k, v in zip(["pyramid", "blur", "fast", "distribute", "describe"], stage_ms)}
        # whole front-end against the HBM roofline: A_orb = 3 P + 56 N bytes per mono image,
        # BF / stereo matching (N1 + N2) * 32 + N1 * 16 bytes each
        n_kp = float(blocks[0][1].item()) / (2 * B)
        a_frame = 2 * (3 * P + 56 * n_kp) + 2 * (2 * n_kp * 32 + n_kp * 16)
        out["pipeline_roofline"] = {"bound": "hbm", "algorithmic_bytes_per_frame": int(a_frame),
                                    "achieved": round(a_frame * value / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(a_frame * value / 1e9 / HBM_PEAK_GBS, 5),
                                    "note": "extract (L+R) + stereo match + BF match, algorithmic bytes only"}
        if ba_out is not None:
            out["ba"] = ba_out
        if pose_out is not None:
            out["pose_refine"] = pose_out
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames)
            if ba_out is not None:
                out["cpu_baseline"]["ba"] = cpu_baseline_ba()
        print(json.dumps(out), flush=True)

    parallel.shutdown()


if __name__ == "__main__":
    main()
