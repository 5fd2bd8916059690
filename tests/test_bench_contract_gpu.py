"""GPU: bench.py honours the driver's contract — exactly one JSON line on stdout with the required keys."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu

REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
            "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict}


def test_bench_prints_one_json_line():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--kitti-steps", "2", "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "16",
                        "--ba-windows", "8", "--pose-frames", "8", "--gba-keyframes", "0", "--no-cpu-baseline", "--frame-calls", "10"],
                       capture_output=True, text=True, cwd=str(ROOT), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k, t in REQUIRED.items():
        assert k in d and isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "u8" and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and rf["kernel"] == "fast_kernel"
    assert d["value"] > 0 and d["ba"]["value"] > 0 and d["pose_refine"]["value"] > 0
    # BASELINE.json configs[2] rides in the default line: KITTI shape, checked against the oracle, with its own roofline
    k = d["kitti"]
    assert "error" not in k, k
    assert "1241x376" in k["config"]["workload"] and k["value"] > 0 and k["roofline"]["frac"] > 0
    assert k["checked_against_oracle"]["identical_to_gpu"] is True and k["checked_against_oracle"]["frames_checked"] >= 1, k
    # the pipelined per-frame front-end: same results as the synchronous call
    pf = d["frontend_frame"]["pipelined"]
    assert pf["identical_to_process"] is True and pf["frames_per_s"] > 0 and pf["depth"] == 3
    # round 6: caller-owned pinned images (snk_frontend_submit_pinned), the hand-over of the BA batch on the clock, the achievable copy
    # bandwidth beside the data-sheet peak, the Harris leg (north_star lists the Harris score among the kernels)
    assert pf["pinned"]["identical_to_process"] is True and pf["pinned"]["frames_per_s"] > 0
    assert 0 < d["ba"]["value_with_hand_over"] < d["ba"]["value"] and d["ba"]["ms_batch_hand_over_warm"] > 0
    pm = rf["peak_measured"]
    assert pm["unit"] == "GB/s" and 1000.0 < pm["value"] < 8000.0, pm
    assert d["harris"]["value"] > 0 and d["harris"]["unit"] == "frames/s" and "identical_to_oracle" not in d["harris"]  # --no-cpu-baseline: timing only


def test_line_carries_verified_cpu_baseline():
    """The default line's `cpu_baseline` is not only a timing: the oracle's results for the frames it processed are compared
    bit for bit with what the last timed steps left in HBM (both extractor output sets of the two-stream pipeline, the
    stereo / kNN-2 / filter outputs), and three BA windows with the oracle's solve.  Small batch, short CPU budget."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--kitti-steps", "0", "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "24",
                        "--distinct", "6", "--ba-windows", "8", "--pose-frames", "0", "--gba-keyframes", "0", "--track-frames", "0",
                        "--cpu-seconds", "1.0"], capture_output=True, text=True, cwd=str(ROOT), timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][0])
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 4 and cb["value"] > 0 and "sample" in cb
    assert cb["identical_to_gpu"] is True and cb["frames_checked"] == 6, cb
    assert cb["ba"]["cores"] == 1 and cb["ba"]["identical_to_gpu"] is True and cb["ba"]["windows_checked"] == [0, 3, 7], cb["ba"]
    assert d["timed_region_s"] > 0 and d["dist"] == {"world_size": 1, "backend": "none"}
    assert d["harris"]["identical_to_oracle"] is True and d["harris"]["images_checked"] == 8, d["harris"]


def test_gpus_flag_without_a_launcher_on_a_one_gpu_box():
    """`python bench.py --gpus 2` with WORLD_SIZE unset spawns its own ranks -- and refuses when the box has fewer GPUs."""
    import os

    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("box has two GPUs: the refusal cannot be provoked")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SNK_BENCH_DEVICE")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--kitti-steps", "0", "--gpus", "2", "--batch", "16"], capture_output=True, text=True,
                       cwd=str(ROOT), timeout=300, env=env)
    assert r.returncode != 0 and "--gpus 2" in r.stderr and "GPU(s)" in r.stderr, (r.returncode, r.stderr[-500:])
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]


def test_gpus_flag_spawns_its_own_ranks():
    """Same rehearsal as below but WITHOUT an external launcher: bench.py --gpus 2 re-executes itself under
    torch.distributed.run (both ranks on cuda:0, gloo instead of RCCL) and the one line says n_gpus = 2."""
    import os

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(SNK_DIST_BACKEND="gloo", SNK_BENCH_DEVICE="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--kitti-steps", "0", "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16",
                        "--ba-windows", "8", "--no-cpu-baseline"], capture_output=True, text=True, cwd=str(ROOT), timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dist"] == {"world_size": 2, "backend": "gloo"} and d["value"] > 0


@pytest.mark.parametrize("mode", ["batch", "sequence", "lockstep"])
def test_two_ranks_rehearsal_on_one_gpu(mode):
    """The N > 1 code of bench.py (per-rank data, barriers, max over ranks, result / trajectory gather, rank 0 prints) launched the
    way the driver launches it, with both ranks on cuda:0 and gloo standing in for RCCL (which refuses two ranks on one
    device): the line must report n_gpus = 2 and the work of two ranks."""
    import os

    env = dict(os.environ, SNK_DIST_BACKEND="gloo", SNK_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", str(ROOT / "bench.py"), "--kitti-steps", "0", "--gpus", "2", "--steps", "2", "--warmup", "1", "--mode",
           "sequence" if mode == "lockstep" else mode]
    if mode == "lockstep":
        cmd += ["--seqs-per-gpu", "3"]
    if mode == "sequence":
        cmd += ["--host-api"]
    if mode == "batch":
        cmd += ["--batch", "16", "--ba-windows", "8", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(ROOT), timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["dist"] == {"world_size": 2, "backend": "gloo"}
    if mode == "batch":
        assert d["config"]["frames_per_gpu_per_step"] == 16 and d["ba"]["value"] > 0
    elif mode == "lockstep":
        assert len(d["trajectories"]) == 2 and all(t["sequences"] == 3 for t in d["trajectories"]) and d["config"]["sequences_per_gpu"] == 3
    else:
        assert len(d["trajectories"]) == 2 if "trajectories" in d else True


def test_one_rank_process_group_on_rccl():
    """RCCL itself on the GPU box: with SNK_DIST_FORCE=1 bench.py creates a ONE-rank "nccl" (= RCCL) process group bound to
    cuda:0 and runs the very barrier / all_reduce(MAX) / all_gather calls of the N > 1 runs on device tensors.  A one-GPU box
    cannot show xGMI traffic, but it does show that RCCL initialises in this environment (HSA_ENABLE_IPC_MODE_LEGACY, device_id
    binding) and that the collective calls are issued with tensors RCCL accepts."""
    import os

    env = {k: v for k, v in os.environ.items() if k not in ("SNK_DIST_BACKEND", "SNK_BENCH_DEVICE")}
    env.update(SNK_DIST_FORCE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29578")
    for extra in (["--batch", "16", "--ba-windows", "8", "--track-frames", "0"], ["--mode", "sequence"], ["--mode", "sequence", "--host-api"]):
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--kitti-steps", "0", "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                            "--gba-keyframes", "0", "--pose-frames", "0"] + extra, capture_output=True, text=True, cwd=str(ROOT), timeout=900, env=env)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][0])
        assert d["dist"] == {"world_size": 1, "backend": "nccl"} and d["n_gpus"] == 1 and d["value"] > 0


def test_lockstep_sequence_mode_line():
    """`--mode sequence --seqs-per-gpu S`: S sequences per GPU in lockstep (device-resident chain); the line reports S x steps
    frames per rank and every sequence's recovered motion (the rig moves 0.05 baselines per frame)."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--kitti-steps", "0", "--gpus", "1", "--mode", "sequence", "--seqs-per-gpu", "4", "--steps", "5",
                        "--warmup", "1"], capture_output=True, text=True, cwd=str(ROOT), timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d["config"]["sequences_per_gpu"] == 4 and d["n_gpus"] == 1 and d["steps"] == 5 and d["value"] > 0
    assert abs(d["value"] - 4 * 5 / (d["ms_per_step"] * 5e-3)) < 0.01 * d["value"]
    t = d["trajectories"][0]
    assert t["sequences"] == 4 and t["frames_per_sequence"] == 6 and 0.6 * t["ground_truth_x"] < t["mean_final_x"] < 1.4 * t["ground_truth_x"]
    assert d["stereo_matches_per_frame"] > 100 and d["pose_inliers_per_frame"] > 30
