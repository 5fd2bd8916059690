#!/usr/bin/env python3
"""BASELINE.json config 5 on whatever node this runs on: every GPU one rank, both gather paths, with the checks a reviewer would make.

    python tools/run_all_gpus.py [--out gpurun_out/all_gpus] [--steps 50]

1. the NATIVE gather (include/snake_hip.h snk_dist_*, RCCL opened by the library, no torch / MPI in the process): tests/cpp/dist_driver.cpp
   built with g++ and started once per device (rank r on device r, rendezvous through a file), every rank gathering all ranks' TUM
   trajectory blocks (Snake/System/System.cpp:546-563) -- asserted: RCCL reported world = device count in every rank, every rank holds
   every rank's rows; the gathered blocks are written to <out>/native_rank<r>.txt;
2. the torch.distributed path: `bench.py --gpus N --mode sequence` (one stereo sequence per GPU, one all_gather of the trajectory
   blocks; bench.py starts its own N ranks on 127.0.0.1) -- asserted: n_gpus = N, dist.backend = nccl (= RCCL), N trajectories; the
   line goes to <out>/bench_sequence.json and the trajectories as TUM text to <out>/tum_rank<r>.txt;
3. `bench.py --gpus N` (the batch metric, weak scaling) -> <out>/bench_batch.json.
With one device everything runs with world size 1 (what the round-end GPU box has); the multi-rank paths are the same code."""
import argparse
import json
import os
import subprocess
import sys
import uuid
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "all_gpus"))
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--skip-batch", action="store_true")
    a = ap.parse_args()
    import torch

    n = torch.cuda.device_count()
    if n < 1:
        raise SystemExit("no GPU")
    n = min(n, 8)
    out = Path(a.out)
    out.mkdir(parents=True, exist_ok=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    summary = {"devices": n}

    # 1. native gather
    lib = ROOT / "snake_slam_amd" / "lib"
    exe = out / "dist_driver"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", f"-I{ROOT / 'include'}", f"-I{ROOT / 'snake_slam_amd' / 'cpp'}", str(ROOT / "tests" / "cpp" / "dist_driver.cpp"),
                    f"-L{lib}", "-lsnake_hip", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    rendezvous = out / f"rccl_id_{uuid.uuid4().hex}"
    procs = [subprocess.Popen([str(exe), str(rendezvous), str(r), str(n), str(r), str(out / f"native_rank{r}.txt")], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(n)]
    for r, p in enumerate(procs):
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, f"native gather, rank {r}: {err[-2000:]}"
    for r in range(n):
        lines = (out / f"native_rank{r}.txt").read_text().splitlines()
        assert lines[0].startswith("rccl ") and lines[0].endswith(f"world {n}"), lines[0]  # RCCL itself saw n ranks
        got = [ln for ln in lines if ln.startswith("rank ") and " identical " in ln]
        assert len(got) == n, (r, got)  # ... and this rank holds every rank's block, compared with what that rank sent
    assert not rendezvous.exists()
    summary["native_gather"] = {"world": n, "rccl": (out / "native_rank0.txt").read_text().splitlines()[0], "blocks_per_rank": n, "files": [f"native_rank{r}.txt" for r in range(n)]}

    # 2. torch.distributed, sequence mode
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(n), "--mode", "sequence", "--steps", str(a.steps), "--warmup", "3"], env=env,
                       capture_output=True, text=True, timeout=1800, cwd=str(ROOT))
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert r.returncode == 0 and lines, r.stderr[-3000:]
    seq = json.loads(lines[-1])
    (out / "bench_sequence.json").write_text(json.dumps(seq) + "\n")
    assert seq["n_gpus"] == n and len(seq["trajectories"]) >= n, (seq["n_gpus"], len(seq["trajectories"]))
    assert n == 1 or seq["dist"].get("backend") == "nccl", seq["dist"]
    summary["sequence_mode"] = {"value": seq["value"], "unit": seq["unit"], "n_gpus": seq["n_gpus"], "dist": seq["dist"], "trajectories": len(seq["trajectories"])}

    # 3. the batch metric on all GPUs (weak scaling)
    if not a.skip_batch:
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(n), "--steps", "20", "--warmup", "3"], env=env, capture_output=True, text=True,
                           timeout=3600, cwd=str(ROOT))
        lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
        assert r.returncode == 0 and lines, r.stderr[-3000:]
        bat = json.loads(lines[-1])
        (out / "bench_batch.json").write_text(json.dumps(bat) + "\n")
        assert bat["n_gpus"] == n
        summary["batch_mode"] = {"value": bat["value"], "unit": bat["unit"], "n_gpus": n, "ba": (bat.get("ba") or {}).get("value")}
    (out / "summary.json").write_text(json.dumps(summary, indent=1) + "\n")
    print(json.dumps(summary))
    return 0


if __name__ == "__main__":
    sys.exit(main())
