"""CPU: structural rules of the repository — the product path never touches the oracle, there is no CPU
fallback, bench.py refuses to run without a GPU, every C-ABI entry point cites a reference seam."""
import re
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_product_package_never_imports_the_oracle():
    offenders = []
    for f in (ROOT / "snake_slam_amd").rglob("*"):
        if f.suffix in (".py", ".hip", ".hpp", ".h", ".inc") and "build" not in f.parts:
            txt = f.read_text(errors="ignore")
            if re.search(r"import oracle|from oracle|from \.\.oracle|snk_oracle\.h|liborc|orc_[a-z_]+\(", txt):
                offenders.append(str(f.relative_to(ROOT)))
    assert offenders == [], offenders


def test_only_the_checker_legs_use_the_oracle():
    bench = (ROOT / "bench.py").read_text()
    # every use sits in a cpu_baseline leg (functions named cpu_baseline*, or the pose leg's baseline block)
    for m in re.finditer(r"from oracle import|import oracle", bench):
        head = bench[: m.start()]
        fn = re.findall(r"\ndef (\w+)\(", head)
        ctx = bench[max(0, m.start() - 400): m.start()]
        assert (fn and fn[-1].startswith("cpu_baseline")) or "no_cpu_baseline" in ctx, bench[m.start() - 80: m.start() + 40]
    entry = (ROOT / "__graft_entry__.py").read_text()
    body = entry[entry.index("def smoke"):]
    assert "oracle" in body  # smoke() checks against the oracle
    build = entry[entry.index("def build"): entry.index("def smoke")]
    assert "import oracle" not in build or "build(" in build  # build() may only compile it


def test_bench_refuses_to_run_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       cwd=str(ROOT), timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


def test_bench_gpus_flag_refuses_fewer_devices_than_ranks():
    """`python bench.py --gpus N` without a launcher spawns N ranks itself, and refuses -- before spawning anything -- when the
    box has fewer than N GPUs; under a launcher WORLD_SIZE must equal --gpus (the round-2 bench ignored the flag)."""
    import os

    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SNK_BENCH_DEVICE")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8"], capture_output=True, text=True, cwd=str(ROOT),
                       timeout=300, env=env)
    assert r.returncode != 0 and "--gpus 8" in r.stderr and "0 GPU(s)" in r.stderr, r.stderr[-600:]
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4"], capture_output=True, text=True, cwd=str(ROOT),
                       timeout=300, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr, r.stderr[-600:]


def test_handles_fail_loudly_without_a_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from snake_slam_amd import SnakeHipError
    from snake_slam_amd.matcher import BruteForceMatcher

    with pytest.raises(SnakeHipError):
        BruteForceMatcher()


def test_every_entry_point_cites_the_reference():
    h = (ROOT / "include" / "snake_hip.h").read_text()
    decls = list(re.finditer(r"SNK_API\s+\w[\w\s\*]*?\b(snk_\w+)\s*\(", h))
    assert len(decls) > 40
    cites = [m.start() for m in re.finditer(r"\.(?:cpp|h):\d+", h)]
    assert len(cites) > 40
    # plumbing entry points (status, handles, sync, debug, profiling) are exempt; every compute entry point
    # must have a reference citation (file:line) within the 3000 characters before its declaration
    exempt = re.compile(r"(create|destroy|sync|error|version|device_count|status|debug|profiling|stage_times|set_chains|set_stagger|"
                        r"max_keypoints|configure|reset|get_state|set_outliers|solve_async)$")
    missing = [m.group(1) for m in decls if not exempt.search(m.group(1)) and not any(m.start() - 3000 < c < m.start() for c in cites)]
    assert missing == [], missing
