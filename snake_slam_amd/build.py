"""In-tree build of the gfx950 C-ABI library (hipcc cross-compiles without a GPU).

`build_library()` compiles every ``csrc/*.hip`` to an object and links
``snake_slam_amd/lib/libsnake_hip.so``; objects are rebuilt only when a source or header is
newer.  No CUDA / multi-backend switches: the only target is gfx950.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
LIB = LIBDIR / "libsnake_hip.so"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: float results must be bit-identical between the kernels and the CPU oracle,
# so no silent FMA contraction anywhere (explicit fma() only where the algorithm says so).
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
    "-fno-fast-math", "-Wall", "-Wno-unused-function", f"-I{ROOT / 'include'}", f"-I{CSRC}",
]
# Per-source additions (appended, so they win).  ba.hip: bundle adjustment is specified by a tolerance (1e-5 RMSE, only
# summation orders ever differed from the oracle's) and its kernels are bound by fp64 issue slots -- a * b + c as one
# v_fma_f64 halves them.  Everything that must match the oracle bit for bit keeps -ffp-contract=off.
# pose.hip (round 4): pose refinement is specified by a tolerance too (<= 1e-9 on the pose against the oracle; the sums already differ
# from the oracle's by their order), its kernel is 62 % VALU-issue bound in fp64 multiply-add chains.  The flag covers the whole file: the
# glue kernels backproject_kernel (world points of the next frame) and gather_matches_kernel's weights are fp64 multiply-add chains too and
# were tolerance-specified like pose_kernel until round 6; backproject_kernel -- whose output is the next frame's local map -- now carries
# `#pragma clang fp contract(off)` and is bit-exact against the same expressions in double on the host (gather_matches_kernel's weight,
# sqrt(1 / (s * s)), has nothing to contract).
# pose.hip, -disable-machine-licm (round 6): pose_kernel's solver section (se3 exponential / logarithm, Cholesky) is full of 64-bit polynomial
# literals; the machine-level loop-invariant code motion hoists ~60 of their v_mov pairs in front of the iteration loops, the kernel
# sits at 254 registers and the allocator then SPILLS three of those constants to scratch (28-36 bytes per lane, reloaded in the loop
# where a v_mov would do).  Without the pass: 180 registers, no scratch (tests/test_kernel_resources.py holds that).
HIP_FLAGS_PER_SOURCE = {"ba.hip": ["-ffp-contract=fast"], "pose.hip": ["-mllvm", "-disable-machine-licm"]}  # pose.hip: contraction by pragma inside the file


def _newer(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(map(str, cmd)) + "\n" + r.stdout + r.stderr)
        raise RuntimeError(f"build step failed: {cmd[0]} ... {cmd[-1]}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


def build_library(force: bool = False, verbose: bool = False) -> Path:
    srcs = sorted(CSRC.glob("*.hip"))
    hdrs = sorted(CSRC.glob("*.hpp")) + sorted((ROOT / "include").glob("*.h"))
    objdir = CSRC / "build"
    objdir.mkdir(exist_ok=True)
    LIBDIR.mkdir(exist_ok=True)
    jobs = []
    objs = []
    for s in srcs:
        o = objdir / (s.stem + ".o")
        objs.append(o)
        if force or _newer(o, [s, Path(__file__)] + hdrs):  # the flags live in this file: a changed flag set rebuilds
            jobs.append([HIPCC, *HIP_FLAGS, *HIP_FLAGS_PER_SOURCE.get(s.name, []), "-c", str(s), "-o", str(o)])
    if jobs:
        if verbose:
            print(f"[build] compiling {len(jobs)} HIP source(s) for gfx950", flush=True)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(_run, jobs))
    if force or jobs or _newer(LIB, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB)])
        if verbose:
            print(f"[build] linked {LIB}", flush=True)
    return LIB


def build_variant(name: str, defines: dict, source: str = "ba.hip") -> Path:
    """A/B measurements of build-time choices: `lib/variants/libsnake_hip_<name>.so` = the library with `source` recompiled
    under -D<key>=<value>.  Select it with SNK_HIP_LIB=<path> (snake_slam_amd/_lib.py)."""
    build_library()
    vdir = LIBDIR / "variants"
    vdir.mkdir(exist_ok=True)
    obj = CSRC / "build" / f"{Path(source).stem}_{name}.o"
    _run([HIPCC, *HIP_FLAGS, *HIP_FLAGS_PER_SOURCE.get(source, []), *[f"-D{k}={v}" for k, v in defines.items()], "-c", str(CSRC / source),
          "-o", str(obj)])
    others = [CSRC / "build" / (s.stem + ".o") for s in sorted(CSRC.glob("*.hip")) if s.name != source]
    out = vdir / f"libsnake_hip_{name}.so"
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", str(obj), *map(str, others), "-o", str(out)])
    return out


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, verbose=True)
