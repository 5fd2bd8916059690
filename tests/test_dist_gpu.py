"""GPU: the native multi-GPU result gather (include/snake_hip.h snk_dist_*, snake_slam_amd/csrc/dist.hip: RCCL called directly, no
torch.distributed / MPI in the process) through the C++ adaptor's snake_hip::Dist -- one process per rank, exactly how a Snake-SLAM
process per GPU would use it (BASELINE config 5; the block is the TUM trajectory of Snake/System/System.cpp:546-563).
World size 1 runs on any GPU box; world size = min(#GPUs, 8) runs where there are at least two devices (RCCL refuses two ranks on one device)."""
import os
import subprocess
import uuid
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def build_driver(out_dir: Path) -> Path:
    lib = ROOT / "snake_slam_amd" / "lib"
    exe = out_dir / "dist_driver"
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", f"-I{ROOT / 'include'}", f"-I{ROOT / 'snake_slam_amd' / 'cpp'}",
           str(ROOT / "tests" / "cpp" / "dist_driver.cpp"), f"-L{lib}", "-lsnake_hip", "-L/opt/rocm/lib", "-lamdhip64",
           f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def run_world(tmp_path: Path, world: int):
    exe = build_driver(tmp_path)
    rendezvous = tmp_path / f"rccl_id_{uuid.uuid4().hex}"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([str(exe), str(rendezvous), str(r), str(world), str(r), str(tmp_path / f"out_{r}.txt")], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    for r, p in enumerate(procs):
        try:
            _, err = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, f"rank {r}: {err}"
    assert not rendezvous.exists(), "rank 0 removes the rendezvous file"
    for r in range(world):
        lines = (tmp_path / f"out_{r}.txt").read_text().splitlines()
        assert lines[0].startswith("rccl ") and lines[0].endswith(f"world {world}")
        at = 1
        for q in range(world):
            assert lines[at] == f"rank {q} rows {37 + 11 * q} identical stats {37 + 11 * q} {1000 + q} {250 * q}", (r, lines[at])
            at += 1
            if q == r:  # this rank's own rows in the reference's text form
                rows = np.array([[float(v) for v in ln.split()] for ln in lines[at:at + 37 + 11 * q]])
                i = np.arange(37 + 11 * q)
                a = 0.01 * i + q
                assert rows.shape == (37 + 11 * q, 8)
                assert np.allclose(rows[:, 0], 1403636579.0 + 0.05 * i, rtol=0, atol=1e-5) and np.allclose(rows[:, 1], a, atol=1e-12)
                assert np.allclose(rows[:, 6] ** 2 + rows[:, 7] ** 2, 1.0, atol=1e-12)
                at += 37 + 11 * q


def test_native_rccl_gather_world_1(tmp_path):
    run_world(tmp_path, 1)


def test_native_rccl_gather_all_devices(tmp_path):
    import torch

    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("one GPU: RCCL refuses two ranks on one device")
    run_world(tmp_path, min(n, 8))


def test_python_binding_world_1():
    """The same entry points through ctypes, inside a process that has torch's RCCL mapped: dlopen by SONAME must resolve to that
    copy (one RCCL, one HIP runtime per process)."""
    import ctypes as C

    from snake_slam_amd import _lib

    lib = _lib.load()
    ver = C.c_int(0)
    _lib.check(lib.snk_dist_rccl_version(C.byref(ver)), "snk_dist_rccl_version")
    assert ver.value > 20000
    ident = (C.c_uint8 * 128)()
    _lib.check(lib.snk_dist_get_unique_id(ident), "snk_dist_get_unique_id")
    h = C.c_void_p()
    _lib.check(lib.snk_dist_init(ident, 0, 1, 0, C.byref(h)), "snk_dist_init")
    try:
        send = np.arange(1000, dtype=np.float64)
        recv = np.zeros(1000, np.float64)
        _lib.check(lib.snk_dist_all_gather(h, send.ctypes.data, send.nbytes, recv.ctypes.data), "snk_dist_all_gather")
        assert np.array_equal(send, recv)
        out = C.c_int64(0)
        _lib.check(lib.snk_dist_max_i64(h, 1234567890123, C.byref(out)), "snk_dist_max_i64")
        assert out.value == 1234567890123
    finally:
        lib.snk_dist_destroy(h)
    assert len(_lib.hip_runtimes_mapped()) == 1, _lib.hip_runtimes_mapped()
    rccl = {ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln}
    assert len(rccl) <= 1, rccl
