"""CPU: the N>1 host logic (sharding, timing reduction, result gather) with world_size 2 on gloo."""
import os
import socket
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["REPO"])
import torch
from snake_slam_amd import parallel as P
dev = torch.device("cpu")
rank, world = P.init_distributed(dev, backend="gloo")
assert world == 2 and P.is_distributed()
# strong-scaling split of 11 BA windows / frames
mine = P.shard_range(11, rank, world)
counts = [len(P.shard_range(11, r, world)) for r in range(world)]
assert counts == [6, 5] and sum(counts) == 11
assert list(P.shard_range(11, 0, 2))[-1] + 1 == list(P.shard_range(11, 1, 2))[0]
# timing: max over ranks
t = P.max_over_ranks(1.0 + rank, dev)
assert t == 2.0
# result gather: one block per rank
blk = torch.zeros(P.RESULT_BLOCK, dtype=torch.float64)
blk[0] = len(mine); blk[1] = 100.0 * (rank + 1); blk[4] = 0.5 + rank
blocks = P.gather_result_blocks(blk)
assert len(blocks) == 2 and [float(b[0]) for b in blocks] == [6.0, 5.0] and float(blocks[1][1]) == 200.0
total_units = sum(float(b[0]) for b in blocks)
assert total_units / t == 5.5
# sequence mode: every rank contributes its padded TUM trajectory block {n, [t tx ty tz qx qy qz qw] x frames}
import numpy as np
from snake_slam_amd.sequence import trajectory_block, trajectory_rows, inverse_pose_tum
n_mine = 5 + 2 * rank  # sequences of different lengths, blocks padded to the longest (9)
rows = np.array([np.concatenate([[float(t)], inverse_pose_tum([0, 0, 0, 1.0, -0.01 * t * (rank + 1), 0, 0])]) for t in range(n_mine)])
blk = torch.from_numpy(trajectory_block(rows, 9))
assert blk.numel() == 1 + 8 * 9
got = P.gather_blocks(blk)
assert len(got) == 2
for r in range(2):
    tr = trajectory_rows(got[r].numpy())
    assert len(tr) == 5 + 2 * r
    assert np.allclose(tr[:, 0], np.arange(5 + 2 * r))
    assert np.allclose(tr[:, 1], 0.01 * np.arange(5 + 2 * r) * (r + 1))   # camera position = inverse of the pose translation
    assert np.allclose(tr[:, 4:], [0, 0, 0, 1])
P.barrier()
P.shutdown()
print("rank", rank, "ok")
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   REPO=str(ROOT))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


def test_shard_range_properties():
    from snake_slam_amd.parallel import shard_range

    for n in (0, 1, 7, 8, 256, 1001):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert sum(len(p) for p in parts) == n
            flat = [i for p in parts for i in p]
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_trajectory_block_roundtrip_and_tum_layout(tmp_path):
    import numpy as np

    from snake_slam_amd import synth
    from snake_slam_amd.sequence import inverse_pose_tum, trajectory_block, trajectory_rows, write_tum

    rng = np.random.default_rng(5)
    poses = [synth.random_pose(rng) for _ in range(7)]
    rows = np.array([np.concatenate([[0.05 * i], inverse_pose_tum(p)]) for i, p in enumerate(poses)])
    blk = trajectory_block(rows, 10)
    assert blk.shape == (81,) and blk[0] == 7 and not blk[1 + 56:].any()
    assert np.array_equal(trajectory_rows(blk), rows)
    # inverse pose: R^T, -R^T t (System.cpp:553-555), unit quaternion with w >= 0
    for p, r in zip(poses, rows):
        R = synth.quat_to_R(p[:4])
        assert np.allclose(r[1:4], -R.T @ p[4:]) and np.allclose(synth.quat_to_R(r[4:8]), R.T) and r[7] >= 0
    write_tum(tmp_path / "traj.txt", rows)
    back = np.loadtxt(tmp_path / "traj.txt")
    assert back.shape == (7, 8) and np.allclose(back, rows, rtol=1e-14, atol=0)
    try:
        trajectory_block(rows, 3)
        assert False
    except ValueError:
        pass
