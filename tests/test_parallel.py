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
