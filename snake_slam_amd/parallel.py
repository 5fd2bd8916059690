"""Multi-GPU plumbing of the hot path: one process per GPU, no data-path collective.

The path shards over independent units (frames of a batch, sequences, disjoint BA windows —
SURVEY.md §8e); the only communication is a result gather after the work is done.  Backend
"nccl" is RCCL on ROCm (xGMI); "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

RESULT_BLOCK = 8  # doubles per rank: units, keypoints, stereo matches, bf pairs, seconds, 3 spare


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_distributed(device: torch.device | None = None, backend: str | None = None) -> tuple[int, int]:
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1)."""
    rank, world, _ = env_rank_world()
    # SNK_DIST_FORCE=1: a process group also for ONE rank, so that a one-GPU box exercises the same RCCL initialisation, barrier,
    # all_reduce and all_gather calls (device tensors, device_id binding) that the N > 1 runs make -- the only RCCL rehearsal
    # possible without a multi-GPU lease (tests/test_bench_contract_gpu.py)
    if world <= 1 and not os.environ.get("SNK_DIST_FORCE"):
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        # torchrun / the driver export MASTER_PORT for every multi-rank job.  Without a launcher only the one-rank rehearsal gets here:
        # it takes a port the kernel hands out (a fixed default made two jobs on one node collide); ranks of a real job cannot agree
        # on a port by themselves, so that is an error rather than a guess.
        if world > 1:
            raise RuntimeError("WORLD_SIZE > 1 but MASTER_PORT is not set: launch with torch.distributed.run (or export MASTER_ADDR / MASTER_PORT)")
        import socket

        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        # SNK_DIST_BACKEND=gloo: rehearsal of the multi-rank code on a box with one GPU (every rank on the same device, see
        # bench.py SNK_BENCH_DEVICE); RCCL refuses two ranks on one device
        backend = os.environ.get("SNK_DIST_BACKEND") or ("nccl" if (device is not None and device.type == "cuda") else "gloo")
    kw = {"device_id": device} if backend == "nccl" and device is not None else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or bool(os.environ.get("SNK_DIST_FORCE")))


def describe() -> dict:
    """What carried the barriers / gathers of this run (goes into bench.py's line): world size as torch.distributed sees it and
    the backend ("nccl" = RCCL over xGMI; "gloo" only in the one-GPU rehearsal and the CPU tests; "none" for one rank)."""
    if not is_distributed():
        return {"world_size": 1, "backend": "none"}
    return {"world_size": int(dist.get_world_size()), "backend": str(dist.get_backend())}


def barrier() -> None:
    if is_distributed():
        dist.barrier()


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced split of n_items independent units (strong scaling: the frames of a
    sequence list / the windows of a BA batch).  The first n_items % world ranks get one more."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def _host_backend() -> bool:
    return is_distributed() and dist.get_backend() == "gloo"


def max_over_ranks(value: float, device: torch.device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if _host_backend() else device)
    if is_distributed():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_result_blocks(block: torch.Tensor) -> list[torch.Tensor]:
    """One fixed-size block per rank -> list of all ranks' blocks on every rank (all_gather)."""
    assert block.numel() == RESULT_BLOCK and block.dtype == torch.float64
    if not is_distributed():
        return [block]
    return _all_gather(block)


def _all_gather(block: torch.Tensor) -> list[torch.Tensor]:
    src = block.cpu() if _host_backend() else block  # gloo gathers host tensors
    out = [torch.zeros_like(src) for _ in range(dist.get_world_size())]
    dist.all_gather(out, src)
    return [o.to(block.device) for o in out]


def gather_blocks(block: torch.Tensor) -> list[torch.Tensor]:
    """all_gather of one equally sized 1-D block per rank -- the per-rank trajectory block of the sequence mode
    (`{n, [t, tx, ty, tz, qx, qy, qz, qw] x frames}`; reference Snake/System/System.cpp:552-563).
    Every rank must pass the same number of elements (blocks are padded to the longest sequence)."""
    assert block.dim() == 1
    if not is_distributed():
        return [block]
    return _all_gather(block)


def shutdown() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
