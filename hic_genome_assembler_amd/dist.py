"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm; "gloo"
in the CPU tests).

What shards in this path: whole maps (independent genomes) and, inside one map,
the chromosomes of Part 2.  The UPGMA chain does not.  Work units are therefore dealt to ranks with
no collective in the data path; the only exchanges are a barrier, a MAX-reduction of the elapsed
time and an object gather of the (tiny) per-unit results.
"""
from __future__ import annotations

import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend: str = "nccl", device=None):
    """Initialise the default process group from the torchrun environment (no-op for one process)."""
    import torch.distributed as dist
    rank, world, _local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kwargs = {}
        if backend == "nccl" and device is not None:
            kwargs["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world


def units_of_rank(n_units: int, rank: int, world: int):
    """Deal work units (maps, chromosomes) round-robin: unit u belongs to rank u % world."""
    return [u for u in range(n_units) if u % world == rank]


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(seconds: float, device=None) -> float:
    """The bench contract's timing rule: the slowest rank defines the step."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_results(local: dict) -> dict:
    """Union of {unit: result} dictionaries over all ranks (results are a few hundred bytes each)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return dict(local)
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, local)
    out = {}
    for p in parts:
        out.update(p)
    return out
