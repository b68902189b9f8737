"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm; "gloo"
in the CPU tests).

What shards in this path: whole maps (independent genomes) and, inside one map,
(a) the row-independent stages of Part 1 - row sums, the per-row argsort / rank matrix, the per-row
counts of every cut and filter scan - by a cyclic row partition (hicmi_set_row_shard), with ONE
all-gather of the owned entries per vector (``gather_owned``: N / world row sums, or N / world flags
per scan), and (b) the chromosomes of Part 2, with one object all-gather of the ordered lists.  The
UPGMA chain does not shard (replicated).  Besides those: a barrier and a MAX-reduction of the time.
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


def gather_owned(local, first_row: int, rank: int, world: int):
    """All-gather of a per-row vector whose entries were computed by the rank that owns the row.

    ``local``: 1-D NumPy array over the rows ``first_row, first_row + 1, ...``; entry t is meaningful on this rank
    iff ``(first_row + t) % world == rank`` (the cyclic row partition of hicmi_set_row_shard).  Every rank sends its
    owned entries (N / world of them); the result has every entry filled in, identical on all ranks.
    Backend "nccl" (RCCL): the pieces travel as device tensors; "gloo": as host tensors."""
    import numpy as np
    import torch
    import torch.distributed as dist
    local = np.ascontiguousarray(local)
    if world <= 1 or not (dist.is_available() and dist.is_initialized()):
        return local
    n = len(local)
    chunk = (n + world - 1) // world                      # owned entries per rank, padded to a common length
    t0 = (rank - first_row) % world
    mine = local[t0::world]
    send = np.zeros(max(chunk, 1), dtype=local.dtype)
    send[:len(mine)] = mine
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t_send = torch.from_numpy(send).to(dev)
    t_recv = torch.empty(world * len(send), dtype=t_send.dtype, device=dev)
    dist.all_gather_into_tensor(t_recv, t_send)
    parts = t_recv.cpu().numpy().reshape(world, len(send))
    out = np.empty_like(local)
    for r in range(world):
        tr = (r - first_row) % world
        cnt = len(range(tr, n, world))
        out[tr::world] = parts[r, :cnt]
    return out
