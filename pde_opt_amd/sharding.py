"""Sharding of batched episodes over the GPUs of a node.

Environments are independent units: rank r advances environments [lo, hi) on its own GPU with NO
collective inside the step; only per-environment scalars (rewards, termination flags) are gathered
afterwards.  ``gather_per_env`` works on any torch.distributed backend: 'nccl' (= RCCL over xGMI)
with one process per GPU, 'gloo' in the CPU tests.
"""

from __future__ import annotations

import numpy as np


def shard_envs(total: int, world: int, rank: int):
    """Contiguous, balanced block [lo, hi) of environment indices owned by ``rank``."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_per_env(local_values, total: int, group=None, device=None) -> np.ndarray:
    """All ranks receive the per-environment values of the whole job, in environment order."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [shard_envs(total, world, r)[1] - shard_envs(total, world, r)[0] for r in range(world)]
    width = max(counts)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
    send = torch.zeros(width, dtype=torch.float64, device=device)
    local = np.asarray(local_values, dtype=np.float64).ravel()
    if local.size != counts[rank]:
        raise ValueError(f"rank {rank} owns {counts[rank]} environments, got {local.size} values")
    send[: local.size] = torch.from_numpy(local).to(device)
    recv = torch.empty(world * width, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.cpu().numpy().reshape(world, width)
    return np.concatenate([recv[r, : counts[r]] for r in range(world)])
