"""Multi-GPU layout: envs are independent, so the path shards with no data-path exchange.

Rank g of W owns the contiguous global env range ``shard_of(total, W, g)``; the serve RNG
is keyed by GLOBAL env id (crl_opts.env_id_base), so the union of the shards is
bit-identical to one big unsharded env batch.  The only collective is the optional
all-gather that concatenates per-step outputs on every rank (RCCL over xGMI via
torch.distributed "nccl"; "gloo" in the CPU tests).
"""
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class ShardSpec:
    total: int
    world: int
    rank: int
    base: int   # global id of this shard's env 0
    count: int  # envs in this shard


def shard_of(total, world, rank):
    """Contiguous near-equal split; the first ``total % world`` ranks get one extra env."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    q, r = divmod(int(total), int(world))
    count = q + (1 if rank < r else 0)
    base = rank * q + min(rank, r)
    return ShardSpec(int(total), int(world), int(rank), base, count)


def all_gather_step(tensors, group=None):
    """All-gather a tuple of per-shard tensors along dim 0 (shards must be equal-sized, as
    in weak scaling).  One collective per tensor; returns the concatenated tensors in
    global env order on every rank."""
    world = dist.get_world_size(group)
    out = []
    for t in tensors:
        t = t.contiguous()
        g = torch.empty((world * t.shape[0], *t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(g, t, group=group)
        out.append(g)
    return tuple(out)
