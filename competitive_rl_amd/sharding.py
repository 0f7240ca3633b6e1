"""Multi-GPU layout: envs are independent, so the path shards with no data-path exchange.

Rank g of W owns the contiguous global env range ``shard_of(total, W, g)``; the serve RNG
is keyed by GLOBAL env id (crl_opts.env_id_base), so the union of the shards is
bit-identical to one big unsharded env batch.  The only collective is the optional
all-gather of BASELINE config #5 that leaves every rank with every shard's step outputs
(RCCL over xGMI via torch.distributed "nccl"; "gloo" in the CPU tests).

The step's tensors (observation, rewards, done flags ...) are PACKED into one byte buffer and
gathered by ONE ``all_gather_into_tensor`` per step: on the 8-GPU xGMI mesh a direct all-gather
is bound by one link per peer (shard_bytes / ~153 GB/s), so what matters is that the
big observation message is not followed by two latency-bound tiny ones.  ``StepGather`` runs
that collective on a side stream so that gather(t) overlaps simulate(t+1) -- legitimate whenever
the actions of step t+1 do not depend on the gathered result (the benchmark; evaluation of fixed
policies); a learner that needs the global batch calls ``wait()`` first.
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


def _layout(tensors):
    """Byte offsets of the tensors inside the packed per-rank message (each padded to 16 bytes)."""
    offs, o = [], 0
    for t in tensors:
        offs.append(o)
        o += (t.numel() * t.element_size() + 15) // 16 * 16
    return offs, o


class StepGather:
    """One packed all-gather per step.  ``launch(tensors)`` copies the shard's tensors into the send buffer and starts the
    collective (on a side stream when the tensors live on a GPU); ``wait()`` returns the gathered tensors in global env
    order, each (world * n_shard, ...).  Shards must be equal-sized (weak scaling)."""

    def __init__(self, group=None, overlap=True):
        self.group, self.overlap = group, overlap
        self.world = dist.get_world_size(group)
        self.send = self.recv = self.stream = None
        self.meta = self.work = None

    def launch(self, tensors):
        tensors = [t.contiguous() for t in tensors]
        dev = tensors[0].device
        offs, nbytes = _layout(tensors)
        if self.send is None or self.send.numel() != nbytes or self.send.device != dev:
            self.send = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            self.recv = torch.empty(nbytes * self.world, dtype=torch.uint8, device=dev)
            self.stream = torch.cuda.Stream(device=dev) if (dev.type == "cuda" and self.overlap) else None
        self.meta = [(o, t.numel() * t.element_size(), tuple(t.shape), t.dtype) for o, t in zip(offs, tensors)]

        def pack_and_gather():
            for o, t in zip(offs, tensors):
                self.send[o:o + t.numel() * t.element_size()].copy_(t.reshape(-1).view(torch.uint8), non_blocking=True)
            return dist.all_gather_into_tensor(self.recv, self.send, group=self.group, async_op=True)

        if self.stream is not None:
            ready = torch.cuda.Event()
            ready.record()                      # the step's kernels, on the caller's stream
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ready)
                self.work = pack_and_gather()
        else:
            self.work = pack_and_gather()

    def wait(self, materialize=True):
        """Blocks the caller's stream on the collective.  ``materialize=False`` only orders the streams (the gathered bytes
        stay in ``self.recv``, rank-major); otherwise the tensors are sliced out per field, in global env order."""
        if self.work is None:
            return None
        self.work.wait()
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        self.work = None
        if not materialize:
            return None
        per_rank = self.recv.view(self.world, -1)
        out = []
        for o, nb, shape, dtype in self.meta:
            out.append(per_rank[:, o:o + nb].contiguous().view(dtype).view(self.world * shape[0], *shape[1:]))
        return tuple(out)


def all_gather_step(tensors, group=None):
    """Blocking form: the tuple of per-shard tensors, concatenated along dim 0 in global env order on every rank, moved by
    a single packed collective."""
    g = StepGather(group, overlap=False)
    g.launch(tensors)
    return g.wait()
