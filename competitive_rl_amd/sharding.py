"""Multi-GPU layout: envs are independent, so the path shards with no data-path exchange.

Rank g of W owns the contiguous global env range ``shard_of(total, W, g)``; the serve RNG
is keyed by GLOBAL env id (crl_opts.env_id_base), so the union of the shards is
bit-identical to one big unsharded env batch.  The only collective is the optional
all-gather of BASELINE config #5 that leaves every rank with every shard's step outputs
(RCCL over xGMI via torch.distributed "nccl"; "gloo" in the CPU tests).

The step's tensors (observation -- or the 64-byte frame descriptors it was drawn from --, rewards, done flags ...) are
PACKED into one byte buffer and gathered by ONE ``all_gather_into_tensor`` per step: on the 8-GPU xGMI mesh a direct all-gather
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
    """One packed all-gather per step; ``wait()`` returns every shard's step outputs in global env order, each
    (world * n_shard, ...).  Shards must be equal-sized (weak scaling).

    mode="obs"          the message is [observation | rewards | done ...].  ``obs_slot(shape, dtype)`` hands out the
                        observation's place INSIDE the send buffer, so that the env can draw straight into it
                        (``env.step_device(a, obs_out=slot)``): no pack copy of the big tensor.
    mode="descriptors"  the message is [frame descriptors (64 bytes per env) | rewards | done ...]; after the collective
                        every rank re-draws all shards' observations locally (``env.render_descriptors``): a 65 536-env
                        shard ships 4 MB instead of 3.7 GB (fused 4-stack) or 13.2 GB (raw) over its xGMI links.

    The small tensors are packed on the CALLER's stream, so a later step that rewrites them cannot be seen by the pack; only
    the collective itself runs on the side stream."""

    def __init__(self, group=None, overlap=True, mode="obs"):
        assert mode in ("obs", "descriptors")
        self.group, self.overlap, self.mode = group, overlap, mode
        self.world = dist.get_world_size(group)
        self.send = self.recv = self.stream = None
        self.meta = self.work = self.env = None
        self._slot = None

    def _buffers(self, nbytes, dev):
        if self.send is None or self.send.numel() != nbytes or self.send.device != dev:
            self.send = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            self.recv = torch.empty(nbytes * self.world, dtype=torch.uint8, device=dev)
            self.stream = torch.cuda.Stream(device=dev) if (dev.type == "cuda" and self.overlap) else None

    def obs_slot(self, shape, dtype, device, rest_bytes=None):
        """mode="obs": a tensor of the observation's shape that IS the head of a send buffer (the other tensors of a step
        must fit in ``rest_bytes`` after it; default: 32 bytes per env of the shard + 4 KB -- rewards (n, 2) float32, done flags and
        two more per-env words.  A fixed 4 KB, as it was until round 5, is too small from 410 envs per shard on: found by the
        one-rank RCCL test).  Call it before EVERY step: there are two send buffers and the slots alternate, so that step t+1 can
        be drawn while the collective of step t still reads the other one."""
        if rest_bytes is None:
            rest_bytes = 4096 + 32 * int(shape[0])
        nb = (int(torch.Size(shape).numel()) * torch.empty((), dtype=dtype).element_size() + 15) // 16 * 16
        total, dev = nb + (rest_bytes + 15) // 16 * 16, torch.device(device)
        if getattr(self, "_sends", None) is None or self._sends[0].numel() != total or self._sends[0].device != dev:
            self._sends = [torch.zeros(total, dtype=torch.uint8, device=dev) for _ in range(2)]
            self.recv = torch.empty(total * self.world, dtype=torch.uint8, device=dev)
            self.stream = torch.cuda.Stream(device=dev) if (dev.type == "cuda" and self.overlap) else None
            self._k = 0
        self._k ^= 1
        self.send = self._sends[self._k]
        self._slot = self.send[:nb].view(dtype)[:torch.Size(shape).numel()].view(shape)
        return self._slot

    def launch(self, tensors, env=None):
        """tensors: the step's outputs (observation first in mode="obs"; in mode="descriptors" pass (rewards, done, ...) and
        the env whose descriptors are to be shipped).

        Order: ``launch(t) ... wait(t) ... launch(t+1)``.  A launch that finds the previous collective still pending first
        orders the caller's stream behind it (its result is dropped: ``recv`` is about to be rewritten) -- the pack below
        writes the send buffer that collective reads, and ``_buffers`` may replace both buffers on a size change."""
        if self.work is not None:
            self.wait(materialize=False)
        if self.mode == "descriptors":
            assert env is not None, 'mode="descriptors" needs the env'
            self.env = env
            desc_bytes = 8 * env.num_envs * 8
            small = [t.contiguous() for t in tensors]
            offs, nb_small = _layout(small)
            dev = small[0].device
            self._buffers(desc_bytes + nb_small, dev)
            env.obs_descriptors(out=self.send[:desc_bytes].view(torch.int64).view(8, env.num_envs))  # the library writes its slot
            pack = [(desc_bytes + o, t) for o, t in zip(offs, small)]
            self.meta = [(desc_bytes + o, t.numel() * t.element_size(), tuple(t.shape), t.dtype) for o, t in zip(offs, small)]
            self.desc_bytes = desc_bytes
        else:
            tensors = list(tensors)
            aliased = self._slot is not None and tensors[0].data_ptr() == self._slot.data_ptr()
            small = [t.contiguous() for t in (tensors[1:] if aliased else tensors)]
            offs, nb = _layout(small)
            head = (self._slot.numel() * self._slot.element_size() + 15) // 16 * 16 if aliased else 0
            dev = tensors[0].device
            if aliased:
                assert head + nb <= self.send.numel(), "obs_slot(rest_bytes=...) too small for the step's other tensors"
            else:
                self._buffers(nb, dev)
            pack = [(head + o, t) for o, t in zip(offs, small)]
            self.meta = ([(0, self._slot.numel() * self._slot.element_size(), tuple(self._slot.shape), self._slot.dtype)] if aliased else []) + \
                        [(head + o, t.numel() * t.element_size(), tuple(t.shape), t.dtype) for o, t in zip(offs, small)]
        for o, t in pack:  # on the caller's stream, behind the kernels that produced the tensors, ahead of whatever rewrites them
            self.send[o:o + t.numel() * t.element_size()].copy_(t.reshape(-1).view(torch.uint8), non_blocking=True)
        if self.stream is not None:
            ready = torch.cuda.Event()
            ready.record()                      # the step's kernels AND the pack, on the caller's stream
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ready)
                self.work = dist.all_gather_into_tensor(self.recv, self.send, group=self.group, async_op=True)
        else:
            self.work = dist.all_gather_into_tensor(self.recv, self.send, group=self.group, async_op=True)

    def wait(self, materialize=True):
        """Blocks the caller's stream on the collective.  ``materialize=False`` only orders the streams (the gathered bytes
        stay in ``self.recv``, rank-major); otherwise the tensors are sliced out per field, in global env order -- in
        mode="descriptors" the first one is the global observation, re-drawn here from every shard's descriptors."""
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
        if self.mode == "descriptors":
            env, n = self.env, self.env.num_envs
            shape = (self.world * n,) + tuple(env._obs_shape[1:])
            if getattr(self, "_global_obs", None) is None or tuple(self._global_obs.shape) != shape:
                self._global_obs = torch.empty(shape, dtype=env._buf_dtype, device=env.device)  # reused: valid until the next wait()
            obs = self._global_obs
            for r in range(self.world):
                desc = per_rank[r, :self.desc_bytes].contiguous().view(torch.int64)
                env.render_descriptors(desc, out=obs[r * n:(r + 1) * n])
            out.append(obs)
        for o, nb, shape, dtype in self.meta:
            out.append(per_rank[:, o:o + nb].contiguous().view(dtype).view(self.world * shape[0], *shape[1:]))
        return tuple(out)


def all_gather_step(tensors, group=None):
    """Blocking form: the tuple of per-shard tensors, concatenated along dim 0 in global env order on every rank, moved by
    a single packed collective."""
    g = StepGather(group, overlap=False)
    g.launch(tensors)
    return g.wait()
