"""Multi-GPU layout: envs are independent, so the path shards with no data-path exchange.

Rank g of W owns the contiguous global env range ``shard_of(total, W, g)``; the serve RNG
is keyed by GLOBAL env id (crl_opts.env_id_base), so the union of the shards is
bit-identical to one big unsharded env batch.  The only collective is the optional
all-gather of BASELINE config #5 that leaves every rank with every shard's step outputs
(RCCL over xGMI via torch.distributed "nccl"; "gloo" in the CPU tests).

The observation is gathered by an ``all_gather_into_tensor`` whose receive buffer IS the global (world * n, ...) tensor
(round 6: zero-copy on both sides of the collective); the step's small tensors (rewards, done flags ... -- or, in
mode="descriptors", the 64-byte frame descriptors an observation is drawn from) are PACKED into one byte buffer and gathered
by one more: on the 8-GPU xGMI mesh a direct all-gather is bound by one link per peer (shard_bytes / ~153 GB/s), so
what matters is that the big message is not followed by a train of latency-bound tiny ones.  ``StepGather`` runs
the collectives on a side stream so that gather(t) overlaps simulate(t+1) -- legitimate whenever
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
    """The step's gather; ``wait()`` returns every shard's step outputs in global env order, each (world * n_shard, ...).
    Shards must be equal-sized (weak scaling).

    mode="obs"          TWO collectives per step: the observation -- the only big tensor -- is gathered by an
                        ``all_gather_into_tensor`` of its own whose RECEIVE buffer is the global ``(world * n, ...)`` tensor
                        (shards are contiguous global ranges, so rank-major IS global env order): what ``wait()`` returns
                        for it aliases that buffer, nothing is copied behind the collective (round 5 packed everything into
                        one message per rank and sliced the observation out of the rank-strided receive buffer with a
                        29.6 GB-read + 29.6 GB-write ``.contiguous()`` at 8 x 65 536 fused envs).  ``obs_slot(shape, dtype)``
                        hands out the send side, so that the env can draw straight into it (``env.step_device(a,
                        obs_out=slot)``): no copy in front of the collective either.  Rewards, done flags ... travel packed in
                        a second, latency-bound message of a few hundred KB (the 24 ms of link time of the first hide it).
    mode="descriptors"  one packed message [frame descriptors (64 bytes per env) | rewards | done ...]; after the collective
                        every rank re-draws all shards' observations locally (``env.render_descriptors``): a 65 536-env
                        shard ships 4 MB instead of 3.7 GB (fused 4-stack) or 13.2 GB (raw) over its xGMI links.

    The gathered observation is valid until the next ``launch()`` (one receive tensor; the collective that rewrites it is
    ordered behind everything the caller enqueued before that launch).  The small tensors are packed on the CALLER's stream, so
    a later step that rewrites them cannot be seen by the pack; only the collectives run on the side stream."""

    def __init__(self, group=None, overlap=True, mode="obs"):
        assert mode in ("obs", "descriptors")
        self.group, self.overlap, self.mode = group, overlap, mode
        self.world = dist.get_world_size(group)
        self.send = self.recv = self.stream = None       # the packed (small) message
        self.obs_send = self.obs_recv = None             # mode="obs": the observation's own collective
        self.meta = self.work = self.env = None
        self._slot = self._sends = None

    def _side_stream(self, dev):
        if self.stream is None or self._stream_dev != dev:
            self.stream = torch.cuda.Stream(device=dev) if (dev.type == "cuda" and self.overlap) else None
            self._stream_dev = dev

    def _buffers(self, nbytes, dev):
        if self.send is None or self.send.numel() != nbytes or self.send.device != dev:
            self.send = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            self.recv = torch.empty(nbytes * self.world, dtype=torch.uint8, device=dev)
        self._side_stream(dev)

    def _obs_recv(self, shape, dtype, dev):
        want = (self.world * int(shape[0]),) + tuple(int(d) for d in shape[1:])
        if self.obs_recv is None or tuple(self.obs_recv.shape) != want or self.obs_recv.dtype != dtype or self.obs_recv.device != dev:
            self.obs_recv = torch.empty(want, dtype=dtype, device=dev)
        return self.obs_recv

    def obs_slot(self, shape, dtype, device, rest_bytes=None):
        """mode="obs": a tensor of the observation's shape that IS the send buffer of the observation's collective.  Call it
        before EVERY step: there are two and they alternate, so that step t+1 can be drawn while the collective of step t still
        reads the other one.  (``rest_bytes`` is accepted for callers of round 5's single packed message and ignored: the
        other tensors of a step have a message of their own now.)"""
        dev = torch.device(device)
        shape = tuple(int(d) for d in shape)
        if self._sends is None or tuple(self._sends[0].shape) != shape or self._sends[0].dtype != dtype or self._sends[0].device != dev:
            self._sends = [torch.zeros(shape, dtype=dtype, device=dev) for _ in range(2)]
            self._k = 0
        self._k ^= 1
        self._slot = self._sends[self._k]
        return self._slot

    def launch(self, tensors, env=None):
        """tensors: the step's outputs (observation first in mode="obs"; in mode="descriptors" pass (rewards, done, ...) and
        the env whose descriptors are to be shipped).

        Order: ``launch(t) ... wait(t) ... launch(t+1)``.  A launch that finds the previous collective still pending first
        orders the caller's stream behind it (its result is dropped: the receive buffers are about to be rewritten) -- the pack
        below writes the send buffer that collective reads, and ``_buffers`` may replace both buffers on a size change."""
        if self.work is not None:
            self.wait(materialize=False)
        obs_in = None
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
            obs, small = tensors[0], [t.contiguous() for t in tensors[1:]]
            dev = obs.device
            aliased = self._slot is not None and obs.data_ptr() == self._slot.data_ptr() and tuple(obs.shape) == tuple(self._slot.shape)
            if aliased:
                obs_in = self._slot  # drawn in place by the env: nothing to copy
            else:
                # a tensor of the caller's own: copied (on the caller's stream) into a send buffer, so that whatever rewrites it
                # after this launch cannot reach the message
                obs_in = self.obs_slot(obs.shape, obs.dtype, dev)
                obs_in.copy_(obs, non_blocking=True)
                self._slot = None  # (that slot was not handed out: the next obs_slot() alternates as usual)
            self._obs_recv(obs_in.shape, obs_in.dtype, dev)
            offs, nb = _layout(small)
            self._buffers(max(nb, 16), dev)
            pack = list(zip(offs, small))
            self.meta = [(o, t.numel() * t.element_size(), tuple(t.shape), t.dtype) for o, t in zip(offs, small)]
        for o, t in pack:  # on the caller's stream, behind the kernels that produced the tensors, ahead of whatever rewrites them
            self.send[o:o + t.numel() * t.element_size()].copy_(t.reshape(-1).view(torch.uint8), non_blocking=True)

        def collectives():
            works = []
            if obs_in is not None:  # receive buffer = the global tensor itself
                works.append(dist.all_gather_into_tensor(self.obs_recv, obs_in, group=self.group, async_op=True))
            works.append(dist.all_gather_into_tensor(self.recv, self.send, group=self.group, async_op=True))
            return works

        if self.stream is not None:
            ready = torch.cuda.Event()
            ready.record()                      # the step's kernels, the pack AND every reader of the last result, on the caller's stream
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ready)
                self.work = collectives()
        else:
            self.work = collectives()
        self._has_obs = obs_in is not None

    def wait(self, materialize=True):
        """Blocks the caller's stream on the collectives.  ``materialize=False`` only orders the streams (the gathered
        observation is in ``self.obs_recv`` all the same, the small fields rank-major in ``self.recv``); otherwise the tuple of
        global tensors in the order they were launched -- the observation first: in mode="obs" the receive buffer itself
        (zero-copy), in mode="descriptors" re-drawn here from every shard's descriptors."""
        if self.work is None:
            return None
        for w in self.work:
            w.wait()
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
        elif self._has_obs:
            out.append(self.obs_recv)  # (world * n, ...): the collective's receive buffer, already in global env order
        for o, nb, shape, dtype in self.meta:  # a few bytes per env: sliced out of the rank-major packed message
            out.append(per_rank[:, o:o + nb].contiguous().view(dtype).view(self.world * shape[0], *shape[1:]))
        return tuple(out)


def all_gather_step(tensors, group=None):
    """Blocking form: the tuple of per-shard tensors, concatenated along dim 0 in global env order on every rank, moved by
    a single packed collective."""
    g = StepGather(group, overlap=False)
    g.launch(tensors)
    return g.wait()
