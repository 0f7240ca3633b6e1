"""N>1 path on CPU: world_size-2 gloo.  Each rank steps ITS shard (here with the CPU oracle
standing in for the per-GPU kernels -- the sharding / gather layer is backend-agnostic) and
the all-gather must reproduce one unsharded batch bit for bit: RNG is keyed by global env id."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, steps, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from competitive_rl_amd import _native
    from competitive_rl_amd.sharding import all_gather_step, shard_of
    from oracle import pong_oracle as po

    atlas = _native.load_score_atlas()
    sh = shard_of(total, world, rank)
    env = po.PongOracle(sh.count, atlas, obs_mode=po.GRAY, resized_dim=42, frame_stack=1, seed=21, env_id_base=sh.base)
    acts = np.random.RandomState(5).randint(0, 3, (steps, total, 2)).astype(np.int32)
    env.reset()
    outs = []
    for t in range(steps):
        obs, rew, done = env.step(acts[t, sh.base:sh.base + sh.count])
        g = all_gather_step((torch.from_numpy(obs.copy()), torch.from_numpy(rew.copy()), torch.from_numpy(done.copy())))
        outs.append([x.numpy().copy() for x in g])
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"), obs=np.stack([o[0] for o in outs]),
                 rew=np.stack([o[1] for o in outs]), done=np.stack([o[2] for o in outs]))
    dist.barrier()
    dist.destroy_process_group()


def _race_worker(rank, world, port, out_dir):
    """rewards / dones of step t must be what the gather of step t returns even when the caller rewrites its
    (single) reward and done buffers right after launch()."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from competitive_rl_amd.sharding import StepGather

    g = StepGather(overlap=True)
    n = 64
    rew, done = torch.zeros((n, 2)), torch.zeros((n,), dtype=torch.uint8)
    obs = torch.zeros((n, 2, 1, 8, 8), dtype=torch.uint8)
    ok = True
    for t in range(50):
        slot = g.obs_slot(obs.shape, obs.dtype, "cpu")          # the "env" draws straight into the send buffer
        slot.fill_((t + rank) % 251)
        rew.fill_(float(t + 10 * rank))
        done.fill_((t + rank) % 2)
        g.launch((slot, rew, done))
        rew.fill_(-1.0)                                          # the next step rewrites the single reward / done buffers at once
        done.fill_(9)
        o, r, d = g.wait()
        # the gathered observation IS the collective's receive buffer (global env order by construction), not a copy of it
        ok &= o.data_ptr() == g.obs_recv.data_ptr() and tuple(o.shape) == (world * n, 2, 1, 8, 8) and o.is_contiguous()
        ok &= o[rank * n:(rank + 1) * n].data_ptr() != slot.data_ptr()   # (send and receive sides are different buffers: step t+1 draws into the other slot)
        for k in range(world):
            ok &= bool((o[k * n:(k + 1) * n] == (t + k) % 251).all()) and bool((r[k * n:(k + 1) * n] == t + 10 * k).all())
            ok &= bool((d[k * n:(k + 1) * n] == (t + k) % 2).all())
    # launch(t+1) before wait(t) -- in the non-aliased modes there is ONE send buffer, and a size change replaces
    # both buffers: the second launch must first order itself behind the pending collective; its own result is what wait() returns
    g2 = StepGather(overlap=True)
    for t in range(20):
        a = torch.full((n, 3), float(t + rank))
        g2.launch((a, done))
        b = torch.full((n + t, 3), float(100 + t + rank))      # (different size: the buffers are reallocated)
        g2.launch((b,))
        (rb,) = g2.wait()
        for k in range(world):
            ok &= bool((rb[k * (n + t):(k + 1) * (n + t)] == 100 + t + k).all())
        ok &= g2.wait() is None
    with open(os.path.join(out_dir, f"race{rank}.txt"), "w") as f:
        f.write("ok" if ok else "bad")
    dist.barrier()
    dist.destroy_process_group()


def test_gather_is_unaffected_by_buffers_rewritten_after_launch(tmp_path):
    world = 2
    mp.start_processes(_race_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    assert all(open(tmp_path / f"race{r}.txt").read() == "ok" for r in range(world))


def test_shard_of_partitions_exactly():
    from competitive_rl_amd.sharding import shard_of

    for total, world in [(65536, 8), (524288, 8), (10, 3), (7, 7), (5, 8)]:
        spans = [shard_of(total, world, r) for r in range(world)]
        assert spans[0].base == 0 and sum(s.count for s in spans) == total
        for a, b in zip(spans, spans[1:]):
            assert a.base + a.count == b.base
    with pytest.raises(ValueError):
        shard_of(8, 2, 2)


def test_two_rank_gather_equals_unsharded(tmp_path, atlas):
    from oracle import pong_oracle as po

    total, steps, world = 16, 150, 2
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, total, steps, str(tmp_path)), nprocs=world, join=True,
                       start_method="spawn")
    got = np.load(tmp_path / "gathered.npz")
    env = po.PongOracle(total, atlas, obs_mode=po.GRAY, resized_dim=42, frame_stack=1, seed=21)
    acts = np.random.RandomState(5).randint(0, 3, (steps, total, 2)).astype(np.int32)
    env.reset()
    for t in range(steps):
        obs, rew, done = env.step(acts[t])
        assert np.array_equal(got["obs"][t], obs), t
        assert np.array_equal(got["rew"][t], rew) and np.array_equal(got["done"][t], done), t
    assert got["rew"].any()
