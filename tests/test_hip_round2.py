"""GPU tests of what round 2 added behind the C ABI: action containment (base_pong_env.py:42), the float32 store
epilogue (SURVEY 8d config-3 variant), other resized_dim values against the oracle, device-side terminal
observations, the car-0-only done rule, BASELINE config #4 at its real size, and the info validity window."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")


def test_action_containment_host_and_device():
    """pong/base_pong_env.py:42 asserts action_space.contains(action).  Host arrays are checked on the host; device
    tensors by the step kernel: the bat of a bad action does not move, and the error surfaces at the next call."""
    _need_gpu()
    import competitive_rl_amd as crl
    from competitive_rl_amd._native import CrlActionError

    n = 64
    env = crl.HipPongVecEnv(n, seed=1, mode="raw")
    env.reset()
    with pytest.raises(AssertionError):
        env.step(np.full((n, 2), 5))
    with pytest.raises(AssertionError):
        env.step(np.full((n, 2), -1))
    ok = torch.ones((n, 2), dtype=torch.int32, device="cuda")
    env.step_device(ok)
    env.check()                                       # nothing pending
    st0 = env.get_state()
    bad = ok.clone()
    bad[7, 0] = 5
    env.step_device(bad)
    st1 = env.get_state()
    assert st1["bat_l_y"][7] == st0["bat_l_y"][7]    # the reference would not have stepped; here the bat stays put
    with pytest.raises(CrlActionError) as e:
        env.check()
    assert isinstance(e.value, AssertionError) and "5" in str(e.value)
    env.check()                                       # cleared
    env.step_device(bad)
    torch.cuda.synchronize()
    serial = env._serial
    with pytest.raises(CrlActionError):
        env.step_device(ok)                           # the next step reports the earlier one ...
    assert env._serial == serial                      # ... did no work, left the host mirror alone ...
    env.step_device(ok)                               # ... and the report was made once: this call proceeds
    env.step(np.ones((n, 2), np.int64))
    env.reset()
    env.close()


@pytest.mark.parametrize("R,K", [(84, 4), (42, 1), (64, 2)])
def test_float32_store_epilogue_equals_uint8(atlas, R, K):
    """obs_dtype="float32" (DummyVecEnv's buffer dtype) is produced by the raster's store epilogue, not by a torch
    pass: same values as the uint8 path -- and as the oracle -- on every step, terminal observations included."""
    _need_gpu()
    import competitive_rl_amd as crl
    from oracle import pong_oracle as po

    n, steps = 70, 130
    rs = np.random.RandomState(R + K)
    f = crl.HipPongVecEnv(n, seed=4, mode="wrapped", resized_dim=R, frame_stack=K, obs_dtype="float32")
    ora = po.PongOracle(n, atlas, obs_mode=po.GRAY, resized_dim=R, frame_stack=K, seed=4)
    of = torch.stack(f.reset(), 1)
    assert of.dtype == torch.float32 and f._obs[0].dtype == torch.float32
    assert np.array_equal(of.cpu().numpy(), ora.reset().astype(np.float32))
    seen = 0
    for t in range(steps):
        a = rs.randint(0, 3, (n, 2))
        obs, rew, done, infos = f.step(a)
        oo, orew, odone = ora.step(a)
        assert np.array_equal(torch.stack(obs, 1).cpu().numpy(), oo.astype(np.float32)), t
        assert np.array_equal(rew.cpu().numpy(), orew)
        for i in np.nonzero(odone)[0][:2]:
            term = infos[int(i)]["terminal_observation"]
            assert term[0].dtype == torch.float32
            assert np.array_equal(torch.stack(term).cpu().numpy()[:, 0], ora.terminal_observation(int(i)).astype(np.float32))
            seen += 1
    assert seen > 0
    f.close()


@pytest.mark.parametrize("R,K", [(84, 4), (42, 1), (84, 1), (42, 4)])
def test_float32_ref_equals_the_oracles_unrounded_observation(atlas, R, K):
    """obs_dtype="float32_ref" (include/crl.h CRL_OBS_F32_REF): the reference's own float32 values -- unrounded INTER_AREA
    averages of the float gray frame on step() (utils/atari_wrappers.py:104-116, 215-219), rounded on reset() and on the
    auto-reset of a finished env -- bit for bit against the oracle (same float32 operation order, no contraction), stack
    planes and terminal observations included."""
    _need_gpu()
    import competitive_rl_amd as crl
    from oracle import pong_oracle as po

    n, steps = 50, 400 if R == 42 else 160
    rs = np.random.RandomState(R + K)
    f = crl.HipPongVecEnv(n, seed=6, mode="wrapped", resized_dim=R, frame_stack=K, obs_dtype="float32_ref")
    ora = po.PongOracle(n, atlas, obs_mode=po.GRAY, resized_dim=R, frame_stack=K, seed=6, obs_dtype="float32_ref")
    of = torch.stack(f.reset(), 1)
    assert of.dtype == torch.float32
    assert np.array_equal(of.cpu().numpy(), ora.reset())
    fractional = seen = 0
    for t in range(steps):
        a = rs.randint(0, 3, (n, 2))
        obs, rew, done, infos = f.step(a)
        oo, orew, odone = ora.step(a)
        got = torch.stack(obs, 1).cpu().numpy()
        assert np.array_equal(got, oo), (t, float(np.abs(got - oo).max()), int((got != oo).sum()))
        assert np.array_equal(rew.cpu().numpy(), orew)
        fractional += int((got != np.rint(got)).sum())
        for i in np.nonzero(odone)[0][:2]:
            term = torch.stack(infos[int(i)]["terminal_observation"]).cpu().numpy()[:, 0]
            assert np.array_equal(term, ora.terminal_observation(int(i))), (t, i)
            assert np.array_equal(got[i, :, -1], np.rint(got[i, :, -1]))  # the auto-reset's observation: rounded again
            seen += 1
    assert fractional > 1000 and (seen > 0 or R != 42)
    assert po.f32ref_ambiguous() == 0  # (the kernel infers "reset observation" from identical kept frames; the oracle is told: never a disagreement)
    f.close()


@pytest.mark.parametrize("R", [64, 32, 48])
def test_other_resized_dims_match_oracle(atlas, R):
    """crl_create accepts any even resized_dim in 8..84; sizes other than the reference's 84 / 42 take the same
    kernels (up to five taps per axis) or the per-pixel evaluator: bit-exact against the oracle as well."""
    _need_gpu()
    import competitive_rl_amd as crl
    from oracle import pong_oracle as po

    n, steps, K = 48, 120, 4
    rs = np.random.RandomState(R)
    env = crl.HipPongVecEnv(n, seed=2, mode="wrapped", resized_dim=R, frame_stack=K)
    ora = po.PongOracle(n, atlas, obs_mode=po.GRAY, resized_dim=R, frame_stack=K, seed=2)
    assert np.array_equal(torch.stack(env.reset(), 1).cpu().numpy(), ora.reset())
    for t in range(steps):
        a = rs.randint(0, 3, (n, 2))
        obs, rew, done, _ = env.step(a)
        oo, orew, odone = ora.step(a)
        assert np.array_equal(torch.stack(obs, 1).cpu().numpy(), oo), (R, t)
        assert np.array_equal(rew.cpu().numpy(), orew) and np.array_equal(done[:, 0].cpu().numpy().astype(np.uint8), odone)
    env.close()


def test_terminal_observations_by_device_index_and_validity_window(atlas):
    """All finished envs of a step are drawn by one crl_terminal_observation_dev call from a device index list;
    reading them after the env has moved on raises instead of returning another episode's frames."""
    _need_gpu()
    import competitive_rl_amd as crl
    from oracle import pong_oracle as po

    n = 300
    env = crl.HipPongVecEnv(n, seed=6, mode="wrapped", resized_dim=42, frame_stack=1)
    ora = po.PongOracle(n, atlas, obs_mode=po.GRAY, resized_dim=42, frame_stack=1, seed=6)
    env.reset(), ora.reset()
    rs = np.random.RandomState(0)
    checked, stale = 0, None
    for t in range(160):
        a = rs.randint(0, 3, (n, 2))
        obs, rew, done, infos = env.step(a)
        _, _, odone = ora.step(a)
        idx = torch.nonzero(done[:, 0]).reshape(-1)
        if stale is not None and stale[2] == t - 1:
            with pytest.raises(RuntimeError):                           # the env has been stepped since
                stale[0][stale[1]]
        if idx.numel():
            if stale is None:
                stale = (infos, int(idx[0]), t)                         # untouched: read only after the next step
                continue
            got = env.terminal_observation(idx)                         # device index list
            host = env.terminal_observation(idx.cpu().numpy())           # host index list (copied to the device)
            for k, i in enumerate(idx.cpu().tolist()):
                want = ora.terminal_observation(i)
                assert np.array_equal(torch.stack(got[k]).cpu().numpy()[:, 0], want)
                assert torch.equal(torch.stack(got[k]), torch.stack(host[k]))
                assert torch.equal(torch.stack(infos[i]["terminal_observation"]), torch.stack(got[k]))
                checked += 1
    assert checked >= 3 and stale is not None
    with pytest.raises(IndexError):
        env.terminal_observation([n])
    env.close()


def test_car0_done_policy_and_infos():
    """make_competitive_car_racing's rule (d[0], make_competitive_car_racing.py:24-33): a finished OPPONENT does not end
    the episode -- it stays where it is, frozen, while car 0 drives on; the default rule (any) ends it."""
    _need_gpu()
    import competitive_rl_amd as crl

    n = 6
    a = crl.HipCarVecEnv(n, seed=3, done_policy="car0")
    b = crl.HipCarVecEnv(n, seed=3, done_policy="any")
    for e in (a, b):
        e.reset()
        st = e.get_state()
        for body in ("hull", "wheel"):
            st["car"][:, 1][body]["cx"][2] += 900.0   # env 2: car 1 leaves the playfield
        e.set_state(st)
    acts = np.zeros((n, 2, 2), np.float32)
    acts[:, :, 1] = 0.5
    for t in range(5):
        oa, ra, da, ia = a.step(acts)
        ob, rb, db, ib = b.step(acts)
        assert not bool(da.any()), t
        assert bool(db[2, 0]) == (t == 0) and int(db.sum()) == (1 if t == 0 else 0)
        dc, ns = (x.cpu().numpy() for x in a._info_snapshot())
        assert dc[2].tolist() == [0, 1] and dc[[0, 1, 3, 4, 5]].sum() == 0 and (ns == t + 1).all()
        assert ia[2][1]["reward"] == (float(np.float32(-0.1)) if t == 0 else 0.0)      # the frozen car earns nothing more
        assert ia[2][0]["num_steps"] == t + 1
    sa = a.get_state()
    assert sa["car"][2, 1]["done"] == 1 and sa["car"][2, 0]["done"] == 0 and sa["episode"][2] == 1
    a.close(), b.close()


def test_full_size_car_properties():
    """BASELINE config #4 at its real size (16 384 envs): determinism checksum, reward bounds, TimeLimit,
    observation invariants."""
    _need_gpu()
    import competitive_rl_amd as crl

    n, steps = 16384, 30

    def run():
        env = crl.HipCarVecEnv(n, seed=0)
        first = env.reset()
        g = torch.Generator(device="cuda").manual_seed(3)
        tot_done = 0
        rsum = torch.zeros((n, 2), device="cuda")
        for t in range(steps):
            a = torch.rand((n, 2, 2), generator=g, device="cuda") * 2 - 1
            buf, rew, done = env.step_device(a)
            assert bool(((rew >= -0.1001) & (rew < 20.0)).all()), t            # -0.1 per step, +1000/len(track) per new tile
            rsum += rew
            tot_done += int(done.sum())
        torch.cuda.synchronize()
        ck = int(buf.reshape(-1).view(torch.int32).sum(dtype=torch.int64))
        st = env.get_state()
        assert env.cap_hits() == (0, 0, 0, 0)   # no wheel touched more than 6 tiles, no pair of cars more than 8 manifolds
        env.close()
        return first.clone(), buf.clone(), ck, st, tot_done, rsum

    first, buf, ck, st, tot_done, rsum = run()
    palette = torch.tensor([0, 29, 44, 60, 76, 101, 103, 107, 149, 161, 176, 255], device="cuda", dtype=torch.uint8)
    assert bool(torch.isin(buf, palette).all()) and bool(torch.isin(first, palette).all())
    own = (buf[:, 0] == 60).sum(dim=(1, 2))                                     # every viewer sees its own red hull
    assert int(own.min()) > 0 and int((buf[:, 1] == 60).sum(dim=(1, 2)).min()) > 0
    assert bool((buf[:, :, 86:, :] != 161).all()) and bool((buf[:, :, 86:, :] != 176).all())   # indicator strip: no grass
    assert (st["elapsed"] <= steps).all() and ((st["elapsed"] == steps) | (st["episode"] > 1)).all()
    assert bool((rsum.abs() < 20.0 * steps).all()) and tot_done < n // 50
    first2, buf2, ck2, st2, tot_done2, _ = run()
    assert ck == ck2 and tot_done == tot_done2 and torch.equal(first, first2) and torch.equal(buf, buf2)
    for f in ("cx", "cy", "a", "vx", "vy", "w"):
        assert np.array_equal(st["car"]["hull"][f], st2["car"]["hull"][f])

    # TimeLimit at full size: park the clock at 999 steps -> every env ends on the next step and restarts
    env = crl.HipCarVecEnv(n, seed=0)
    env.reset()
    s = env.get_state()
    s["elapsed"] = 999
    env.set_state(s)
    _, _, done = env.step_device(torch.zeros((n, 2, 2), device="cuda"))
    assert bool(done.all())
    s = env.get_state()
    assert (s["elapsed"] == 0).all() and (s["episode"] == 2).all()
    # a longer drive at full size: cars spread over their tracks, hit each other, leave the playfield, get reset
    g = torch.Generator(device="cuda").manual_seed(5)
    for t in range(600):
        env.step_device(torch.rand((n, 2, 2), generator=g, device="cuda") * 2 - 1)
    assert env.cap_hits() == (0, 0, 0, 0)
    _, overflow = env.get_map(123)
    assert overflow == 0
    env.close()


def _two_gpu_worker(rank, world, port, out_dir, backend="nccl", mode="obs"):
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    import competitive_rl_amd as crl

    # "nccl": one GPU per rank, RCCL moves the packed message.  "gloo": both ranks on GPU 0 (a one-GPU box), the HIP
    # shards' outputs are staged to the host and the same packed collective runs over gloo.
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo")
    total, steps = 512, 60
    sh = crl.shard_of(total, world, rank)
    env = crl.HipPongVecEnv(sh.count, seed=21, mode="wrapped", resized_dim=42, frame_stack=1, env_id_base=sh.base, device=f"cuda:{dev}")
    env.reset()
    acts = torch.as_tensor(np.random.RandomState(5).randint(0, 3, (steps, total, 2)).astype(np.int32)).cuda()
    got = []
    if mode == "descriptors":
        # every rank ships 64 bytes per env and re-draws the GLOBAL batch locally (crl_obs_descriptors / crl_render_frames_dev)
        if backend == "nccl":
            g = crl.StepGather(overlap=True, mode="descriptors")
            for t in range(steps):
                _, rew, done = env.step_device(acts[t, sh.base:sh.base + sh.count].contiguous())
                g.launch((rew, done), env=env)
                rew.fill_(-7.0)          # (a later step rewriting the single reward buffer must not reach the message)
                got.append([x.cpu().numpy().copy() for x in g.wait()])
        else:  # gloo cannot move device memory: stage the descriptors through the host, re-draw on the GPU
            for t in range(steps):
                _, rew, done = env.step_device(acts[t, sh.base:sh.base + sh.count].contiguous())
                desc = env.obs_descriptors().cpu()
                parts = [torch.empty_like(desc) for _ in range(world)]
                dist.all_gather(parts, desc)
                r_all, d_all = crl.all_gather_step((rew.cpu(), done.cpu()))
                obs = torch.cat([env.render_descriptors(p.cuda()) for p in parts])
                got.append([obs.cpu().numpy(), r_all.numpy(), d_all.numpy()])
    else:
        g = crl.StepGather(overlap=True)
        for t in range(steps):
            a = acts[t, sh.base:sh.base + sh.count].contiguous()
            if backend == "nccl":
                slot = g.obs_slot(env._obs[0].shape, env._obs[0].dtype, env.device)   # the env draws into the send buffer
                out = env.step_device(a, obs_out=slot)
            else:
                out = tuple(x.cpu() for x in env.step_device(a))
            g.launch(out)                       # the observation's all-gather + the packed small one (on a side stream when they run on the GPU) ...
            res = g.wait()
            # what a consumer gets for the observation IS the collective's receive buffer, (world * n, ...) in global env order
            assert res[0].data_ptr() == g.obs_recv.data_ptr() and res[0].is_contiguous() and res[0].shape[0] == world * sh.count
            got.append([x.cpu().numpy().copy() for x in res])
    if rank == 0:
        np.savez(os.path.join(out_dir, "g.npz"), obs=np.stack([o[0] for o in got]), rew=np.stack([o[1] for o in got]),
                 done=np.stack([o[2] for o in got]))
    env.close()
    dist.barrier()
    dist.destroy_process_group()


def _check_against_one_batch(tmp_path):
    import competitive_rl_amd as crl

    got = np.load(tmp_path / "g.npz")
    total, steps = 512, 60
    env = crl.HipPongVecEnv(total, seed=21, mode="wrapped", resized_dim=42, frame_stack=1)
    env.reset()
    acts = torch.as_tensor(np.random.RandomState(5).randint(0, 3, (steps, total, 2)).astype(np.int32)).cuda()
    for t in range(steps):
        buf, rew, done = env.step_device(acts[t].contiguous())
        assert np.array_equal(got["obs"][t], buf.cpu().numpy()) and np.array_equal(got["rew"][t], rew.cpu().numpy())
        assert np.array_equal(got["done"][t], done.cpu().numpy()), t
    env.close()


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_hip_shards_on_one_gpu_plus_packed_gather_equal_one_batch(tmp_path):
    """HIP + sharding together on a ONE-GPU box: two processes, each a HIP shard on GPU 0 (RNG keyed by the global env id
    through env_id_base), their step outputs exchanged by the single packed all-gather (gloo) == one unsharded HIP batch."""
    _need_gpu()
    import torch.multiprocessing as mp

    mp.start_processes(_two_gpu_worker, args=(2, _free_port(), str(tmp_path), "gloo"), nprocs=2, join=True, start_method="spawn")
    _check_against_one_batch(tmp_path)


def test_two_hip_shards_exchanging_descriptors_equal_one_batch(tmp_path):
    """Config #5 without moving pixels: each shard ships its frame descriptors (64 bytes per env), every rank re-draws both
    shards' observations from them (crl_render_frames_dev) == one unsharded HIP batch.  Two processes on GPU 0 over gloo."""
    _need_gpu()
    import torch.multiprocessing as mp

    mp.start_processes(_two_gpu_worker, args=(2, _free_port(), str(tmp_path), "gloo", "descriptors"), nprocs=2, join=True, start_method="spawn")
    _check_against_one_batch(tmp_path)


def test_descriptors_redraw_the_observation_exactly():
    """crl_obs_descriptors + crl_render_frames_dev on one context: raw RGB, fused 84 x 84 4-stack (uint8 and float32), and
    draw-into-the-caller's-buffer (obs_out)."""
    _need_gpu()
    import competitive_rl_amd as crl

    n = 200
    rs = np.random.RandomState(3)
    for kw in (dict(mode="raw"), dict(mode="wrapped", resized_dim=84, frame_stack=4), dict(mode="wrapped", resized_dim=42, frame_stack=1),
               dict(mode="wrapped", resized_dim=84, frame_stack=4, obs_dtype="float32")):
        env = crl.HipPongVecEnv(n, seed=2, **kw)
        env.reset()
        mine = torch.empty_like(env._obs[0])
        for t in range(40):
            a = torch.as_tensor(rs.randint(0, 3, (n, 2)).astype(np.int32)).cuda()
            if t % 2:
                buf, _, _ = env.step_device(a, obs_out=mine)
                assert buf.data_ptr() == mine.data_ptr()
            else:
                buf, _, _ = env.step_device(a)
            again = env.render_descriptors(env.obs_descriptors())
            assert torch.equal(again, buf), (kw, t)
        env.close()


def test_two_hip_shards_plus_packed_gather_equal_one_batch(tmp_path):
    """BASELINE config #5 in small: two HIP shards on two GPUs + the single packed RCCL all-gather == one unsharded HIP batch.
    Needs two visible GPUs (the driver's multi-GPU node); skipped on a one-GPU box."""
    _need_gpu()
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import torch.multiprocessing as mp

    for mode in ("obs", "descriptors"):
        mp.start_processes(_two_gpu_worker, args=(2, _free_port(), str(tmp_path), "nccl", mode), nprocs=2, join=True, start_method="spawn")
        _check_against_one_batch(tmp_path)


def test_one_rank_rccl_gather_equals_one_batch(tmp_path):
    """What a ONE-GPU box can say about the RCCL path: a world of one rank over the "nccl" backend runs the
    same StepGather code as config #5 -- the env drawing straight into the send buffer (obs_slot), the pack on the caller's stream, the
    collective on the side stream behind an event, the wait, the reuse of both buffers step after step, and the descriptor mode's
    re-draw -- with RCCL itself moving the bytes; the result must be the unsharded batch.  (Two ranks on one GPU are refused by
    RCCL: the two-rank variants above run over gloo, the two-GPU ones below are skipped here.)"""
    _need_gpu()
    import torch.multiprocessing as mp

    for mode in ("obs", "descriptors"):
        mp.start_processes(_two_gpu_worker, args=(1, _free_port(), str(tmp_path), "nccl", mode), nprocs=1, join=True, start_method="spawn")
        _check_against_one_batch(tmp_path)


def test_bench_line_with_the_gather_on_one_rank():
    """``bench.py --gather obs`` on one GPU (no process group: the gather is a no-op by design) still names the rank's env range, and the car
    workload accepts the flag (`car` leg of --gather obs)."""
    _need_gpu()
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for wl, envs in (("fused84", 2048), ("car", 512)):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--workload", wl, "--envs", str(envs),
                              "--gather", "obs", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, CRL_BENCH_CAR_PREROLL="50"))
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["n_gpus"] == 1 and line["comm"]["world"] == 1 and line["comm"]["env_ranges"] == [[0, envs]]


def test_bench_under_torchrun_with_one_rank_runs_the_distributed_path():
    """The driver's N > 1 launch line with N = 1 (``python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 ...
    bench.py --gpus 1``) and CRL_BENCH_FORCE_DIST=1: the rank reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, joins an RCCL process group of
    one, passes the barriers, the max over ranks and the --gather collective (obs: drawn into the send buffer; descriptors: re-drawn after
    the collective) -- every line of the multi-GPU path a one-GPU box can execute."""
    _need_gpu()
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for wl, envs, gather in (("fused84", 4096, "obs"), ("fused84", 4096, "descriptors"), ("car", 512, "obs"), ("raw", 1024, "scalars")):
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                              "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                              "--workload", wl, "--envs", str(envs), "--gather", gather, "--no-cpu-baseline"], capture_output=True, text=True,
                             timeout=600, env=dict(os.environ, CRL_BENCH_CAR_PREROLL="50", CRL_BENCH_FORCE_DIST="1"))
        assert out.returncode == 0, (wl, gather, out.stderr[-2000:])
        line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["n_gpus"] == 1 and line["steps"] == 5 and line["config"]["gather"] == gather
        assert line["comm"]["backend"].startswith("nccl") and line["comm"]["world"] == 1 and line["comm"]["env_ranges"] == [[0, envs]]
        assert abs(line["value"] - envs * 5 / (line["ms_per_step"] * 5e-3)) <= 1e-3 * line["value"]


def test_bench_line_on_two_gpus_names_the_ranks_and_their_env_ranges():
    """The first multi-GPU box validates ranks + ranges in one run: ``bench.py --gpus 2 --steps 5`` for the Pong and the
    CarRacing workload with the config-#5 gather -- world 2, RCCL named, contiguous env ranges by global id, whole-job value.
    Needs two visible GPUs; skipped on a one-GPU box."""
    _need_gpu()
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for wl, envs in (("fused84", 4096), ("car", 1024)):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--workload", wl,
                              "--envs", str(envs), "--gather", "obs", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, CRL_BENCH_CAR_PREROLL="50"))
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 5
        comm = line["comm"]
        assert comm["world"] == 2 and comm["backend"].startswith("nccl") and comm["gather"] == "obs"
        assert comm["env_ranges"] == [[0, envs], [envs, 2 * envs]]
        assert abs(line["value"] - 2 * envs * 5 / (line["ms_per_step"] * 5e-3)) <= 1e-3 * line["value"]  # whole-job: both ranks' envs


def test_address_linear_gray_writer_is_bit_exact_too():
    """The address-linear fused-84 writer (header kernel + aligned 1-KiB-block sweep; bit-exact and slower,
    so it lives in the profiling variant only: CRL_LIB_VARIANT=abl CRL_GRAY_SWEEP=1) against the oracle, in a child process
    because library and switch are chosen once per process."""
    _need_gpu()
    import os
    import subprocess
    import sys

    from competitive_rl_amd.build import PKG
    if not os.path.exists(os.path.join(PKG, "libcrl_hip_abl.so")):
        pytest.skip("libcrl_hip_abl.so not built (python -m competitive_rl_amd.build --variant abl -DCRL_ABLATION)")

    code = r"""
import numpy as np, torch, sys
sys.path.insert(0, %r)
import competitive_rl_amd as crl
from competitive_rl_amd import _native
from oracle import pong_oracle as po
atlas = _native.load_score_atlas()
for K, n, dt in ((4, 130, "uint8"), (1, 67, "uint8"), (4, 70, "float32")):   # (float32: round 6, the same blocks widened on the way out)
    env = crl.HipPongVecEnv(n, seed=5, mode="wrapped", resized_dim=84, frame_stack=K, obs_dtype=dt)
    ora = po.PongOracle(n, atlas, obs_mode=po.GRAY, resized_dim=84, frame_stack=K, seed=5)
    assert np.array_equal(torch.stack(env.reset(), 1).cpu().numpy(), ora.reset().astype(dt))
    rs = np.random.RandomState(K)
    for t in range(150):
        a = rs.randint(0, 3, (n, 2))
        obs, rew, done, _ = env.step(a)
        oo, orew, odone = ora.step(a)
        assert np.array_equal(torch.stack(obs, 1).cpu().numpy(), oo.astype(dt)), (K, t, dt)
    env.close()
print("sweep ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CRL_GRAY_SWEEP="1", CRL_LIB_VARIANT="abl"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sweep ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_wheel_sensor_overflow_path_gives_the_same_results():
    """car_sensor_kernel sends a car whose per-wheel tile lists overflow through car_sensor_serial_kernel (one lane walks the whole
    track); CRL_CAR_SENSOR_SERIAL=1 forces every car that way.  The step-bookkeeping fixture recorded from the reference, the
    teacher-forced oracle comparison, action repeat and the free-running comparison must pass unchanged (child process: the
    switch is read once)."""
    _need_gpu()
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "tests/test_hip_car_step_golden.py",
                        "tests/test_hip_car_parity.py", "-k",
                        "bookkeeping or teacher_forced or action_repeat or free_running or single_car"],
                       cwd=root, env=dict(os.environ, CRL_CAR_SENSOR_SERIAL="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_small_class_launches_loop_when_the_class_is_larger_than_expected():
    """The frames of a step's finished / coupled envs are drawn by launches sized from the PREVIOUS step's class counts; when a
    class is suddenly much larger (here: 300 of 512 envs finish in one step after steps with none) the launch loops over its
    list.  Every frame the step returns -- terminal observation, first frame of the new episode, the untouched envs -- must be
    what one plain render of the same state gives."""
    _need_gpu()
    import competitive_rl_amd as crl

    n = 512
    env = crl.HipCarVecEnv(n, seed=11)
    env.reset()
    acts = torch.zeros((n, 2, 2), device="cuda")
    acts[:, :, 1] = 0.3
    for _ in range(3):
        env.step_device(acts)                      # steps without a finished env: the expected class sizes settle near zero
    st = env.get_state()
    out = np.arange(n) % 5 < 3                     # 308 envs: car 0 is put far outside the playfield
    for body in ("hull", "wheel"):
        st["car"][:, 0][body]["cx"][out] += 900.0
    env.set_state(st)
    before = env.render_current().clone()          # what the terminal observation of the finishing envs must show ... after one more step
    obs, rew, done = env.step_device(acts)
    obs = obs.clone()
    assert np.array_equal(done.cpu().numpy().astype(bool), out)
    assert torch.equal(obs, env.render_current())  # finished envs: first frame of the new episode; the others: their new state
    idx = torch.nonzero(done).reshape(-1)
    term = torch.stack(env.terminal_observation(idx))
    assert term.shape[0] == int(out.sum()) and not torch.equal(term, obs[idx])
    # a car 900 units off the track sees grass only, apart from the indicator strip: the terminal frames are drawn, not blank
    assert int((term[:, 0, :80] == 161).sum() + (term[:, 0, :80] == 176).sum()) > 0.9 * term[:, 0, :80].numel()
    del before
    env.close()


def test_car_stream_order_switch_is_robust():
    """CRL_CAR_STREAM_ORDER only deals the context's hardware queues differently: any string -- letters left out,
    repeated, unknown -- must still give a working context with identical results (child process: read at create)."""
    _need_gpu()
    import os
    import subprocess
    import sys

    code = r"""
import sys, torch
sys.path.insert(0, %r)
import competitive_rl_amd as crl
n = 96
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.rand((40, n, 2, 2), generator=g, device="cuda") * 2 - 1
env = crl.HipCarVecEnv(n, seed=9)
env.reset()
tot = torch.zeros((), dtype=torch.int64, device="cuda")
for t in range(40):
    o, r, d = env.step_device(acts[t])
    tot += o.to(torch.int64).sum() + (r * 1000).to(torch.int64).sum()
print("checksum", int(tot))
env.close()
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sums = set()
    for order in ("s2oDg", "g", "xyz", "ddddHHs2ogDD", ""):
        env = dict(os.environ, CRL_CAR_STREAM_ORDER=order) if order is not None else dict(os.environ)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "checksum" in r.stdout, (order, r.stdout[-500:], r.stderr[-1500:])
        sums.add(r.stdout.strip().splitlines()[-1])
    assert len(sums) == 1, sums


def test_car_walk_ahead_pieces_give_the_same_tracks():
    """The walk-ahead advances every pending track walk by a bounded number of iterations per launch and resumes it from its saved
    state (csrc/car_track.hip WalkSave).  Whatever the piece size -- 7 iterations (hundreds of resumes per walk, mid-attempt and
    across failed attempts), the default, or unbounded -- and however far the walks have got when an env is reset (a reset
    that comes too early walks inline), the envs must see the same tracks: episodes of 60 steps, five resets per env (child
    processes of the profiling build: the piece size is read once)."""
    _need_gpu()
    import os
    import subprocess
    import sys

    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import competitive_rl_amd as crl
n = 48
g = torch.Generator(device="cuda").manual_seed(2)
acts = torch.rand((16, n, 2, 2), generator=g, device="cuda") * 2 - 1
env = crl.HipCarVecEnv(n, seed=5)
env.reset()
st = env.get_state()
st["elapsed"] = (1000 - 60 + np.arange(n) %% 60).astype(st["elapsed"].dtype)   # every env ends within 60 steps, then every 1000
env.set_state(st)
tot = torch.zeros((), dtype=torch.int64, device="cuda")
for rnd in range(5):
    for t in range(70):
        o, r, d = env.step_device(acts[t %% 16])
        tot += o.to(torch.int64).sum() + d.to(torch.int64).sum() * 7
    st = env.get_state()
    st["elapsed"] = (1000 - 60 + np.arange(n) %% 60).astype(st["elapsed"].dtype)
    env.set_state(st)
trk = sum(float(np.asarray(env.get_track(i)["tile_poly"], np.float64).sum()) for i in range(0, n, 5))
print("checksum", int(tot), repr(trk), int(env.get_state()["episode"].astype("int64").sum()))
env.close()
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = set()
    for budget in ("7", "160", "0"):
        env = dict(os.environ, CRL_LIB_VARIANT="abl", CRL_CAR_WALK_BUDGET=budget)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "checksum" in r.stdout, (budget, r.stdout[-500:], r.stderr[-1500:])
        outs.add(r.stdout.strip().splitlines()[-1])
    assert len(outs) == 1, outs


def test_car_full_resets_with_the_walk_ahead_running_give_the_same_tracks():
    """A full reset walks every env's track itself on the caller's stream, into the scratch the walk-ahead pieces use; the piece
    queued behind a reset must not start another attempt of the same walk beside it (it would overwrite points of the lap the
    reset is about to read: seen once in ~50 runs of the sharding test while the pieces were not ordered behind the reset).
    Repeated resets with steps in between, against a context without the pipeline (no walk-ahead at all).  (The race needs the
    piece to start while the reset's walk is in a later attempt: this test is a consistency check, not a reliable detector.)"""
    _need_gpu()
    import os

    import competitive_rl_amd as crl

    n = 1024
    a = crl.HipCarVecEnv(n, seed=21)
    os.environ["CRL_CAR_NO_OVERLAP"] = "1"
    try:
        b = crl.HipCarVecEnv(n, seed=21)
    finally:
        del os.environ["CRL_CAR_NO_OVERLAP"]
    g = torch.Generator(device="cuda").manual_seed(4)
    acts = torch.rand((3, n, 2, 2), generator=g, device="cuda") * 2 - 1
    for rnd in range(10):
        oa, ob = a.reset(), b.reset()
        assert torch.equal(oa, ob), rnd
        for t in range(3 if rnd % 2 else 0):  # (some resets directly behind one another: the piece queued by the last one is still pending)
            xa, xb = a.step_device(acts[t]), b.step_device(acts[t])
            assert torch.equal(xa[0], xb[0]) and torch.equal(xa[1], xb[1]), (rnd, t)
        for i in range(rnd, n, 41):
            ta, tb = a.get_track(i), b.get_track(i)
            assert ta["n"] == tb["n"] and np.array_equal(ta["tile_poly"], tb["tile_poly"]), (rnd, i)
    a.close()
    b.close()


def test_default_bench_line_fits_the_drivers_buffer():
    """``python bench.py`` (every workload, CPU baselines included) prints ONE JSON line, and the driver keeps 8 KB of stdout: the line
    must stay well under that whatever digits the numbers have (small batches here: the line's structure does not depend on the size)."""
    _need_gpu()
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--envs", "2048"], capture_output=True, text=True,
                         timeout=900, env=dict(os.environ, CRL_BENCH_CAR_PREROLL="20", CRL_BENCH_CPU_BUDGET_S="1.0"))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:3]
    assert len(lines[0]) < 6800, len(lines[0])
    line = json.loads(lines[0])
    assert set(line["configs_brief"]) == {"raw", "fused84", "fused84_newest", "fused84_f32", "fused84_f32_ref", "car", "car_fma", "tournament",
                                          "tournament_full", "protocol"}
    assert line["roofline"] and line["cpu_baseline"]["cores"] >= 1 and "cpu_quota" in line["cpu_baseline"]
    car = line["configs"]["car"]["roofline"]
    assert car["step_frac"] > 0 and car["critical_path"]["kernel"] == "car_touch_kernel" and car["critical_path"]["max_us"] >= car["critical_path"]["mean_us"] > 0
    legs = line["configs"]["protocol"]["legs_ms_per_step"]
    assert {"step_envs", "step_envs_u8_stack", "step_envs_unbound"} <= set(legs) and line["configs"]["protocol"]["fused_updates"]["step_envs"] == 4
