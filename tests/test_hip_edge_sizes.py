"""Batch sizes at the edges of the kernels' lane / workgroup mappings -- 1, 2, 3 envs, one short of and one past a wavefront, one past a
workgroup of four wavefronts -- against the oracle at tolerance 0, both env families (empty / ragged inputs; the reference's
own smoke scripts run 1, 3 and 5 envs: test/test_pong.py, make_envs.py:121-170, test/test_car_racing.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 63, 65, 257]


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")


@pytest.fixture(scope="module")
def atlas():
    from competitive_rl_amd import _native as N

    return N.load_score_atlas()


@pytest.mark.parametrize("n", SIZES)
def test_pong_batches_of_every_awkward_size_match_the_oracle(atlas, n):
    _need_gpu()
    import competitive_rl_amd as crl
    from oracle import pong_oracle as po

    steps = 60 if n > 100 else 120
    for kw, okw in ((dict(mode="raw"), dict(obs_mode=po.RAW)),
                    (dict(mode="wrapped", resized_dim=84, frame_stack=4), dict(obs_mode=po.GRAY, resized_dim=84, frame_stack=4)),
                    (dict(mode="wrapped", resized_dim=42, frame_stack=1, obs_dtype="float32_ref"),
                     dict(obs_mode=po.GRAY, resized_dim=42, frame_stack=1, obs_dtype="float32_ref"))):
        rs = np.random.RandomState(n)
        f = crl.HipPongVecEnv(n, seed=9, **kw)
        ora = po.PongOracle(n, atlas, seed=9, **okw)
        got = torch.stack(f.reset(), 1).cpu().numpy()
        assert np.array_equal(got, ora.reset()), (n, kw)
        for t in range(steps):
            a = rs.randint(0, 3, (n, 2))
            obs, rew, done, _ = f.step(a)
            oo, orew, odone = ora.step(a)
            assert np.array_equal(torch.stack(obs, 1).cpu().numpy(), oo), (n, kw, t)
            d = np.asarray(done.cpu() if isinstance(done, torch.Tensor) else done).astype(bool)
            d = d.reshape(n, -1)[:, 0]  # (DummyVecEnv convention: the flag once per agent)
            assert np.array_equal(rew.cpu().numpy(), orew) and np.array_equal(d, np.asarray(odone).astype(bool).reshape(n, -1)[:, 0]), (n, kw, t)
        f.close()


@pytest.mark.parametrize("solver", ["box2d", "fma"])
@pytest.mark.parametrize("n", SIZES)
def test_car_batches_of_every_awkward_size_match_the_oracle(n, solver):
    """Free-running (no re-sync) from the oracle's tracks and states: bodies bit for bit, frames pixel for pixel."""
    _need_gpu()
    import competitive_rl_amd as crl
    from competitive_rl_amd import _native as N
    from oracle import car_oracle as co
    from tests.car_scenarios import make_oracle_envs
    from tests.test_hip_car_parity import ORACLE_OF, oracle_to_hip_state, push_tracks

    co.set_text(N.load_car_text())
    steps = 20 if n > 100 else 40
    envs = make_oracle_envs(n, seed0=3 + n, libm=ORACLE_OF[solver])
    hip = crl.HipCarVecEnv(n, solver=solver)
    hip.reset()
    push_tracks(hip, envs)
    hip.set_state(oracle_to_hip_state(envs))
    rs = np.random.RandomState(n)
    for t in range(steps):
        acts = rs.uniform(-1, 1, (n, 2, 2)).astype(np.float32)
        acts[:, :, 1] = np.abs(acts[:, :, 1])
        obs, rew, done = hip.step_device(torch.as_tensor(acts).cuda())
        for i, e in enumerate(envs):
            e.step(acts[i].astype(np.float64))
    hs = hip.get_state()
    got = obs.cpu().numpy()
    check = range(n) if n <= 65 else list(range(0, n, 16)) + [n - 1]
    for i in check:
        e = envs[i]
        for c in range(2):
            for f in ("cx", "cy", "a", "vx", "vy", "w"):
                assert hs[i]["car"][c]["hull"][f] == e.e["car"][c]["hull"][f], (n, i, c, f)
                assert np.array_equal(hs[i]["car"][c]["wheel"][f], e.e["car"][c]["wheel"][f]), (n, i, c, f)
        for v in range(2):
            assert np.array_equal(got[i, v], e.render(v)), (n, i, v)
    co.set_text(None)
    hip.close()


@pytest.mark.parametrize("R,n", [(84, 32768 + 3), (42, 32768 + 1), (84, 65536 + 2)])
def test_four_envs_per_wavefront_launch_at_ragged_sizes(R, n):
    """Round 6: from 32 768 envs on, the K = 1 launch (one plane per agent: make_envs("cPongDouble-v0")'s own observation) runs FOUR
    consecutive envs per wavefront -- here with env counts that are not multiples of four, so the last wavefront has one to three envs.
    The tail and a sample of the batch against SMALL contexts that hold the same global envs (env_id_base: the serve sampler is keyed by
    the global id, so they play the same games) and go through the one-env-per-wavefront launch; the head of the batch against the oracle."""
    _need_gpu()
    import competitive_rl_amd as crl
    from competitive_rl_amd import _native as N
    from oracle import pong_oracle as po

    steps = 24
    big = crl.HipPongVecEnv(n, seed=5, mode="wrapped", resized_dim=R, frame_stack=1)
    spans = [(0, 6), (n // 2 - 3, 8), (n - 9, 9)]   # (global id of the first env, count): head, middle, tail incl. the ragged wavefront
    small = [crl.HipPongVecEnv(c, seed=5, mode="wrapped", resized_dim=R, frame_stack=1, env_id_base=b) for b, c in spans]
    ora = po.PongOracle(spans[0][1], N.load_score_atlas(), obs_mode=po.GRAY, resized_dim=R, frame_stack=1, seed=5)
    got = torch.stack(big.reset(), 1)
    assert np.array_equal(got[:spans[0][1]].cpu().numpy(), ora.reset())
    for (b, c), s in zip(spans, small):
        assert torch.equal(got[b:b + c], torch.stack(s.reset(), 1)), ("reset", b)
    g = torch.Generator(device="cuda").manual_seed(3)
    for t in range(steps):
        a = torch.randint(0, 3, (n, 2), generator=g, device="cuda", dtype=torch.int32)
        buf, rew, done = big.step_device(a)
        for (b, c), s in zip(spans, small):
            sb, sr, sd = s.step_device(a[b:b + c].contiguous())
            assert torch.equal(buf[b:b + c], sb) and torch.equal(rew[b:b + c], sr) and torch.equal(done[b:b + c], sd), (t, b)
        oo, orew, odone = ora.step(a[:spans[0][1]].cpu().numpy())
        assert np.array_equal(buf[:spans[0][1]].cpu().numpy(), oo) and np.array_equal(rew[:spans[0][1]].cpu().numpy(), orew), t
    # nothing was written past the tensor: the launch's last wavefront stops at env n - 1 (a guard plane behind the buffer stays as it was)
    guard = torch.full((n + 4, 2, 1, R, R), 77, dtype=torch.uint8, device="cuda")
    view = guard[:n]
    big.step_device(torch.zeros((n, 2), dtype=torch.int32, device="cuda"), obs_out=view)
    assert bool((guard[n:] == 77).all())
    big.close()
    for s in small:
        s.close()
    ora.close()
