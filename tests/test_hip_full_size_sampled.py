"""Oracle parity AT THE BASELINE SIZES.  The other parity tests compare every env of a small batch (<= 192 raw,
96 wrapped, 2 048 CarRacing envs); the BASELINE configurations run 65 536 / 16 384 envs, where the raw observation tensor is
13.2 GB (byte offsets beyond 2^32 and 2^33) and the CarRacing maps 24 GB.  Here configs #2, #3 and #4 run at full size for >= 40
steps and 512 SAMPLED global env ids -- 0, N - 1, the ids whose data straddles a 2^32-byte boundary of the big tensors, the rest
random -- are stepped by oracle replicas (``env_id_base`` = the id; the RNG is keyed by the global id) on the same action rows:
frames, rewards, dones and state with tolerance 0."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

RUN = 8  # sampled ids come in runs of 8 consecutive envs (one oracle instance per run)


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")


def _sample_runs(n, bytes_per_env, count=64, seed=0):
    """`count` run starts: 0, n - RUN, the runs that contain a 2^32 k byte boundary of an (n, bytes_per_env) tensor, random others"""
    starts = {0, n - RUN}
    k = 1
    while (k << 32) < n * bytes_per_env:
        e = (k << 32) // bytes_per_env  # the env whose bytes straddle the boundary
        starts.add(min(max(e - RUN // 2, 0), n - RUN))
        k += 1
    rs = np.random.RandomState(seed)
    while len(starts) < count:
        s = int(rs.randint(0, n - RUN))
        if all(abs(s - q) >= RUN for q in starts):
            starts.add(s)
    return sorted(starts)


def _pong_full_size(atlas, mode, steps, n=65536, **kw):
    import competitive_rl_amd as crl
    from oracle import pong_oracle as po
    from tests.test_hip_pong_parity import assert_state_equal

    wrapped = mode == "wrapped"
    bytes_per_env = 2 * kw["frame_stack"] * kw["resized_dim"] ** 2 if wrapped else 2 * 210 * 160 * 3
    starts = _sample_runs(n, bytes_per_env)
    ids = np.concatenate([np.arange(s, s + RUN) for s in starts])
    ids_dev = torch.as_tensor(ids, device="cuda")
    boundary = [s for s in starts if any(s * bytes_per_env < (k << 32) <= (s + RUN) * bytes_per_env for k in range(1, 4))]
    assert mode != "raw" or n != 65536 or len(boundary) == 3, boundary  # 13.2 GB: the 2^32, 2^33 and 3 x 2^32 boundaries are sampled
    env = crl.HipPongVecEnv(n, seed=7, mode=mode, **kw)
    oras = [po.PongOracle(RUN, atlas, obs_mode=po.GRAY if wrapped else po.RAW, seed=7, env_id_base=s,
                          **({"resized_dim": kw["resized_dim"], "frame_stack": kw["frame_stack"]} if wrapped else {})) for s in starts]

    def sampled(buf):  # (N, 2, ...) device tensor -> the sampled envs on the host
        return buf[ids_dev].cpu().numpy()

    first = torch.stack([v[ids_dev] for v in env.reset()], 1).cpu().numpy()
    want = np.concatenate([o.reset() for o in oras])
    assert np.array_equal(first, want), "reset observations"
    g = torch.Generator(device="cuda").manual_seed(3)
    dones = 0
    for t in range(steps):
        a = torch.randint(0, 3, (n, 2), generator=g, device="cuda", dtype=torch.int32)
        a[torch.rand((n, 2), generator=g, device="cuda") < 0.1] = 999  # the cheat action (auto_action)
        buf, rew, done = env.step_device(a)
        rows = a[ids_dev].cpu().numpy()
        render = wrapped or t % 8 == 7 or t == steps - 1
        outs = [o.step(rows[j * RUN:(j + 1) * RUN], **({} if wrapped else {"render": render})) for j, o in enumerate(oras)]
        assert np.array_equal(rew[ids_dev].cpu().numpy(), np.concatenate([o[1] for o in outs])), t
        od = np.concatenate([o[2] for o in outs])
        assert np.array_equal(done[ids_dev].cpu().numpy().astype(np.uint8), od), t
        dones += int(od.sum())
        if render:
            got, exp = sampled(buf), np.concatenate([o[0] for o in outs])
            bad = np.nonzero((got != exp).reshape(len(ids), -1).any(1))[0]
            assert bad.size == 0, (t, "global env ids", ids[bad][:8])
    hs = env.get_state()[ids]
    os_ = np.concatenate([o.state for o in oras])
    assert_state_equal(hs, os_, ctx="final state of the sampled envs")
    if wrapped:
        assert np.array_equal(hs["keep"], os_["keep"]) and np.array_equal(hs["hist"], os_["hist"])
    env.close()
    for o in oras:
        o.close()
    return dones


def test_config2_raw_65536_envs_sampled_against_the_oracle(atlas):
    """BASELINE config #2: 65 536 envs, raw (N, 2, 210, 160, 3) u8 = 13.2 GB"""
    _need_gpu()
    _pong_full_size(atlas, "raw", 48)


def test_config3_fused84_65536_envs_sampled_against_the_oracle(atlas):
    """BASELINE config #3: 65 536 envs, fused gray + 84 x 84 + 4-stack; 48 steps = 192 frames"""
    _need_gpu()
    _pong_full_size(atlas, "wrapped", 48, resized_dim=84, frame_stack=4)


def test_config5_global_batch_524288_envs_on_one_device_sampled_against_the_oracle(atlas):
    """BASELINE config #5's node-wide batch (8 x 65 536 envs) as ONE context: 29.6 GB of fused observations, byte offsets up to 2^34.8
    (six 2^32 boundaries sampled), env ids up to 2^19 -- the index arithmetic of every Pong kernel one more power of eight out."""
    _need_gpu()
    _pong_full_size(atlas, "wrapped", 24, n=524288, resized_dim=84, frame_stack=4)


@pytest.mark.parametrize("solver", ["box2d", "fma"])
def test_config4_car_16384_envs_sampled_against_the_oracle(solver):
    """BASELINE config #4: 16 384 cCarRacingDouble envs through the production (pipelined) step, free running from reset() for 48 steps;
    512 sampled envs -- among them the ones whose 739 328-byte maps straddle 2^32 / 2^33 bytes of the map array -- against oracle replicas
    that were reset from the same draws: complete car state every 8 steps, frames every 8 steps, rewards and dones every step."""
    _need_gpu()
    import competitive_rl_amd as crl
    from competitive_rl_amd import _native as N
    from oracle import car_oracle as co
    from tests.test_hip_car_episodes import assert_state_equal as car_state_equal
    from tests.test_hip_car_episodes import batch_to_hip_state  # noqa: F401  (same state layout helpers)

    n, steps, A = 16384, 48, 8
    starts = _sample_runs(n, 739328)
    ids = np.concatenate([np.arange(s, s + RUN) for s in starts])
    assert any(s * 739328 < (1 << 32) <= (s + RUN) * 739328 for s in starts) and any(s * 739328 < (1 << 33) <= (s + RUN) * 739328 for s in starts)
    ids_dev = torch.as_tensor(ids, device="cuda")
    co.set_text(N.load_car_text())
    rs = np.random.RandomState(41)
    u = rs.random_sample((n, A, 24))
    swap = rs.randint(0, 2, (n, A)).astype(np.uint8)
    env = crl.HipCarVecEnv(n, seed=2, solver=solver)
    env.set_replay(u, swap)
    obs = env.reset()
    B = co.CarBatch(len(ids), libm="fma" if solver == "fma" else False)
    for j, i in enumerate(ids):
        o = B.view(j)
        att = o.reset(u[i].reshape(-1), 0)
        assert att > 0
        o.reset(u[i].reshape(-1), int(swap[i, att - 1]))
        B.E[j]["contacts_enabled"] = 1
        o.step(None)
    got = obs[ids_dev].cpu().numpy()
    for j in range(0, len(ids), 7):
        for v in range(2):
            assert np.array_equal(got[j, v], B.view(j).render(v)), ("first frame", int(ids[j]), v)
    g = torch.Generator(device="cuda").manual_seed(8)
    touching = 0
    for t in range(steps):
        act = torch.rand((n, 2, 2), generator=g, device="cuda") * 2 - 1
        if t < 30:
            act[:, :, 1] = act[:, :, 1].abs()
        act[:, 1, 0] = torch.where(torch.arange(n, device="cuda") % 2 == 0, -act[:, 0, 0], act[:, 1, 0])  # half the pairs steer into each other
        obs, rew, done = env.step_device(act)
        r, d = B.step(act[ids_dev].cpu().numpy().astype(np.float64))
        assert np.array_equal(rew[ids_dev].cpu().numpy(), r.astype(np.float32)), t
        assert not d.any() and not bool(done[ids_dev].any()), "no sampled env ends inside 48 steps"
        touching += int((B.E["n_contact"] > 0).sum())
        if t % 8 == 7:
            hs = env.get_state()
            car_state_equal(hs[ids], B.E, ("full size", t))
            got = obs[ids_dev].cpu().numpy()
            for j in range(t % 5, len(ids), 5):
                for v in range(2):
                    assert np.array_equal(got[j, v], B.view(j).render(v)), ("frame", t, int(ids[j]), v)
    print("config #4 sampled parity: touching env-steps among the sampled envs:", touching)
    assert env.cap_hits() == (0, 0, 0, 0)
    co.set_text(None)
    env.close()
