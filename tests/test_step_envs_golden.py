"""``step_envs`` + ``FrameStackTensor`` and the SubprocVecEnv conventions (row P20) against fixtures
recorded from the reference's own code (tests/golden/gen_step_envs_golden.py: utils/utils.py:23-60,145-173 over the
reference's DummyVecEnv; utils/subproc_vec_env.py with one forked worker per env).

CPU tests here drive this package's ``step_envs`` / ``FrameStackTensor`` (host mode) with the oracle as the env;
the ``-m gpu`` tests run the same fixtures through the HIP env and the HIP frame-stack kernel.
"""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def blank_atlas():
    return np.full((22, 22, 34, 160), 255, np.uint8)  # the golden frames carry no score text


class OracleVecEnv:
    """DummyVecEnv-shaped adapter over the oracle batch (test infrastructure)."""

    def __init__(self, g, n, R):
        from oracle import pong_oracle as po

        self.o = po.PongOracle(n, blank_atlas(), obs_mode=po.GRAY, resized_dim=R, frame_stack=1)
        self.o.set_replay(g["draw_u"], g["draw_bx"], g["draw_by"])
        self.n = n

    def reset(self):
        obs = self.o.reset()
        return tuple(obs[:, v].astype(np.float32) for v in range(2))

    def step(self, actions):
        obs, rew, done = self.o.step(actions)
        infos = [{"real_reward": list(map(float, self.o.real_reward[i])), "num_steps": int(self.o.num_steps[i])} for i in range(self.n)]
        d = done.astype(bool)
        return tuple(obs[:, v].astype(np.float32) for v in range(2)), rew.copy(), np.stack([d, d], 1), infos


def replay_step_envs(g, venv, device, to_host):
    """Runs this package's step_envs over the fixture's action sequence and compares every output with the reference's."""
    import competitive_rl_amd as crl

    N, R, K = g["acts"].shape[1], int(g["resized_dim"]), int(g["frame_stack"])
    fst = crl.FrameStackTensor(N, (1, R, R), K, device)
    obs0 = venv.reset()
    fst.update(obs0[0])
    assert np.array_equal(to_host(fst.get()), g["stack0"].astype(np.float32))
    episode_rewards = np.zeros((N, 2), np.float64)
    reward_recorder, length_recorder = [], []
    total_steps = total_episodes = 0
    for t in range(len(g["acts"])):
        ret = crl.step_envs(g["acts"][t], venv, episode_rewards, fst, reward_recorder, length_recorder, total_steps, total_episodes,
                            device, False)
        obs, reward, done, info, masks, total_episodes, total_steps, episode_rewards = ret
        assert tuple(masks.shape) == (N, 1) and masks.dtype == torch.float32
        assert np.array_equal(to_host(masks)[:, 0], g["masks"][t]), t
        assert np.array_equal(to_host(done).astype(bool), g["done"][t]), t
        assert total_steps == g["total_steps"][t] and total_episodes == g["total_episodes"][t], t
        assert np.array_equal(episode_rewards, g["episode_rewards"][t]), t
        assert len(reward_recorder) == g["rec_count"][t], t
        st = to_host(fst.get())
        assert st.dtype == np.float32 and np.array_equal(st, g["stack"][t].astype(np.float32)), t
    assert np.array_equal(np.array(reward_recorder), g["reward_recorder"])
    assert np.array_equal(np.array(length_recorder), g["length_recorder"])
    assert total_episodes >= 15


def test_step_envs_and_frame_stack_match_reference_host_mode():
    g = np.load(os.path.join(G, "step_envs.npz"))
    venv = OracleVecEnv(g, g["acts"].shape[1], int(g["resized_dim"]))
    replay_step_envs(g, venv, "cpu", lambda t: t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t))


def test_frame_stack_tensor_mask_shapes_and_errors():
    import competitive_rl_amd as crl

    f = crl.FrameStackTensor(3, (2, 4, 4), 3, "cpu")
    assert f.obs_shape == (6, 4, 4) and tuple(f.get().shape) == (3, 6, 4, 4)
    a = np.arange(3 * 2 * 16, dtype=np.uint8).reshape(3, 2, 4, 4)
    f.update(a)
    f.update(a + 1, np.array([1.0, 0.0, 1.0], np.float32).reshape(3, 1, 1, 1))
    got = f.get().numpy()
    assert np.array_equal(got[0, 2:4], a[0]) and np.array_equal(got[0, 4:], a[0] + 1) and not got[0, :2].any()
    assert not got[1, :4].any() and np.array_equal(got[1, 4:], a[1] + 1)     # history erased, newest kept
    with pytest.raises(ValueError):
        f.update(a[:2])
    with pytest.raises(ValueError):
        f.update(a, np.ones(4, np.float32))
    f.reset()
    assert not f.get().any()


def test_subproc_fixture_conventions_hold_for_the_oracle():
    """P20: what SubprocVecEnv returns for wrapped cPongDouble (recorded): obs uint8 tuple, rews (N, 2), dones (N,),
    infos a tuple; the oracle reproduces values, terminal observations and auto-reset timing of the forked workers."""
    from oracle import pong_oracle as po

    g = np.load(os.path.join(G, "pong_subproc.npz"))
    assert str(g["meta_obs_type"]) == "tuple" and str(g["meta_obs_dtype"]) == "uint8" and str(g["meta_reset_dtype"]) == "uint8"
    assert g["meta_done_shape"].tolist() == [3] and str(g["meta_done_dtype"]) == "bool" and g["meta_rew_shape"].tolist() == [3, 2]
    assert str(g["meta_infos_type"]) == "tuple"
    N, R = g["acts"].shape[1], int(g["resized_dim"])
    env = po.PongOracle(N, blank_atlas(), obs_mode=po.GRAY, resized_dim=R, frame_stack=1)
    env.set_replay(g["draw_u"], g["draw_bx"], g["draw_by"])
    assert np.array_equal(env.reset()[:, :, 0], g["obs0"])
    term = {(int(t), int(i)): k for k, (t, i) in enumerate(zip(g["term_t"], g["term_i"]))}
    for t in range(len(g["acts"])):
        obs, rew, done = env.step(g["acts"][t])
        assert np.array_equal(rew, g["rew"][t].astype(np.float32)) and np.array_equal(done.astype(bool), g["done"][t]), t
        assert np.array_equal(obs[:, :, 0], g["obs"][t]), t
        assert np.array_equal(env.real_reward, g["real_reward"][t]) and np.array_equal(env.num_steps, g["num_steps"][t]), t
        for i in np.nonzero(done)[0]:
            assert np.array_equal(env.terminal_observation(int(i)), g["term_obs"][term[(t, int(i))]]), (t, i)
    assert len(term) >= 5


# --------------------------------------------------------------------------------------------- GPU
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")


@pytest.mark.gpu
def test_step_envs_and_frame_stack_match_reference_through_hip():
    """The same fixture with the HIP env underneath and the HIP frame-stack kernel: observations never leave HBM."""
    _need_gpu()
    import competitive_rl_amd as crl

    g = np.load(os.path.join(G, "step_envs.npz"))
    N, R = g["acts"].shape[1], int(g["resized_dim"])
    venv = crl.HipPongVecEnv(N, mode="wrapped", resized_dim=R, frame_stack=1, score_atlas=blank_atlas())
    venv.set_replay(g["draw_u"], g["draw_bx"], g["draw_by"])
    replay_step_envs(g, venv, venv.device, lambda t: t.cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t))
    venv.close()


@pytest.mark.gpu
@pytest.mark.parametrize("out_of_place", [True, False])
@pytest.mark.parametrize("dtype", ["uint8", "float32"])
def test_frame_stack_kernel_equals_torch_ops(dtype, out_of_place):
    """crl_frame_stack_update / crl_frame_stack_update_to against the three torch ops they replace, on strided uint8 / float32
    observations, vector and scalar paths, with and without a mask.  Out of place (the default: the reference binds a new tensor
    per update too) the tensor the LAST update handed out must still hold its values after this one."""
    _need_gpu()
    import competitive_rl_amd as crl

    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(5)
    for (n, c, k, h, w) in ((37, 1, 4, 84, 84), (5, 2, 3, 7, 9), (64, 1, 1, 42, 42), (3, 3, 2, 96, 96)):
        f = crl.FrameStackTensor(n, (c, h, w), k, dev, out_of_place=out_of_place)
        ref = torch.zeros((n, c * k, h, w), device=dev)
        last_out = last_ref = None
        for step in range(6):
            wide = torch.randint(0, 256, (n, 2, c, h, w), generator=g, device=dev, dtype=torch.uint8)
            obs = wide[:, 1] if dtype == "uint8" else wide[:, 1].float()     # a strided view, like agent 1's half
            mask = None if step % 3 == 0 else (torch.rand((n, 1, 1, 1), generator=g, device=dev) > 0.3).float()
            if mask is not None:
                ref *= mask
            ref = ref.roll(shifts=-c, dims=1)
            ref[:, -c:] = obs.float()
            out = f.update(obs, mask)
            assert torch.equal(out, ref), (n, c, k, h, w, step)
            assert out.data_ptr() == f.get().data_ptr()
            if out_of_place and last_out is not None:
                assert out.data_ptr() != last_out.data_ptr() and torch.equal(last_out, last_ref), (n, c, k, h, w, step)
            last_out, last_ref = out, ref.clone()


@pytest.mark.gpu
@pytest.mark.parametrize("out_of_place", [True, False])
@pytest.mark.parametrize("dtype", ["uint8", "float32"])
def test_uint8_frame_stack_kernel_equals_tensor_ops(dtype, out_of_place):
    """crl_frame_stack_update_u8 (the generic update of the opt-in byte stack, in place and into the other buffer) against the tensor
    operations it stands for: history kept or erased by the mask's zeros, planes shifted by C, the observation (uint8, or float32 holding
    integers; strided) as the newest planes -- 16-byte and one-byte paths."""
    _need_gpu()
    import competitive_rl_amd as crl

    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(6)
    for (n, c, k, h, w) in ((37, 1, 4, 84, 84), (5, 2, 3, 7, 9), (64, 1, 1, 42, 42), (3, 3, 2, 96, 96), (9, 1, 4, 42, 42)):
        f = crl.FrameStackTensor(n, (c, h, w), k, dev, out_of_place=out_of_place, dtype=torch.uint8)
        ref = torch.zeros((n, c * k, h, w), device=dev, dtype=torch.uint8)
        last_out = last_ref = None
        for step in range(6):
            wide = torch.randint(0, 256, (n, 2, c, h, w), generator=g, device=dev, dtype=torch.uint8)
            obs = wide[:, 1] if dtype == "uint8" else wide[:, 1].float()
            mask = None if step % 3 == 0 else (torch.rand((n, 1, 1, 1), generator=g, device=dev) > 0.3).float()
            if mask is not None:
                ref = ref * (mask != 0).to(torch.uint8)
            ref = ref.roll(shifts=-c, dims=1)
            ref[:, -c:] = wide[:, 1]
            out = f.update(obs, mask)
            assert out.dtype == torch.uint8 and torch.equal(out, ref), (n, c, k, h, w, step)
            assert out.data_ptr() == f.get().data_ptr()
            if out_of_place and last_out is not None:
                assert out.data_ptr() != last_out.data_ptr() and torch.equal(last_out, last_ref), (n, c, k, h, w, step)
            last_out, last_ref = out, ref.clone()


@pytest.mark.gpu
def test_frame_stack_update_to_rejects_overlapping_tensors():
    _need_gpu()
    import ctypes as C

    import competitive_rl_amd as crl
    from competitive_rl_amd import _native as N

    lib = N.load()
    buf = torch.zeros(2 * 4 * 8 * 8 + 64, device="cuda")
    obs = torch.zeros((2, 1, 8, 8), dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.crl_frame_stack_update_to(C.c_void_p(buf.data_ptr() + 64), C.c_void_p(buf.data_ptr()), C.c_void_p(obs.data_ptr()), N.CRL_OBS_U8, 64,
                                       None, 2, 1, 4, 64, st)
    assert rc != 0
    with pytest.raises(RuntimeError, match="overlap"):
        N.check(rc)
    assert crl.FrameStackTensor(2, (1, 8, 8), 4, "cuda").out_of_place


@pytest.mark.gpu
def test_subproc_conventions_through_hip():
    """P20: make_envs(asynchronous=True) returns what the reference's SubprocVecEnv returns (pong_subproc.npz)."""
    _need_gpu()
    import competitive_rl_amd as crl

    g = np.load(os.path.join(G, "pong_subproc.npz"))
    N, R = g["acts"].shape[1], int(g["resized_dim"])
    env = crl.make_envs("cPongDouble-v0", num_envs=N, asynchronous=True, resized_dim=R, frame_stack=None, log_dir=None,
                        output="numpy", score_atlas=blank_atlas())
    env.set_replay(g["draw_u"], g["draw_bx"], g["draw_by"])
    o0 = env.reset()
    assert isinstance(o0, tuple) and o0[0].dtype == np.uint8 and o0[0].shape == tuple(g["meta_obs_shape"])
    term = {(int(t), int(i)): k for k, (t, i) in enumerate(zip(g["term_t"], g["term_i"]))}
    for t in range(len(g["acts"])):
        o, r, d, infos = env.step(g["acts"][t])
        assert isinstance(o, tuple) and o[0].dtype == np.uint8 and o[0].shape == tuple(g["meta_obs_shape"])
        assert d.shape == tuple(g["meta_done_shape"]) and d.dtype == np.bool_ and r.shape == tuple(g["meta_rew_shape"])
        assert np.array_equal(r, g["rew"][t].astype(np.float32)) and np.array_equal(d, g["done"][t]), t
        assert np.array_equal(np.stack([o[0][:, 0], o[1][:, 0]], 1), g["obs"][t]), t
        for i in range(N):
            assert infos[i]["num_steps"] == g["num_steps"][t][i] and infos[i]["real_reward"] == list(g["real_reward"][t][i])
            assert ("terminal_observation" in infos[i]) == bool(d[i])
            if d[i]:
                to = infos[i]["terminal_observation"]
                want = g["term_obs"][term[(t, i)]]
                assert isinstance(to, tuple) and to[0].shape == (1, R, R)
                assert np.array_equal(np.stack([to[0][0], to[1][0]]), want), (t, i)
    env.close()
