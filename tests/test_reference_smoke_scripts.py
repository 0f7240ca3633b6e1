"""The reference's own test scripts (competitive_rl/test/test_pong.py, test_car_racing.py), call for call, through this package's
``make_envs`` on the GPU: same arguments, same attribute accesses, the shapes the reference prints.  (test_make_car_racing_double.py builds
ONE gym env through ``make_car_racing_double(...)()``: the single-env facade is not on the vectorised hot path -- the same env as a
batch of one is ``make_envs("cCarRacingDouble-v0", num_envs=1, ...)``, exercised below.)"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")


def test_reference_test_pong_script(tmp_path):
    _need_gpu()
    from competitive_rl_amd import make_envs

    envs = make_envs(
        env_id="cPong-v0",
        seed=0,
        log_dir=str(tmp_path / "demo"),  # this will create a "demo" directory
        num_envs=1,
        asynchronous=False,
        resized_dim=42
    )
    env = envs.envs[0]
    obs = envs.reset()
    env.close()
    assert os.path.isdir(tmp_path / "demo")
    assert tuple(obs.shape) == (1, 4, 42, 42)          # (the reference prints this: FrameStack(4) of (1, 42, 42) frames, one env)
    assert env.observation_space.shape == (4, 42, 42) and env.action_space.n == 3
    envs.close()


def test_reference_test_car_racing_script(tmp_path):
    _need_gpu()
    from competitive_rl_amd import make_envs

    envs = make_envs(
        env_id="cCarRacing-v0",
        seed=0,
        log_dir=str(tmp_path / "demo"),  # this will create a "demo" directory
        num_envs=5,
        asynchronous=True,
        resized_dim=42
    )
    obs = envs.reset()
    assert tuple(obs.shape) == (5, 4, 96, 96)          # frame_stack defaults to 4; resized_dim does not apply to CarRacing (96 x 96 as rendered)
    assert os.path.isdir(tmp_path / "demo")
    envs.close()


def test_reference_make_car_racing_double_as_a_batch_of_one():
    """make_car_racing_double(0, 0, 4, None)(); reset(); step(action_space.sample()); close()  -- as one env of the vector backend"""
    _need_gpu()
    from competitive_rl_amd import make_envs

    e = make_envs("cCarRacingDouble-v0", seed=0, log_dir=None, num_envs=1, frame_stack=4, action_repeat=None)
    obs = e.reset()
    a = e.action_space.sample()                        # Box(-1, 1, (2, 2)): FlattenMultiAgentObservation's action space (utils/atari_wrappers.py:316)
    assert np.asarray(a).shape == (2, 2) and e.action_space.contains(a)
    obs2, rew, done, info = e.step([a])                # one env: a list of one per-env action
    # reward = car 0's (FlattenMultiAgentObservation returns r[0], :329), DummyVecEnv's (N, 1) buffers; both cars' in the infos
    assert tuple(obs.shape) == tuple(obs2.shape) == (1, 8, 96, 96) and tuple(rew.shape) == (1, 1) and tuple(done.shape) == (1, 1)
    assert float(info[0][0]["reward"]) == float(rew[0, 0]) and "reward" in info[0][1]
    obs3, rew3, _, _ = e.step([{0: a[0], 1: a[1]}])    # the raw env's Dict form {car: action} (car_racing_multi_players.py:245) is indexed the same way
    assert tuple(obs3.shape) == tuple(obs.shape)
    e.close()


def test_render_tiles_and_a_vec_env_wrapper_over_the_hip_env():
    """``envs.render("rgb_array")`` of several envs is ONE tiled picture (VecEnv.render -> tile_images, utils/base_vec_env.py:10-38,173-192);
    a user's ``VecEnvWrapper`` subclass (:255-374) wraps what ``make_envs`` returns and finds the env's attributes through the chain."""
    _need_gpu()
    import competitive_rl_amd as crl

    envs = crl.make_envs("cPongDouble-v0", num_envs=3, frame_stack=None, log_dir=None, resized_dim=42)

    class ClipObs(crl.VecEnvWrapper):
        def reset(self):
            return self.venv.reset()

        def step_wait(self):
            o, r, d, i = self.venv.step_wait()
            return o, r * 2, d, i

    w = ClipObs(envs)
    o = w.reset()
    o2, r, d, info = w.step(np.array([[0, 1], [2, 0], [999, 999]]))
    assert tuple(o2[0].shape) == tuple(o[0].shape) == (3, 1, 42, 42) and tuple(r.shape) == (3, 2)
    big = w.render("rgb_array")
    each = envs.get_images()
    assert big.shape == (2 * 210, 2 * 160, 3) and big.dtype == np.uint8
    assert np.array_equal(big[:210, 160:], each[1]) and np.array_equal(big[210:, :160], each[2]) and not big[210:, 160:].any()
    assert w.unwrapped is envs and w.device == envs.device and w.num_envs == 3
    assert crl.make_envs("cPongDouble-v0", num_envs=1, frame_stack=None, log_dir=None).render().shape == (210, 160, 3)
    w.close()
