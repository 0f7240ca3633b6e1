"""The oracle's wrapped path against the reference's own wrappers + DummyVecEnv
(tests/golden/gen_pong_wrapped_golden.py = BASELINE config #1: 4 envs, 1000 steps)."""
import os

import numpy as np
import pytest

from oracle import pong_oracle as po

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pong_wrapped.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLDEN)


def blank_atlas():
    return np.full((22, 22, 34, 160), 255, np.uint8)  # the golden frames carry no score text


def test_wrapped_bookkeeping_and_pixels_match_reference(g):
    N, R = g["acts"].shape[1], int(g["resized_dim"])
    env = po.PongOracle(N, blank_atlas(), obs_mode=po.GRAY, resized_dim=R, frame_stack=1)
    env.set_replay(g["draw_u"], g["draw_bx"], g["draw_by"])
    obs = env.reset()
    assert np.array_equal(obs[:, :, 0], g["obs0"])
    term = {(int(t), int(i)): k for k, (t, i) in enumerate(zip(g["term_t"], g["term_i"]))}
    seen = 0
    for t in range(len(g["acts"])):
        obs, rew, done = env.step(g["acts"][t])
        assert np.array_equal(rew, g["rew"][t]), t                       # ClipRewardEnv sign
        assert np.array_equal(done.astype(bool), g["done"][t][:, 0]), t  # DummyVecEnv broadcasts the scalar
        assert np.array_equal(g["done"][t][:, 0], g["done"][t][:, 1])
        assert np.array_equal(env.real_reward, g["real_reward"][t]), t   # MaxAndSkipEnv sum
        assert np.array_equal(env.num_steps, g["num_steps"][t]), t
        assert np.array_equal(obs[:, :, 0], g["obs"][t]), t              # max-2 + gray + INTER_AREA
        for i in np.nonzero(done)[0]:
            k = term[(t, int(i))]
            assert np.array_equal(env.terminal_observation(int(i)), g["term_obs"][k]), (t, i)
            seen += 1
    assert seen == len(g["term_t"]) >= 20
    assert np.array_equal(env.state["serve_ctr"], g["ndraws"])


def test_f32_area_resize_equals_exact_rational_on_pong_frames(g):
    """The OpenCV-order f32 accumulation (oracle) and the exact rational average (golden
    generator) agree on every pixel of the fixture -- no +-1 LSB slack needed."""
    assert g["obs"].shape[2:] == (2, 42, 42)
    levels = np.unique(g["obs"])
    assert levels.size > 4  # fractional coverage levels are present, not just 0/255


def test_single_player_framestack_matches_reference():
    """cPong-v0: PongSinglePlayerEnv (AutoBat opponent) + MaxAndSkip + WarpFrame + ClipReward +
    FrameStack(4) + WrapPyTorch under the reference's DummyVecEnv (pong_single_wrapped.npz)."""
    g = np.load(os.path.join(os.path.dirname(GOLDEN), "pong_single_wrapped.npz"))
    N, R, K = g["acts"].shape[1], int(g["resized_dim"]), int(g["frame_stack"])
    env = po.PongOracle(N, blank_atlas(), obs_mode=po.GRAY, resized_dim=R, frame_stack=K, single=True, replicate=True)
    env.set_replay(g["draw_u"], g["draw_bx"], g["draw_by"])
    obs = env.reset()
    assert obs.shape == (N, 1, K, R, R)
    assert np.array_equal(obs[:, 0], g["obs0"])
    term = {(int(t), int(i)): k for k, (t, i) in enumerate(zip(g["term_t"], g["term_i"]))}
    prev = obs[:, 0].copy()
    seen = 0
    for t in range(len(g["acts"])):
        obs, rew, done = env.step(g["acts"][t])
        assert np.array_equal(rew, g["rew"][t][:, 0]), t
        assert np.array_equal(done.astype(bool), g["done"][t][:, 0]), t
        assert np.array_equal(env.real_reward[:, 0], g["real_reward"][t]), t
        assert np.array_equal(env.num_steps, g["num_steps"][t]), t
        assert np.array_equal(obs[:, 0], g["obs"][t]), t
        for i in np.nonzero(done)[0]:
            want = g["term_obs"][term[(t, int(i))]]  # FrameStack: (K, R, R) = 3 older planes + the terminal one
            assert np.array_equal(env.terminal_observation(int(i))[0], want[-1])
            assert np.array_equal(prev[i, 1:], want[:-1])
            seen += 1
        prev = obs[:, 0].copy()
    assert seen == len(g["term_t"]) >= 5
