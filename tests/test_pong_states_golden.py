"""Single frames from 20 000 random states of the reference's PongGame (tests/golden/pong_states.npz):
corner cases the trajectory goldens rarely reach.  CPU: oracle; GPU (marked): the HIP kernel."""
import os

import numpy as np
import pytest

from oracle import pong_oracle as po

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pong_states.npz")
C = {n: i for i, n in enumerate(["ball_x", "ball_y", "sx", "sy", "bat_l", "bat_r", "score_l", "score_r", "rounds", "steps", "a_l", "a_r"])}


def build_states(g):
    inp = g["inp"]
    st = np.zeros(len(inp), po.STATE_DT)
    st["ball_x"], st["ball_y"] = inp[:, C["ball_x"]], inp[:, C["ball_y"]]
    st["speed_x"], st["speed_y"] = inp[:, C["sx"]], inp[:, C["sy"]]
    st["bat_l_y"], st["bat_r_y"] = inp[:, C["bat_l"]], inp[:, C["bat_r"]]
    st["score_l"], st["score_r"] = inp[:, C["score_l"]], inp[:, C["score_r"]]
    st["num_rounds"], st["num_steps"] = inp[:, C["rounds"]], inp[:, C["steps"]]
    st["keep"]["score_l"] = 255
    st["hist"]["score_l"] = 255
    acts = np.ascontiguousarray(inp[:, [C["a_l"], C["a_r"]]].astype(np.int32))
    return st, acts


def check(g, post, rew, done, term):
    out = g["out"].view(np.int64)
    assert np.array_equal(rew.astype(np.int32), g["rew"])
    assert np.array_equal(done.astype(np.uint8), g["done"])
    live = g["done"] == 0
    for k, f in enumerate(["ball_x", "ball_y", None, None, "bat_l_y", "bat_r_y", "score_l", "score_r", "num_rounds", "num_steps"]):
        if f is None:
            continue
        assert np.array_equal(post[f][live].astype(np.int64), out[live, k]), f
    assert np.array_equal(post["speed_x"][live].view(np.uint64), g["out"][live, 2])
    assert np.array_equal(post["speed_y"][live].view(np.uint64), g["out"][live, 3])
    # finished episodes: the frame the episode ended on (pre-auto-reset state of the reference)
    fin = ~live
    for k, f in [(0, "ball_x"), (1, "ball_y"), (4, "bat_l_y"), (5, "bat_r_y"), (6, "score_l"), (7, "score_r")]:
        assert np.array_equal(term[f][fin].astype(np.int64), out[fin, k]), f
    assert fin.sum() > 1000 and (g["rew"][:, 0] != 0).sum() > 1000


def test_oracle_single_frames_match_reference(atlas):
    g = np.load(GOLD)
    st, acts = build_states(g)
    n = len(st)
    env = po.PongOracle(n, atlas, obs_mode=po.RAW)
    d = g["draws"]
    env.set_replay(d[:, 0:1], d[:, 1:2].astype(np.uint8), d[:, 2:3].astype(np.uint8))
    env.state[:] = st
    _, rew, done = env.step(acts, render=False)
    check(g, env.state, rew, done, env.terminal_frames[:, 0])


@pytest.mark.gpu
def test_hip_single_frames_match_reference():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    import competitive_rl_amd as crl

    g = np.load(GOLD)
    st, acts = build_states(g)
    n = len(st)
    env = crl.HipPongVecEnv(n, mode="raw")
    d = g["draws"]
    env.set_replay(d[:, 0:1], d[:, 1:2].astype(np.uint8), d[:, 2:3].astype(np.uint8))
    env.reset()
    env.set_state(st)
    _, rew, done = env.step_device(torch.as_tensor(acts).cuda(), render=False)
    post = env.get_state()
    # terminal frames: render-independent check through the lazily drawn terminal observation is
    # covered elsewhere; here compare the descriptors via a second state read of finished envs
    idx = np.nonzero(g["done"])[0]
    term = np.zeros(n, po.FRAME_DT)
    obs = torch.stack([torch.stack(t) for t in env.terminal_observation(idx[:64])]).cpu().numpy()
    want = po.render_raw(np.array([(o[0], o[1], o[4], o[5], o[6], o[7]) for o in g["out"].view(np.int64)[idx[:64]]], po.FRAME_DT),
                         np.ascontiguousarray(np.load(os.path.join(os.path.dirname(GOLD), "..", "..", "competitive_rl_amd", "assets",
                                                                   "pong_score_atlas.npz"))["atlas"]))
    assert np.array_equal(obs, want)
    out = g["out"].view(np.int64)
    live = g["done"] == 0
    assert np.array_equal(rew.cpu().numpy().astype(np.int32), g["rew"])
    assert np.array_equal(done.cpu().numpy(), g["done"])
    for k, f in [(0, "ball_x"), (1, "ball_y"), (4, "bat_l_y"), (5, "bat_r_y"), (6, "score_l"), (7, "score_r"), (8, "num_rounds"), (9, "num_steps")]:
        assert np.array_equal(post[f][live].astype(np.int64), out[live, k]), f
    assert np.array_equal(post["speed_x"][live].view(np.uint64), g["out"][live, 2])
    assert np.array_equal(post["speed_y"][live].view(np.uint64), g["out"][live, 3])
    env.close()
