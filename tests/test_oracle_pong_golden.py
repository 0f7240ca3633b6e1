"""The CPU oracle against golden vectors recorded from the REFERENCE's own code
(tests/golden/gen_pong_golden.py: /root/reference/competitive_rl/pong/base_pong_env.py
driven through stand-in pygame/gym).  Bit-exact: integers and f64 bit patterns."""
import numpy as np
import pytest

from oracle import pong_oracle as po

F = {n: i for i, n in enumerate(["ball_x", "ball_y", "sx_bits", "sy_bits", "bat_l", "bat_r",
                                 "score_l", "score_r", "rounds", "steps"])}


def golden_state_to_struct(row):
    s = np.zeros(1, po.STATE_DT)
    i64 = row.view(np.int64)
    s["ball_x"], s["ball_y"] = i64[F["ball_x"]], i64[F["ball_y"]]
    s["speed_x"] = row[F["sx_bits"]:F["sx_bits"] + 1].view(np.float64)[0]
    s["speed_y"] = row[F["sy_bits"]:F["sy_bits"] + 1].view(np.float64)[0]
    s["bat_l_y"], s["bat_r_y"] = i64[F["bat_l"]], i64[F["bat_r"]]
    s["score_l"], s["score_r"] = i64[F["score_l"]], i64[F["score_r"]]
    s["num_rounds"], s["num_steps"] = i64[F["rounds"]], i64[F["steps"]]
    return s


def struct_to_row(s):
    vals = [int(s["ball_x"]), int(s["ball_y"]), int(np.float64(s["speed_x"]).view(np.uint64)),
            int(np.float64(s["speed_y"]).view(np.uint64)), int(s["bat_l_y"]), int(s["bat_r_y"]),
            int(s["score_l"]), int(s["score_r"]), int(s["num_rounds"]), int(s["num_steps"])]
    return np.array([v & 0xFFFFFFFFFFFFFFFF for v in vals], np.uint64)


@pytest.mark.parametrize("name", ["random_a", "random_b", "random_cheat", "rule_vs_rule", "rule_vs_random",
                                  "sticky", "timeout"])
def test_dynamics_trace_bit_exact(golden_dyn, atlas, name):
    g = {k.split("/", 1)[1]: golden_dyn[k] for k in golden_dyn.files if k.startswith(name + "/")}
    T = len(g["acts"])
    env = po.PongOracle(1, atlas, obs_mode=po.RAW)
    env.set_replay(g["draw_u"][None], g["draw_bx"][None], g["draw_by"][None])
    env.reset(render=False)
    st = env.state
    assert np.array_equal(struct_to_row(st[0]), g["init"]), "reset state differs"
    inject = dict(zip(g["inject_t"].tolist(), g["inject_v"].tolist()))
    for t in range(T):
        if t in inject:
            st["num_steps"][0] = inject[t]
        _, rew, done = env.step(g["acts"][t][None], render=False)
        assert (int(rew[0, 0]), int(rew[0, 1])) == tuple(g["rew"][t]), (name, t)
        assert int(done[0]) == int(g["done"][t]), (name, t)
        assert np.array_equal(struct_to_row(st[0]), g["post"][t]), (name, t, struct_to_row(st[0]), g["post"][t])
        assert int(st["serve_ctr"][0]) == int(g["ndraws"][t]), (name, t)
        if done[0]:
            # terminal frame == the reference's pre-auto-reset state
            tf = env.terminal_frames[0, 0]
            pre = g["pre"][t].view(np.int64)
            assert (tf["ball_x"], tf["ball_y"], tf["bat_l_y"], tf["bat_r_y"], tf["score_l"], tf["score_r"]) == \
                   (pre[F["ball_x"]], pre[F["ball_y"]], pre[F["bat_l"]], pre[F["bat_r"]], pre[F["score_l"]],
                    pre[F["score_r"]])
    assert g["done"].sum() >= 2 or name == "timeout"


def test_timeout_rounds_counted(golden_dyn):
    # the fixture itself must contain the >10000-step branch (base_pong_env.py:233-237)
    pre = golden_dyn["timeout/pre"].view(np.int64)
    rew = golden_dyn["timeout/rew"]
    bumped = np.diff(pre[:, F["rounds"]]) > 0
    assert (bumped & (rew[1:, 0] == 0)).sum() >= 3


def test_raster_matches_reference_outside_text(golden_frames, atlas):
    st = golden_frames["state"].view(np.int64)
    frames = np.zeros(len(st), po.FRAME_DT)
    frames["ball_x"], frames["ball_y"] = st[:, F["ball_x"]], st[:, F["ball_y"]]
    frames["bat_l_y"], frames["bat_r_y"] = st[:, F["bat_l"]], st[:, F["bat_r"]]
    frames["score_l"], frames["score_r"] = st[:, F["score_l"]], st[:, F["score_r"]]
    out = po.render_raw(frames, atlas)
    v0, v1 = golden_frames["view0"], golden_frames["view1"]
    assert out.shape[1:] == (2, 210, 160, 3)
    # rows 34.. carry no text: bit-exact with the reference's draw calls + mirror
    assert np.array_equal(out[:, 0, 34:], v0[:, 34:])
    assert np.array_equal(out[:, 1, 34:], v1[:, 34:])
    # text band: the golden frames have no text (font stand-in); wherever the atlas has
    # no ink the pixels must still agree, and all ink sits in rows 8..33, cols >= 20
    ink = atlas[frames["score_l"], frames["score_r"]] < 255  # (n,34,160)
    assert not ink[:, :8].any() and not ink[:, :, :20].any()
    top0 = out[:, 0, :34, :, 0]
    assert np.array_equal(top0[~ink], v0[:, :34, :, 0][~ink])
    # mirror quirk (base_pong_env.py:153-154): rows < 25 identical, rows >= 25 flipped
    assert np.array_equal(out[:, 1, :25], out[:, 0, :25])
    assert np.array_equal(out[:, 1, 25:], out[:, 0, 25:, ::-1])
    # achromatic
    assert np.array_equal(out[..., 0], out[..., 1]) and np.array_equal(out[..., 0], out[..., 2])
