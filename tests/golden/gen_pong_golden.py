"""Generate golden vectors for the Pong hot path from the REFERENCE's own code.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_pong_golden.py

Loads ``/root/reference/competitive_rl/pong/base_pong_env.py`` by path with the
stand-ins of ``_ref_standins.py`` (no gym/pygame in this image), drives
``PongDoublePlayerEnv._step/_reset`` (base_pong_env.py:113-147) and records

* ``pong_dynamics.npz``  -- per-frame actions + serve draws -> full game state,
  rewards, done (bit patterns for the f64 speeds), with DummyVecEnv-style
  auto-reset (dummy_vec_env.py:55-58);
* ``pong_frames.npz``    -- a few dozen raw (210,160,3) frames for both views,
  WITHOUT score text (font stand-in is a no-op; text is unpinned).

Only data is written: inputs and expected outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_standins as S  # noqa: E402

S.install()
pong = S.load_ref("competitive_rl.pong.base_pong_env", "pong/base_pong_env.py")


def f64_bits(x):
    return np.float64(x).view(np.uint64)


def to_u64(vals):
    """ints (possibly negative) and uint64 bit patterns -> one uint64 row (two's complement)."""
    return np.array([int(v) & 0xFFFFFFFFFFFFFFFF for v in vals], dtype=np.uint64)


def snapshot(game):
    b = game._ball
    return to_u64((
        b._rect.x, b._rect.y, f64_bits(b._speed_x), f64_bits(b._speed_y),
        game._left_bat._rect.y, game._right_bat._rect.y,
        game._score_left, game._score_right, game._num_rounds, game._num_steps,
    ))


FIELDS = ["ball_x", "ball_y", "sx_bits", "sy_bits", "bat_l", "bat_r",
          "score_l", "score_r", "rounds", "steps"]


def run_trace(policy, n_frames, seed, inject=None, grab_frames=()):
    """policy(t, rs) -> (a_left, a_right).  Returns dict of arrays."""
    stream = S.ServeStream(1000 + seed)
    pong.random = stream  # the module-level name the reference draws from
    env = pong.PongDoublePlayerEnv(max_num_rounds=21)  # pong/register.py:20-22
    # construction consumed 2 draws (Ball.__init__, reset_game); forget them so
    # the recorded stream starts at the first reset()
    stream.u.clear(), stream.bx.clear(), stream.by.clear()
    env.reset()
    game = env._game
    rs = np.random.RandomState(seed)
    init = snapshot(game)
    acts = np.zeros((n_frames, 2), np.int32)
    pre = np.zeros((n_frames, len(FIELDS)), np.uint64)
    post = np.zeros((n_frames, len(FIELDS)), np.uint64)
    rew = np.zeros((n_frames, 2), np.int32)
    done = np.zeros((n_frames,), np.uint8)
    ndraws = np.zeros((n_frames,), np.int32)  # draws consumed up to and incl. frame t
    frames = {}
    for t in range(n_frames):
        if inject is not None and t in inject:
            game._num_steps = inject[t]
        a = policy(t, rs)
        acts[t] = a
        obs, r, d, info = env.step(a)
        assert info == {}
        pre[t] = snapshot(game)
        rew[t] = r
        done[t] = d
        if t in grab_frames:
            frames[t] = (obs[0].copy(), obs[1].copy())
        if d:
            env.reset()
        post[t] = snapshot(game)
        ndraws[t] = len(stream.u)
    assert len(stream.u) == len(stream.bx) == len(stream.by)
    return dict(
        init=init, acts=acts, pre=pre, post=post,
        rew=rew, done=done, ndraws=ndraws,
        draw_u=np.array(stream.u, np.float64), draw_bx=np.array(stream.bx, np.uint8),
        draw_by=np.array(stream.by, np.uint8), inject_t=np.array(sorted((inject or {}).keys()), np.int64),
        inject_v=np.array([inject[k] for k in sorted((inject or {}).keys())], np.int64),
    ), frames


def pol_random(t, rs):
    return int(rs.randint(0, 3)), int(rs.randint(0, 3))


def pol_random_cheat(t, rs):
    a = [int(rs.randint(0, 3)), int(rs.randint(0, 3))]
    for k in range(2):
        if rs.random_sample() < 0.3:
            a[k] = 999
    return tuple(a)


def pol_rule_vs_rule(t, rs):
    return 999, 999


def pol_rule_vs_random(t, rs):
    return 999, int(rs.randint(0, 3))


def pol_sticky(t, rs):
    # long runs of one direction: bats pinned on the walls, moving-bat hits
    if t % 37 == 0:
        pol_sticky.cur = (int(rs.randint(0, 3)), int(rs.randint(0, 3)))
    return pol_sticky.cur


pol_sticky.cur = (1, 1)


def main():
    out = {}
    frames_v0, frames_v1, frames_state = [], [], []
    traces = [
        ("random_a", pol_random, 6000, 1, None),
        ("random_b", pol_random, 6000, 2, None),
        ("random_cheat", pol_random_cheat, 6000, 3, None),
        ("rule_vs_rule", pol_rule_vs_rule, 12000, 4, None),
        ("rule_vs_random", pol_rule_vs_random, 6000, 5, None),
        ("sticky", pol_sticky, 6000, 6, None),
        # forced round time-out: num_steps jumps to 9 998 three times
        ("timeout", pol_rule_vs_rule, 3000, 7, {100: 9998, 900: 9998, 1700: 9999}),
    ]
    names = []
    for name, pol, n, seed, inject in traces:
        grab = set(range(0, n, max(1, n // 8))) if name in ("random_a", "sticky", "random_cheat") else ()
        tr, frames = run_trace(pol, n, seed, inject, grab)
        for k, v in tr.items():
            out[f"{name}/{k}"] = v
        names.append(name)
        for t, (v0, v1) in sorted(frames.items()):
            frames_v0.append(v0), frames_v1.append(v1)
            frames_state.append(tr["pre"][t])  # obs of step t shows the pre-auto-reset state
        ev = tr["rew"][:, 0]
        print(f"{name:16s} frames={n} points={np.count_nonzero(ev)} dones={int(tr['done'].sum())} "
              f"draws={len(tr['draw_u'])} timeouts={(np.diff(tr['pre'][:, 8].astype(np.int64)) > 0).sum() - np.count_nonzero(ev)}")
    out["names"] = np.array(names)
    out["fields"] = np.array(FIELDS)
    np.savez_compressed(os.path.join(HERE, "pong_dynamics.npz"), **out)
    np.savez_compressed(
        os.path.join(HERE, "pong_frames.npz"),
        view0=np.stack(frames_v0), view1=np.stack(frames_v1), state=np.stack(frames_state),
        fields=np.array(FIELDS),
    )
    print("frames:", len(frames_v0))


if __name__ == "__main__":
    main()
