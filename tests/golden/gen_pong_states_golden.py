"""Golden vectors for single frames from RANDOM STATES of the reference's PongGame.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_pong_states_golden.py

Trajectory goldens (gen_pong_golden.py) rarely visit corner cases: ball about to cross a bat face
while a wall bounce happens in the same frame, |speed_y| large, bats pinned at the limits, ball
already behind a bat, round time-out on the same frame as a score, score 20 -> done...  Here the
reference's ``PongDoublePlayerEnv._step`` (pong/base_pong_env.py:113-142) is run for ONE frame from
20 000 randomly injected states; inputs and outputs are recorded (f64 as bit patterns).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_standins as S  # noqa: E402

S.install()
pong = S.load_ref("competitive_rl.pong.base_pong_env", "pong/base_pong_env.py")


def main():
    n = 20000
    rs = np.random.RandomState(123)
    stream = S.ServeStream(77)
    pong.random = stream
    env = pong.PongDoublePlayerEnv(max_num_rounds=21)
    g = env._game
    cols = ["ball_x", "ball_y", "sx", "sy", "bat_l", "bat_r", "score_l", "score_r", "rounds", "steps", "a_l", "a_r"]
    inp = np.zeros((n, len(cols)), np.float64)
    out = np.zeros((n, 10), np.uint64)   # ball_x, ball_y, sx bits, sy bits, bat_l, bat_r, score_l, score_r, rounds, steps
    rew = np.zeros((n, 2), np.int32)
    done = np.zeros(n, np.uint8)
    draws = np.zeros((n, 3), np.float64)  # the serve draw available to this frame (u, bx, by)
    used = np.zeros(n, np.uint8)
    for i in range(n):
        kind = rs.randint(6)
        bx = int(rs.choice([rs.randint(0, 157), rs.randint(14, 30), rs.randint(128, 142), rs.randint(0, 6), rs.randint(152, 157)]))
        by = int(rs.choice([rs.randint(34, 191), rs.randint(30, 40), rs.randint(185, 196)]))
        sx = float(rs.choice([-4.0, 4.0]))
        sy = float(rs.choice([rs.uniform(-4, 4), rs.uniform(-12, 12), 0.0, rs.choice([-2.8, 2.8, 1.2, -1.2])]))
        bl = int(rs.choice([rs.randint(34, 180), 34, 179, 35, 178]))
        br = int(rs.choice([rs.randint(34, 180), 34, 179, 36, 177]))
        rounds = int(rs.choice([rs.randint(0, 21), 20, 19]))
        sl = int(rs.randint(0, rounds + 1))
        sr = int(rs.randint(0, rounds - sl + 1))
        steps = int(rs.choice([rs.randint(0, 5000), 9999, 10000, 10001]))
        a = [int(rs.choice([0, 1, 2, 999])), int(rs.choice([0, 1, 2, 999]))]
        g._ball._rect.x, g._ball._rect.y = bx, by
        g._ball._speed_x, g._ball._speed_y = sx, sy
        g._left_bat._rect.y, g._right_bat._rect.y = bl, br
        g._score_left, g._score_right, g._num_rounds, g._num_steps = sl, sr, rounds, steps
        inp[i] = [bx, by, sx, sy, bl, br, sl, sr, rounds, steps, a[0], a[1]]
        n0 = len(stream.u)
        u_next = float(np.random.RandomState(1000 + i).random_sample())
        # make the draw of this frame explicit: peek by forcing the stream's generator state
        o, r, d, _ = env.step(tuple(a))
        if len(stream.u) > n0:
            used[i] = 1
            draws[i] = [stream.u[-1], stream.bx[-1], stream.by[-1]]
        b = g._ball
        out[i] = [int(v) & 0xFFFFFFFFFFFFFFFF for v in (
            b._rect.x, b._rect.y, int(np.float64(b._speed_x).view(np.uint64)), int(np.float64(b._speed_y).view(np.uint64)),
            g._left_bat._rect.y, g._right_bat._rect.y, g._score_left, g._score_right, g._num_rounds, g._num_steps)]
        rew[i], done[i] = r, d
    np.savez_compressed(os.path.join(HERE, "pong_states.npz"), inp=inp, out=out, rew=rew, done=done, draws=draws, used=used,
                        cols=np.array(cols))
    print("states", n, "scored", int((rew[:, 0] != 0).sum()), "done", int(done.sum()), "serves", int(used.sum()),
          "bat hits", int(((inp[:, 2] * out[:, 2].view(np.float64)) < 0).sum()))


if __name__ == "__main__":
    main()
