"""Golden vectors for the two-policy match loops that sit directly on ``VecEnv.step`` (the caller side of the hot path):
the reference's ``evaluate_two_policies_in_batch`` and ``evaluate_two_policies`` (pong/evaluate.py:53-88, 6-50).

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_pong_evaluate_golden.py

``pong_evaluate.npz``: the reference's own functions (loaded by path; ``np.float`` -- removed from numpy 1.24, used at
pong/evaluate.py:56 -- is given back as ``float`` for the call) over the reference's ``DummyVecEnv`` of wrapped cPongDouble envs
(the flow of gen_pong_wrapped_golden.py, R = 42) and over ONE wrapped env.  The two policies are pure functions of the observation
(``obs_policy`` below: a weighted pixel sum mod 3 -- every pixel of every observation steers the match, so a replay that differs in
one pixel leaves the recorded trajectory) and the rule-based opponent (CHEAT_CODES = 999, pong/builtin_policies.py:44-48).
Recorded: the returned game results, the per-step actions / rewards / dones the loop saw, and the per-env serve draws.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_standins as S  # noqa: E402
from gen_pong_wrapped_golden import Router, make_cv2  # noqa: E402
from gen_step_envs_golden import serve_arrays  # noqa: E402

CHEAT = 999


def obs_policy(obs):
    """(N, 1, R, R) or (1, R, R) observation -> actions in {0, 1, 2}: sum_k (k mod 7 + 1) * pixel_k, mod 3 (exact in float64)."""
    o = np.asarray(obs, np.float64)
    batch = o.ndim == 4
    o = o.reshape(o.shape[0] if batch else 1, -1)
    a = ((o * (np.arange(o.shape[1]) % 7 + 1)).sum(1).astype(np.int64) % 3)
    return a if batch else int(a[0])


def main():
    S.install(cv2_module=make_cv2())
    import gym

    pong = S.load_ref("competitive_rl.pong.base_pong_env", "pong/base_pong_env.py")
    S.load_ref("competitive_rl.pong.register", "pong/register.py").register_pong()
    aw = S.load_ref("competitive_rl.utils.atari_wrappers", "utils/atari_wrappers.py")
    S.load_ref("competitive_rl.utils.vec_env_utils", "utils/vec_env_utils.py")
    S.load_ref("competitive_rl.utils.base_vec_env", "utils/base_vec_env.py")
    dv = S.load_ref("competitive_rl.utils.dummy_vec_env", "utils/dummy_vec_env.py")
    ev = S.load_ref("competitive_rl.pong.evaluate", "pong/evaluate.py")
    if not hasattr(np, "float"):
        np.float = float  # pong/evaluate.py:56 (numpy < 1.24's alias of the builtin)

    R = 42
    out = {}

    # ------------------------------------------------------------------ evaluate_two_policies_in_batch
    def run_batch(tag, N, seed, num_episodes, side0_is_rule):
        router = Router(N, seed)
        pong.random = router

        def thunk(i):
            inner = aw.make_env_a2c_atari("cPongDouble-v0", 0, i, None, R, None)

            def f():
                router.cur = i

                class Tag(gym.Wrapper):
                    def step(self, a):
                        router.cur = i
                        return self.env.step(a)

                    def reset(self, **kw):
                        router.cur = i
                        return self.env.reset(**kw)

                return Tag(inner())

            return f

        venv = dv.DummyVecEnv([thunk(i) for i in range(N)])
        for s in router.streams:
            s.u.clear(), s.bx.clear(), s.by.clear()
        trace = dict(acts=[], rew=[], done=[])
        step0 = venv.step

        def step(actions):  # (what the loop sends and gets, for diagnosis of a replay that leaves the recording)
            o, r, d, info = step0(actions)
            trace["acts"].append(np.asarray(actions).copy()), trace["rew"].append(r.copy()), trace["done"].append(d.copy())
            return o, r, d, info

        venv.step = step
        rule = lambda obs: [CHEAT] * N  # noqa: E731  (get_rule_based_policy, pong/builtin_policies.py:44-48)
        c0, c1 = (rule, obs_policy) if side0_is_rule else (obs_policy, rule)
        r0, r1 = ev.evaluate_two_policies_in_batch(c0, c1, venv, num_episodes)
        u, bx, by, nd = serve_arrays(router.streams)
        out.update({f"{tag}_result0": np.array(r0, np.float64), f"{tag}_result1": np.array(r1, np.float64), f"{tag}_acts": np.array(trace["acts"], np.int32),
                    f"{tag}_rew": np.array(trace["rew"]), f"{tag}_done": np.array(trace["done"]), f"{tag}_draw_u": u, f"{tag}_draw_bx": bx,
                    f"{tag}_draw_by": by, f"{tag}_ndraws": nd, f"{tag}_num_episodes": num_episodes, f"{tag}_side0_is_rule": int(side0_is_rule)})
        print(tag, "steps", len(trace["acts"]), "results", r0, r1)

    run_batch("batch_a", 4, 6100, 6, False)   # the observation-driven policy on the left against the rule-based bat
    run_batch("batch_b", 3, 6500, 4, True)    # sides swapped

    # ------------------------------------------------------------------ evaluate_two_policies (one env, explicit reset per episode)
    stream = S.ServeStream(6900)
    pong.random = stream
    env = aw.make_env_a2c_atari("cPongDouble-v0", 0, 0, None, R, None)()
    stream.u.clear(), stream.bx.clear(), stream.by.clear()
    trace = dict(acts=[], rew=[], done=[])
    step1 = env.step

    def step(action):
        o, r, d, info = step1(action)
        trace["acts"].append(np.asarray(action).copy()), trace["rew"].append(np.asarray(r, np.float32).copy()), trace["done"].append(bool(np.all(d)))
        return o, r, d, info

    env.step = step
    r0, r1 = ev.evaluate_two_policies(obs_policy, lambda obs: CHEAT, env, 3)
    u, bx, by, nd = serve_arrays([stream])
    out.update(single_result0=np.array(r0, np.float64), single_result1=np.array(r1, np.float64), single_acts=np.array(trace["acts"], np.int32),
               single_rew=np.array(trace["rew"]), single_done=np.array(trace["done"]), single_draw_u=u, single_draw_bx=bx, single_draw_by=by,
               single_ndraws=nd, single_num_episodes=3)
    print("single steps", len(trace["acts"]), "results", r0, r1)
    np.savez_compressed(os.path.join(HERE, "pong_evaluate.npz"), resized_dim=R, **out)


if __name__ == "__main__":
    main()
