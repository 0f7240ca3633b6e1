"""Golden vectors for the trainer-side glue right after ``VecEnv.step`` and for the
SubprocVecEnv conventions.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_step_envs_golden.py

Part 1 -- ``step_envs.npz``: the reference's own ``step_envs`` and ``FrameStackTensor``
(utils/utils.py:23-60,145-173; torch is installed) called in a loop over the reference's
``DummyVecEnv`` of wrapped cPongDouble envs (the flow of gen_pong_wrapped_golden.py, 4 envs, R = 42).
Recorded per call: every element of the returned tuple, the recorders, and the whole stack.

Part 2 -- ``pong_subproc.npz``: the reference's ``SubprocVecEnv`` (``_worker``, ``_flatten_obs``;
utils/subproc_vec_env.py:11-47,81-118,188-222) over the same envs, one forked worker per env, pipes
and pickles as in the reference.  Recorded: shapes and dtypes of every return, values, the
``terminal_observation`` convention of the worker (done is the env's scalar), infos as a tuple.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_standins as S  # noqa: E402
from gen_pong_wrapped_golden import Router, make_cv2  # noqa: E402


def serve_arrays(streams):
    L = max(len(s.u) for s in streams)
    n = len(streams)
    u, bx, by = np.zeros((n, L)), np.zeros((n, L), np.uint8), np.zeros((n, L), np.uint8)
    nd = np.array([len(s.u) for s in streams])
    for i, s in enumerate(streams):
        u[i, :nd[i]], bx[i, :nd[i]], by[i, :nd[i]] = s.u, s.bx, s.by
    return u, bx, by, nd


def main():
    import torch

    S.install(cv2_module=make_cv2())
    import gym

    pong = S.load_ref("competitive_rl.pong.base_pong_env", "pong/base_pong_env.py")
    S.load_ref("competitive_rl.pong.register", "pong/register.py").register_pong()
    aw = S.load_ref("competitive_rl.utils.atari_wrappers", "utils/atari_wrappers.py")
    S.load_ref("competitive_rl.utils.vec_env_utils", "utils/vec_env_utils.py")
    S.load_ref("competitive_rl.utils.base_vec_env", "utils/base_vec_env.py")
    dv = S.load_ref("competitive_rl.utils.dummy_vec_env", "utils/dummy_vec_env.py")
    sv = S.load_ref("competitive_rl.utils.subproc_vec_env", "utils/subproc_vec_env.py")
    uu = S.load_ref("competitive_rl.utils.utils", "utils/utils.py")

    # ------------------------------------------------------------------ part 1: step_envs + FrameStackTensor
    N, T, R, K = 4, 700, 42, 4
    router = Router(N, 4100)
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
    fst = uu.FrameStackTensor(N, (1, R, R), K, "cpu")
    obs0 = venv.reset()
    fst.update(obs0[0])
    stack0 = fst.get().numpy().copy()
    acts = np.random.RandomState(5).randint(0, 3, (T, N, 2))
    acts[np.random.RandomState(6).random_sample((T, N, 2)) < 0.04] = 999
    episode_rewards = np.zeros((N, 2), np.float64)
    reward_recorder, length_recorder = [], []
    total_steps, total_episodes = 0, 0
    out = dict(masks=[], done=[], total_steps=[], total_episodes=[], episode_rewards=[], stack=[], rec_count=[])
    for t in range(T):
        ret = uu.step_envs(acts[t], venv, episode_rewards, fst, reward_recorder, length_recorder, total_steps, total_episodes, "cpu", False)
        obs, reward, done, info, masks, total_episodes, total_steps, episode_rewards = ret
        assert done.shape == (N,) and masks.shape == (N, 1) and masks.dtype == torch.float32
        st = fst.get().numpy()
        assert np.array_equal(st, np.round(st)) and st.min() >= 0 and st.max() <= 255
        out["masks"].append(masks.numpy()[:, 0].copy()), out["done"].append(done.copy())
        out["total_steps"].append(total_steps), out["total_episodes"].append(total_episodes)
        out["episode_rewards"].append(episode_rewards.copy()), out["stack"].append(st.astype(np.uint8))
        out["rec_count"].append(len(reward_recorder))
    u, bx, by, nd = serve_arrays(router.streams)
    np.savez_compressed(
        os.path.join(HERE, "step_envs.npz"), acts=acts.astype(np.int32), draw_u=u, draw_bx=bx, draw_by=by, ndraws=nd, resized_dim=R,
        frame_stack=K, stack0=stack0.astype(np.uint8), masks=np.array(out["masks"]), done=np.array(out["done"]),
        total_steps=np.array(out["total_steps"]), total_episodes=np.array(out["total_episodes"]),
        episode_rewards=np.array(out["episode_rewards"]), stack=np.array(out["stack"]), rec_count=np.array(out["rec_count"]),
        reward_recorder=np.array(reward_recorder), length_recorder=np.array(length_recorder))
    print("step_envs: steps", T, "episodes", total_episodes, "recorded returns", np.array(reward_recorder)[:3].tolist(),
          "lengths", length_recorder[:5])

    # ------------------------------------------------------------------ part 2: SubprocVecEnv
    N2, T2 = 3, 360

    def sthunk(i):
        inner = aw.make_env_a2c_atari("cPongDouble-v0", 0, i, None, R, None)

        def f():  # runs in the forked worker: the worker's process-global `random` becomes env i's recorded stream
            pong.random = S.ServeStream(7300 + 31 * i)
            return inner()

        return f

    senv = sv.SubprocVecEnv([sthunk(i) for i in range(N2)], start_method="fork")
    o0 = senv.reset()
    assert isinstance(o0, tuple) and len(o0) == 2
    acts2 = np.random.RandomState(8).randint(0, 3, (T2, N2, 2))
    acts2[np.random.RandomState(9).random_sample((T2, N2, 2)) < 0.04] = 999
    rec = dict(obs=[], rew=[], done=[], real=[], nsteps=[], term_t=[], term_i=[], term_obs=[])
    meta = {}
    for t in range(T2):
        o, r, d, infos = senv.step(acts2[t])
        if t == 0:
            meta = dict(obs_type=type(o).__name__, obs_dtype=str(o[0].dtype), obs_shape=np.array(o[0].shape), rew_dtype=str(r.dtype),
                        rew_shape=np.array(r.shape), done_dtype=str(d.dtype), done_shape=np.array(d.shape), infos_type=type(infos).__name__,
                        reset_dtype=str(o0[0].dtype))
        assert isinstance(infos, tuple) and len(infos) == N2 and d.shape == (N2,)
        rec["obs"].append(np.stack([o[0][:, 0], o[1][:, 0]], 1)), rec["rew"].append(r), rec["done"].append(d)
        rec["real"].append([infos[i]["real_reward"] for i in range(N2)]), rec["nsteps"].append([infos[i]["num_steps"] for i in range(N2)])
        for i in range(N2):
            if "terminal_observation" in infos[i]:
                to = infos[i]["terminal_observation"]
                assert isinstance(to, tuple) and d[i]
                rec["term_t"].append(t), rec["term_i"].append(i), rec["term_obs"].append(np.stack([to[0][0], to[1][0]]))
            else:
                assert not d[i]
    senv.close()
    # the workers' serve streams are deterministic functions of their seeds: regenerate them here
    streams = []
    nd2 = []
    for i in range(N2):
        s = S.ServeStream(7300 + 31 * i)
        for _ in range(2 + 400):  # Ball.__init__ + PongGame.__init__ draws come first (dropped below), then a long tail
            s.uniform(0, 1), s.choice([0, 1]), s.choice([0, 1])
        s.u, s.bx, s.by = s.u[2:], s.bx[2:], s.by[2:]
        streams.append(s)
    u2, bx2, by2, _ = serve_arrays(streams)
    obs = np.array(rec["obs"])
    assert np.array_equal(obs, np.round(obs))
    np.savez_compressed(
        os.path.join(HERE, "pong_subproc.npz"), acts=acts2.astype(np.int32), draw_u=u2, draw_bx=bx2, draw_by=by2, resized_dim=R,
        obs0=np.stack([o0[0][:, 0], o0[1][:, 0]], 1).astype(np.uint8), obs=obs.astype(np.uint8), rew=np.array(rec["rew"]),
        done=np.array(rec["done"]), real_reward=np.array(rec["real"], np.float32), num_steps=np.array(rec["nsteps"], np.int32),
        term_t=np.array(rec["term_t"]), term_i=np.array(rec["term_i"]), term_obs=np.array(rec["term_obs"]).astype(np.uint8),
        **{f"meta_{k}": np.array(v) for k, v in meta.items()})
    print("subproc: steps", T2, "dones", int(np.array(rec["done"]).sum()), "terminal obs", len(rec["term_t"]), "meta",
          {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in meta.items()})


if __name__ == "__main__":
    main()
