"""Golden vectors for the WRAPPED path: the reference's own wrappers + DummyVecEnv.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_pong_wrapped_golden.py

Loads, by path, the reference's ``utils/atari_wrappers.py`` (MaxAndSkipEnv, WarpFrame,
ClipRewardEnv, WrapPyTorch, make_env_a2c_atari), ``utils/dummy_vec_env.py`` (+ base,
utils), ``pong/register.py`` and ``pong/base_pong_env.py`` with the stand-ins of
``_ref_standins.py`` and runs BASELINE config #1:
``DummyVecEnv([make_env_a2c_atari("cPongDouble-v0", 0, i, None, 42, None) for i in range(4)])``,
random actions.  Recorded: rewards, dones, info["real_reward"], info["num_steps"],
observations, terminal observations, and the per-env serve draws.

What is and is not pinned by this fixture:
* reward summing / sign clipping, done, auto-reset timing, num_steps, which frames feed
  the max, stale-buffer terminal observations: the REFERENCE's code, pinned.
* pixels: the stand-in ``cv2`` below is this build's definition -- luma of achromatic
  pixels = the value; INTER_AREA = exact rational area average, round-half-even to uint8.
  No score text (font stand-in).  With real gym/cv2 the step-path frames would be float32
  (MaxAndSkipEnv's buffer takes the Box dtype) and unrounded; see DESIGN.md "Deviations".
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_standins as S  # noqa: E402


def exact_area_resize(img, W, H):
    """Exact area average of a (h, w) integer image to (H, W), round-half-even."""
    h, w = img.shape

    def overlaps(src, dst):
        m = np.zeros((dst, src), np.int64)  # overlap length * dst (integers)
        for d in range(dst):
            lo, hi = d * src, (d + 1) * src  # in units of 1/dst source px
            for s in range(src):
                a, b = max(lo, s * dst), min(hi, (s + 1) * dst)
                if b > a:
                    m[d, s] = b - a
        return m

    oy, ox = overlaps(h, H), overlaps(w, W)
    num = oy @ img.astype(np.int64) @ ox.T  # sum S * oy * ox, scaled by H*W
    den = h * w                             # cell area (src px) * H * W
    q, r = np.divmod(num, den)
    up = (2 * r > den) | ((2 * r == den) & (q % 2 == 1))
    return (q + up).astype(np.uint8)


_cache = {}


def make_cv2():
    cv2 = types.ModuleType("cv2")
    cv2.COLOR_RGB2GRAY, cv2.INTER_AREA = 7, 3
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda *_: None)

    def cvtColor(frame, code):
        assert np.array_equal(frame[..., 0], frame[..., 1]) and np.array_equal(frame[..., 0], frame[..., 2])
        return frame[..., 0]

    def resize(frame, size, interpolation=None):
        assert interpolation == cv2.INTER_AREA
        f = np.asarray(frame)
        assert np.array_equal(f, np.round(f))
        return exact_area_resize(f.astype(np.int64), size[0], size[1])

    cv2.cvtColor, cv2.resize = cvtColor, resize
    return cv2


class Router:
    """Process-global ``random`` of the pong module, split into per-env recorded streams."""

    def __init__(self, n, seed):
        self.streams = [S.ServeStream(seed + 17 * i) for i in range(n)]
        self.cur = 0

    def uniform(self, a, b):
        return self.streams[self.cur].uniform(a, b)

    def choice(self, seq):
        return self.streams[self.cur].choice(seq)


def main():
    S.install(cv2_module=make_cv2())
    import gym

    pong = S.load_ref("competitive_rl.pong.base_pong_env", "pong/base_pong_env.py")
    S.load_ref("competitive_rl.pong.register", "pong/register.py").register_pong()
    aw = S.load_ref("competitive_rl.utils.atari_wrappers", "utils/atari_wrappers.py")
    S.load_ref("competitive_rl.utils.vec_env_utils", "utils/vec_env_utils.py")
    S.load_ref("competitive_rl.utils.base_vec_env", "utils/base_vec_env.py")
    dv = S.load_ref("competitive_rl.utils.dummy_vec_env", "utils/dummy_vec_env.py")

    N, T, R = 4, 1000, 42  # BASELINE config #1: num_envs=4, 1000 steps, resized_dim=42
    router = Router(N, 500)
    pong.random = router

    class Tag(gym.Wrapper):
        def __init__(self, env, idx):
            gym.Wrapper.__init__(self, env)
            self.idx = idx

        def step(self, a):
            router.cur = self.idx
            return self.env.step(a)

        def reset(self, **kw):
            router.cur = self.idx
            return self.env.reset(**kw)

    def thunk(i):
        inner = aw.make_env_a2c_atari("cPongDouble-v0", 0, i, None, R, None)

        def f():
            router.cur = i
            return Tag(inner(), i)

        return f

    venv = dv.DummyVecEnv([thunk(i) for i in range(N)])
    for s in router.streams:  # forget the construction draws (2 per env)
        s.u.clear(), s.bx.clear(), s.by.clear()
    obs0 = venv.reset()
    assert isinstance(obs0, tuple) and obs0[0].shape == (N, 1, R, R) and obs0[0].dtype == np.float32
    acts = np.random.RandomState(0).randint(0, 3, (T, N, 2))
    acts[np.random.RandomState(1).random_sample((T, N, 2)) < 0.05] = 999
    rews = np.zeros((T, N, 2), np.float32)
    dones = np.zeros((T, N, 2), bool)
    real = np.zeros((T, N, 2), np.float32)
    nsteps = np.zeros((T, N), np.int32)
    obs = np.zeros((T, N, 2, R, R), np.uint8)
    term_t, term_i, term_obs = [], [], []
    for t in range(T):
        o, r, d, info = venv.step(acts[t])
        assert r.dtype == np.float32 and d.shape == (N, 2) and o[0].dtype == np.float32
        rews[t], dones[t] = r, d
        for i in range(N):
            real[t, i] = info[i]["real_reward"]
            nsteps[t, i] = info[i]["num_steps"]
            if "terminal_observation" in info[i]:
                to = info[i]["terminal_observation"]
                term_t.append(t), term_i.append(i)
                term_obs.append(np.stack([np.asarray(to[0])[0], np.asarray(to[1])[0]]).astype(np.uint8))
        for k in range(2):
            assert np.array_equal(o[k], np.round(o[k]))
            obs[t, :, k] = o[k][:, 0].astype(np.uint8)
    L = max(len(s.u) for s in router.streams)
    du = np.zeros((N, L)), np.zeros((N, L), np.uint8), np.zeros((N, L), np.uint8)
    nd = np.array([len(s.u) for s in router.streams])
    for i, s in enumerate(router.streams):
        du[0][i, :nd[i]], du[1][i, :nd[i]], du[2][i, :nd[i]] = s.u, s.bx, s.by
    np.savez_compressed(
        os.path.join(HERE, "pong_wrapped.npz"), acts=acts.astype(np.int32), rew=rews, done=dones, real_reward=real,
        num_steps=nsteps, obs=obs, obs0=np.stack([obs0[0][:, 0], obs0[1][:, 0]], 1).astype(np.uint8),
        term_t=np.array(term_t), term_i=np.array(term_i), term_obs=np.stack(term_obs) if term_obs else np.zeros((0, 2, R, R), np.uint8),
        draw_u=du[0], draw_bx=du[1], draw_by=du[2], ndraws=nd, resized_dim=R)
    print("steps", T, "envs", N, "dones", int(dones[:, :, 0].sum()), "points", int((real != 0).any(-1).sum()),
          "terminal obs", len(term_t), "draws/env", nd.tolist())

    # ---------------------------------------------------------------- cPong-v0 (single player, FrameStack 4)
    N1, T1, K = 3, 700, 4
    router1 = Router(N1, 900)
    pong.random = router1

    def thunk1(i):
        inner = aw.make_env_a2c_atari("cPong-v0", 0, i, None, R, K)

        def f():
            router1.cur = i

            class Tag1(gym.Wrapper):
                def step(self, a):
                    router1.cur = i
                    return self.env.step(a)

                def reset(self, **kw):
                    router1.cur = i
                    return self.env.reset(**kw)

            return Tag1(inner())

        return f

    venv1 = dv.DummyVecEnv([thunk1(i) for i in range(N1)])
    for s_ in router1.streams:
        s_.u.clear(), s_.bx.clear(), s_.by.clear()
    o0 = venv1.reset()
    assert o0.shape == (N1, K, R, R)
    acts1 = np.random.RandomState(3).randint(0, 3, (T1, N1))
    rew1 = np.zeros((T1, N1, 1), np.float32)
    done1 = np.zeros((T1, N1, 1), bool)
    real1 = np.zeros((T1, N1), np.float32)
    ns1 = np.zeros((T1, N1), np.int32)
    obs1 = np.zeros((T1, N1, K, R, R), np.uint8)
    tt, ti, to = [], [], []
    for t in range(T1):
        o, r, d, info = venv1.step(acts1[t])
        assert r.shape == (N1, 1) and d.shape == (N1, 1)
        rew1[t], done1[t] = r, d
        assert np.array_equal(o, np.round(o))
        obs1[t] = o.astype(np.uint8)
        for i in range(N1):
            real1[t, i] = info[i]["real_reward"]
            ns1[t, i] = info[i]["num_steps"]
            if "terminal_observation" in info[i]:
                tt.append(t), ti.append(i), to.append(np.asarray(info[i]["terminal_observation"]).astype(np.uint8))
    L1 = max(len(s_.u) for s_ in router1.streams)
    du1 = np.zeros((N1, L1)), np.zeros((N1, L1), np.uint8), np.zeros((N1, L1), np.uint8)
    nd1 = np.array([len(s_.u) for s_ in router1.streams])
    for i, s_ in enumerate(router1.streams):
        du1[0][i, :nd1[i]], du1[1][i, :nd1[i]], du1[2][i, :nd1[i]] = s_.u, s_.bx, s_.by
    np.savez_compressed(
        os.path.join(HERE, "pong_single_wrapped.npz"), acts=acts1.astype(np.int32), rew=rew1, done=done1, real_reward=real1,
        num_steps=ns1, obs=obs1, obs0=o0.astype(np.uint8), term_t=np.array(tt), term_i=np.array(ti),
        term_obs=np.stack(to) if to else np.zeros((0, K, R, R), np.uint8), draw_u=du1[0], draw_bx=du1[1], draw_by=du1[2],
        ndraws=nd1, resized_dim=R, frame_stack=K)
    print("single: steps", T1, "envs", N1, "dones", int(done1.sum()), "terminal obs", len(tt), "draws/env", nd1.tolist())


if __name__ == "__main__":
    main()
