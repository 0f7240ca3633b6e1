"""Golden vectors for the CarRacing wrapper chains and their VecEnv conventions.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_car_wrappers_golden.py

The reference's own code is run end to end above the simulator:

* ``make_car_racing_double(seed, rank, frame_stack, action_repeat)`` (car_racing/register.py:43-53):
  ``gym.make`` (+ TimeLimit 1000) -> ``MultipleFrameStack`` -> ``FlattenMultiAgentObservation`` ->
  ``WrapPyTorch`` (utils/atari_wrappers.py:262-334,12-37) under ``DummyVecEnv``;
* ``make_car_racing`` (register.py:29-40): ``FrameStack`` -> ``WrapPyTorch`` for cCarRacing-v0;
* ``make_competitive_car_racing`` (car_racing/make_competitive_car_racing.py:10-58): ``MultipleFrameStack`` ->
  ``WrapPyTorch`` -> ``CarRacingWrapper(opponent_policy)`` under ``DummyVecEnv``.

The simulator underneath is a SCRIPTED stub registered under the reference's env ids: its frames are
symbolic -- a 96x96x1 array whose first four bytes name (env rank, episode, step, agent) -- its rewards
are exact binary fractions of the same ids and its per-car done flags follow a plan.  So what the fixture
pins is what the wrappers DO with frames / rewards / dones / infos (which frame lands in which plane, the
reset fill, agent-0 reward, any-done vs d[0], TimeLimit, terminal observations, which action reaches which
car), independent of pixels.  ``car_wrappers.npz``: per configuration, decoded frame ids of every
observation plane, rewards, dones, info fields, and the actions the stub received.
"""
import os
import sys
import types
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_standins as S  # noqa: E402

def frame(rank, episode, t, agent):
    f = np.zeros((96, 96, 1), np.uint8)
    f[0, :4, 0] = (rank, episode, t & 255, (t >> 8) * 4 + agent)
    return f


def decode(obs_chw):
    """(planes, 96, 96) -> (planes, 4) ids"""
    return np.asarray(obs_chw)[:, 0, :4].astype(np.uint8)


def reward_of(rank, episode, t, agent):
    return (1 if agent == 0 else -1) * ((rank + 1) * 64 + episode * 8 + t / 16.0)


# per env rank: episodes as (step at which car 0 reports done, same for car 1); 0 = never (TimeLimit ends it)
PLANS = {
    0: [(7, 0), (0, 5), (9, 9), (12, 0)] + [(11, 0)] * 200,
    1: [(0, 6), (30, 0), (0, 0), (4, 8)] + [(13, 17)] * 200,
    2: [(0, 0), (3, 0)] + [(21, 0)] * 200,
}


class Stub:
    """Scripted stand-in for CarRacing(num_player=P): the gym.Env surface the wrappers use."""

    metadata = {}
    instances = []

    def __init__(self, num_player=1, verbose=0, action_repeat=None, **kw):
        spaces = sys.modules["gym.spaces"]
        self.P = num_player
        self.repeat = 1 if action_repeat is None else action_repeat
        self.action_space = spaces.Box(-1, 1, (2,), dtype=np.float32)
        self.observation_space = spaces.Box(0, 255, (96, 96, 1), dtype=np.uint8)
        if num_player > 1:
            self.action_space = spaces.Dict({i: self.action_space for i in range(num_player)})
        self.rank = None
        self.episode = -1
        self.received = []
        Stub.instances.append(self)

    unwrapped = property(lambda self: self)

    def seed(self, seed=None):
        self.rank = seed  # make_car_racing*: env.seed(seed + rank) with seed = 0
        return [seed]

    def close(self):
        pass

    def reset(self):
        self.episode += 1
        self.t = 0
        self.step_count = 0
        self.done = {k: False for k in range(self.P)}
        o = {k: frame(self.rank, self.episode, 0, k) for k in range(self.P)}
        return o if self.P > 1 else o[0]

    def step(self, action):
        self.t += 1
        self.step_count += self.repeat
        if self.P > 1:
            assert isinstance(action, dict) and sorted(action) == list(range(self.P))
            self.received.append([np.asarray(action[k], np.float64).reshape(-1)[:2] for k in range(self.P)])
        else:
            self.received.append([np.asarray(action, np.float64).reshape(-1)[:2]])
        plan = PLANS[self.rank][self.episode]
        for k in range(self.P):
            if plan[k] and self.t >= plan[k]:
                self.done[k] = True
        o = {k: frame(self.rank, self.episode, self.t, k) for k in range(self.P)}
        r = {k: reward_of(self.rank, self.episode, self.t, k) for k in range(self.P)}
        if self.P == 1:
            return o[0], r[0], self.done[0], {"num_steps": self.step_count}
        return o, r, dict(self.done), {k: {"num_steps": self.step_count} for k in range(self.P)}


def run():
    S.install()
    gym = sys.modules["gym"]
    spaces = sys.modules["gym.spaces"]
    spaces.Dict.__getitem__ = lambda self, k: self.spaces[k]
    cv2 = types.ModuleType("cv2")
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda *_: None)
    sys.modules["cv2"] = cv2
    aw = S.load_ref("competitive_rl.utils.atari_wrappers", "utils/atari_wrappers.py")
    S.load_ref("competitive_rl.utils.vec_env_utils", "utils/vec_env_utils.py")
    S.load_ref("competitive_rl.utils.base_vec_env", "utils/base_vec_env.py")
    dv = S.load_ref("competitive_rl.utils.dummy_vec_env", "utils/dummy_vec_env.py")
    sv = S.load_ref("competitive_rl.utils.subproc_vec_env", "utils/subproc_vec_env.py")
    # the simulator module is replaced wholesale by the stub; car_racing/register.py imports CarRacing from it
    crmp = types.ModuleType("competitive_rl.car_racing.car_racing_multi_players")
    crmp.CarRacing = Stub
    sys.modules["competitive_rl.car_racing"] = types.ModuleType("competitive_rl.car_racing")
    sys.modules["competitive_rl.car_racing"].__path__ = []
    sys.modules["competitive_rl.car_racing.car_racing_multi_players"] = crmp
    reg = S.load_ref("competitive_rl.car_racing.register", "car_racing/register.py")
    reg.register_car_racing()
    sys.modules["competitive_rl.utils"].DummyVecEnv, sys.modules["competitive_rl.utils"].SubprocVecEnv = dv.DummyVecEnv, sv.SubprocVecEnv
    regall = types.ModuleType("competitive_rl.register")
    regall.register_competitive_envs = lambda: None  # done above with the reference's own register_car_racing
    sys.modules["competitive_rl.register"] = regall
    mc = S.load_ref("competitive_rl.car_racing.make_competitive_car_racing", "car_racing/make_competitive_car_racing.py")

    out = {}

    def record(name, venv, T, N, act_shape, planes, stubs_before):
        rs = np.random.RandomState(zlib.crc32(name.encode()) % 100000)
        o0 = venv.reset()
        res = dict(obs0=np.stack([decode(o0[i]) for i in range(N)]), obs=[], rew=[], done=[], num_steps=[], info_reward=[], truncated=[],
                   term_t=[], term_i=[], term_obs=[], acts=[])
        meta = None
        for t in range(T):
            a = np.round(rs.uniform(-1, 1, (N,) + act_shape) * 64) / 64  # exact in float32
            o, r, d, infos = venv.step(a)
            if meta is None:
                meta = dict(obs_shape=np.array(o.shape), obs_dtype=str(o.dtype), rew_shape=np.array(r.shape), rew_dtype=str(r.dtype),
                            done_shape=np.array(d.shape), done_dtype=str(d.dtype), infos_type=type(infos).__name__)
            res["acts"].append(a)
            res["obs"].append(np.stack([decode(o[i]) for i in range(N)]))
            res["rew"].append(np.asarray(r, np.float64).reshape(N, -1)), res["done"].append(np.asarray(d).reshape(N, -1))
            ns, ir, tr = [], [], []
            for i in range(N):
                inf = infos[i]
                if "num_steps" in inf:  # single car / CarRacingWrapper: info = {"num_steps": k}
                    ns.append(inf["num_steps"]), ir.append([np.nan, np.nan])
                    assert set(inf) <= {"num_steps", "terminal_observation", "TimeLimit.truncated"}, set(inf)
                else:
                    ns.append(inf[0]["num_steps"]), ir.append([inf[0]["reward"], inf[1]["reward"]])
                    assert inf[1]["num_steps"] == inf[0]["num_steps"] and set(inf[0]) == {"num_steps", "reward"}
                    assert set(inf) <= {0, 1, "terminal_observation", "TimeLimit.truncated"}, set(inf)
                tr.append(int(inf["TimeLimit.truncated"]) if "TimeLimit.truncated" in inf else -1)
                if "terminal_observation" in inf:
                    res["term_t"].append(t), res["term_i"].append(i), res["term_obs"].append(decode(inf["terminal_observation"]))
            res["num_steps"].append(ns), res["info_reward"].append(ir), res["truncated"].append(tr)
        stubs = Stub.instances[stubs_before:]
        assert len(stubs) == N and [s.rank for s in stubs] == list(range(N))
        L = min(len(s.received) for s in stubs)
        assert L == T
        res["received"] = np.array([[np.stack(s.received[t]) for s in stubs] for t in range(T)])  # (T, N, P, 2)
        for k, v in res.items():
            out[f"{name}/{k}"] = np.asarray(v)
        for k, v in meta.items():
            out[f"{name}/meta_{k}"] = np.asarray(v)
        print(name, "steps", T, "dones", int(np.asarray(res["done"]).sum()), "terminal obs", len(res["term_t"]),
              {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in meta.items()})
        venv.close()

    N = 3
    for K in (None, 4):
        for rep in (None, 2):
            if K is None and rep == 2:
                continue
            nb = len(Stub.instances)
            venv = dv.DummyVecEnv([reg.make_car_racing_double(0, i, frame_stack=K, action_repeat=rep) for i in range(N)])
            record(f"double_k{K or 0}_r{rep or 1}", venv, 1060 if (K, rep) == (4, None) else 80, N, (2, 2), 2 * (K or 1), nb)
    for K in (None, 4):
        nb = len(Stub.instances)
        venv = dv.DummyVecEnv([reg.make_car_racing("cCarRacing-v0", 0, i, frame_stack=K, action_repeat=None) for i in range(N)])
        record(f"single_k{K or 0}", venv, 1060 if K else 80, N, (2,), K or 1, nb)

    # make_competitive_car_racing: the opponent acts on ITS observation of the step / reset before; its action is a
    # function of the newest frame's id, so the fixture shows which observation it saw
    def opponent(o1):
        ids = decode(o1)[-1].astype(np.float64)
        return np.array([ids[1] / 16 - ids[2] / 64, ids[2] / 128 - 0.5])

    nb = len(Stub.instances)
    venv = mc.make_competitive_car_racing(opponent, seed=0, num_envs=N, asynchronous=False, frame_stack=4)
    record("competitive_k4", venv, 1060, N, (2,), 4, nb)

    out["plans"] = np.array([PLANS[i][:200] for i in range(N)])
    np.savez_compressed(os.path.join(HERE, "car_wrappers.npz"), **out)


if __name__ == "__main__":
    run()
