"""CPU baselines in the REFERENCE'S ARCHITECTURE, for bench.py's ``cpu_baseline`` leg only.  TEST/BENCH INFRASTRUCTURE:
nothing in the product imports this file.

The reference's fast path on a CPU is ``SubprocVecEnv`` (competitive_rl/utils/subproc_vec_env.py:11-118): one daemon
process per env, a duplex pipe each, ``('step', action)`` pickled down and the pickled ``(obs, reward, done, info)``
back up every step, ``_flatten_obs`` = ``np.stack`` in the parent; its slow path is ``DummyVecEnv``
(utils/dummy_vec_env.py:51-63): a Python loop over the envs in one process.  The reference's own env code cannot run
on the GPU box (pygame / gym / cv2 / Box2D are not installable), so each worker steps ONE env of the oracle -- the C
restatement of the same per-env work (full 210x160 raster, max, gray, INTER_AREA; Box2D-style solve + two 96x96 renders)
-- inside that process/pipe architecture.  What is timed is therefore the architecture + a compiled per-env step: an
upper bound on what the reference's Python envs would reach on the same cores.
"""
import multiprocessing as mp
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _PongEnv:
    """One env with the per-env wrapper outputs of make_env_a2c_atari / the raw env."""

    def __init__(self, kind, rank):
        from oracle import pong_oracle as po

        atlas = np.load(os.path.join(ROOT, "competitive_rl_amd", "assets", "pong_score_atlas.npz"))["atlas"]
        if kind == "raw":
            self.o = po.PongOracle(1, atlas, obs_mode=po.RAW, seed=0, env_id_base=rank)
        else:
            self.o = po.PongOracle(1, atlas, obs_mode=po.GRAY, resized_dim=int(kind[5:]), frame_stack=1, seed=0, env_id_base=rank)
        self.o.set_threads(1)

    def reset(self):
        obs = self.o.reset()
        return (obs[0, 0].copy(), obs[0, 1].copy())

    def step(self, action):
        # the oracle auto-resets like the VecEnv does; hand back what the worker protocol needs
        obs, rew, done = self.o.step(np.asarray(action, np.int32).reshape(1, 2))
        info = {"real_reward": self.o.real_reward[0].tolist(), "num_steps": int(self.o.num_steps[0])}
        if done[0]:
            info["terminal_observation"] = tuple(self.o.terminal_observation(0))
        return (obs[0, 0].copy(), obs[0, 1].copy()), rew[0].copy(), bool(done[0]), info

    def sample_action(self, rs):
        return rs.randint(0, 3, 2)


class _CarEnv:
    def __init__(self, kind, rank):
        import ctypes as C

        from oracle import car_oracle as co

        self.co = co
        self.rs = np.random.RandomState(1000 + rank)
        self.e = co.CarEnv()
        self.reset()

    def _obs(self):
        # get_observation per viewer: the nearest-neighbour rotated crop of the episode's pre-rastered map (built once per
        # reset, as _render_road does) plus the overlays
        return np.stack([self.e.render(v) for v in range(2)])

    def reset(self):
        while self.e.reset(self.rs.random_sample(24 * 8), int(self.rs.randint(2))) < 0:
            pass
        self.e.e["contacts_enabled"] = 1
        self.e.step(None)
        self.t = 0
        return self._obs()

    def step(self, action):
        rew, done = self.e.step(np.asarray(action, np.float64).reshape(2, 2))
        self.t += 1
        d = bool(done.any()) or self.t >= 1000
        obs = self._obs()
        info = {0: {"num_steps": self.t, "reward": float(rew[0])}, 1: {"num_steps": self.t, "reward": float(rew[1])}}
        if d:
            info["terminal_observation"] = obs
            obs = self.reset()
        return obs, float(rew[0]), d, info

    def sample_action(self, rs):
        return rs.uniform(-1, 1, (2, 2))


def make_env(kind, rank):
    return _CarEnv(kind, rank) if kind == "car" else _PongEnv(kind, rank)


def _worker(remote, parent_remote, kind, rank):
    """The reference's worker loop (subproc_vec_env.py:11-47), commands 'step' / 'reset' / 'close'."""
    parent_remote.close()
    env = make_env(kind, rank)
    try:
        while True:
            cmd, data = remote.recv()
            if cmd == "step":
                remote.send(env.step(data))
            elif cmd == "reset":
                remote.send(env.reset())
            elif cmd == "close":
                remote.close()
                break
    except EOFError:
        pass


def time_subproc(kind, num_envs, budget_s, start_method="forkserver"):
    """env-steps/s of `num_envs` one-env worker processes stepped in lock-step through pipes (SubprocVecEnv.step)."""
    from oracle import pong_oracle as _po
    _po.build()  # once, here: the workers only load the libraries
    ctx = mp.get_context(start_method)
    remotes, work_remotes = zip(*[ctx.Pipe(duplex=True) for _ in range(num_envs)])
    procs = []
    for rank, (wr, r) in enumerate(zip(work_remotes, remotes)):
        p = ctx.Process(target=_worker, args=(wr, r, kind, rank), daemon=True)
        p.start()
        procs.append(p)
        wr.close()
    for r in remotes:
        r.send(("reset", None))
    obs = [r.recv() for r in remotes]
    rs = np.random.RandomState(0)
    probe = make_env(kind, 10 ** 6)

    def step_all():
        acts = [probe.sample_action(rs) for _ in range(num_envs)]
        for r, a in zip(remotes, acts):       # step_async
            r.send(("step", a))
        results = [r.recv() for r in remotes]  # step_wait
        o, rew, done, infos = zip(*results)
        if isinstance(o[0], tuple):            # _flatten_obs for a Tuple space
            flat = tuple(np.stack([x[i] for x in o]) for i in range(len(o[0])))
        else:
            flat = np.stack(o)
        return flat, np.stack(rew), np.stack(done), infos

    step_all()
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s:
        step_all()
        k += 1
    dt = time.perf_counter() - t0
    for r in remotes:
        r.send(("close", None))
    for p in procs:
        p.join(timeout=5)
    return num_envs * k / dt, k, dt


def time_dummy(kind, num_envs, steps):
    """env-steps/s of the DummyVecEnv loop: `num_envs` envs stepped one after the other in this process."""
    envs = [make_env(kind, i) for i in range(num_envs)]
    for e in envs:
        e.reset()
    rs = np.random.RandomState(0)
    t0 = time.perf_counter()
    for _ in range(steps):
        res = [e.step(e.sample_action(rs)) for e in envs]
        o, rew, done, infos = zip(*res)
        np.stack([x[0] for x in o]) if isinstance(o[0], tuple) else np.stack(o)
    dt = time.perf_counter() - t0
    return num_envs * steps / dt, steps, dt
