"""CPU restatement of the CarRacing wrapper chains and their DummyVecEnv conventions.  TEST INFRASTRUCTURE ONLY
(imported by tests/ alone; the product never touches oracle/).

Restates, for any base env with CarRacing's dict protocol (``reset() -> {k: frame}``,
``step({k: action}) -> ({k: frame}, {k: reward}, {k: done}, {k: info})``; scalars for one player):

* gym ``TimeLimit(max_episode_steps=1000)`` as ``gym.make`` applies it (car_racing/register.py:15-26) [gym from memory];
* ``MultipleFrameStack`` / ``FrameStack`` (utils/atari_wrappers.py:262-305, 222-259): per-agent deque of K frames,
  a reset fills all K with the first frame;
* ``FlattenMultiAgentObservation`` (:308-334): agents concatenated on the channel axis, agent 0's reward,
  ``done = any``, per-agent reward copied into ``info[k]["reward"]``;
* ``CarRacingWrapper`` of ``make_competitive_car_racing`` (car_racing/make_competitive_car_racing.py:17-38): the learner drives
  car 0, ``opponent_policy(o[1])`` of the previous step / reset drives car 1, returns agent 0's obs / reward / ``d[0]`` / info;
* ``WrapPyTorch`` (:12-37): HWC -> CHW;
* ``DummyVecEnv.step_wait`` / ``reset`` (utils/dummy_vec_env.py:51-75): (N, 1) float32 rewards, (N, 1) bool dones,
  ``terminal_observation`` + auto-reset when done.

Pinned against tests/golden/car_wrappers.npz (recorded from the reference's own wrappers over a scripted env).
"""
from collections import deque

import numpy as np


class _Env:
    """One wrapped env: TimeLimit -> stack -> (flatten | competitive) -> CHW."""

    def __init__(self, base, players, frame_stack, mode, opponent_policy=None, max_episode_steps=1000):
        assert mode in ("double", "single", "competitive")
        self.base, self.P, self.K, self.mode = base, players, frame_stack, mode
        self.limit = max_episode_steps
        self.opponent_policy, self.opponent_action = opponent_policy, None
        self.frames = {k: deque([], maxlen=frame_stack or 1) for k in range(players)}

    def _stacked(self, k):  # (H, W, K) -> CHW
        return np.stack([f[..., 0] for f in self.frames[k]], axis=0)

    def _obs(self, frames):
        for k in range(self.P):
            if self.K is None:
                self.frames[k].clear()
            self.frames[k].append(frames[k])
        if self.mode == "double":
            return np.concatenate([self._stacked(k) for k in range(self.P)], axis=0)
        return self._stacked(0) if self.mode == "single" else {k: self._stacked(k) for k in range(self.P)}

    def reset(self):
        self.elapsed = 0
        o = self.base.reset()
        o = o if isinstance(o, dict) else {0: o}
        for k in range(self.P):
            self.frames[k].clear()
            for _ in range((self.K or 1) - 1):
                self.frames[k].append(o[k])
        out = self._obs(o)
        if self.mode == "competitive":
            self.opponent_action = self.opponent_policy(out[1])
            return out[0]
        return out

    def step(self, action):
        if self.mode == "double":
            assert len(action) == self.P
            a = {k: action[k] for k in range(self.P)}
        elif self.mode == "competitive":
            a = {0: action, 1: self.opponent_action}
        else:
            a = action
        o, r, d, info = self.base.step(a)
        self.elapsed += 1
        if self.elapsed >= self.limit:  # TimeLimit
            info["TimeLimit.truncated"] = not d
            d = True
        o = o if isinstance(o, dict) else {0: o}
        out = self._obs(o)
        if self.mode == "single":
            return out, r, d, info
        if self.mode == "double":
            for k in r:
                info[k]["reward"] = r[k]
            if isinstance(d, dict):
                d = any(d.values())
            return out, r[0], d, info
        self.opponent_action = self.opponent_policy(out[1])
        if not isinstance(d, dict):
            d = {k: d for k in out}
        return out[0], r[0], d[0], info[0]


class CarDummyVecEnv:
    def __init__(self, bases, players=2, frame_stack=None, mode="double", opponent_policy=None):
        self.envs = [_Env(b, players, frame_stack, mode, opponent_policy) for b in bases]
        self.n = len(bases)

    def reset(self):
        return np.stack([e.reset() for e in self.envs])

    def step(self, actions):
        obs, rews, dones, infos = [], np.zeros((self.n, 1), np.float32), np.zeros((self.n, 1), bool), []
        for i, e in enumerate(self.envs):
            o, rews[i], dones[i], info = e.step(actions[i])
            if all(dones[i]):
                info["terminal_observation"] = o
                o = e.reset()
            obs.append(o), infos.append(info)
        return np.stack(obs), rews, dones, infos
