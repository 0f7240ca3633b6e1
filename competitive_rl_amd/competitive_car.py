"""``make_competitive_car_racing`` (reference competitive_rl/car_racing/make_competitive_car_racing.py:10-58).

The reference wraps every cCarRacingDouble-v0 env in ``MultipleFrameStack -> WrapPyTorch ->
CarRacingWrapper(opponent_policy)`` and puts N of them under a DummyVecEnv: the learner drives car 0,
``opponent_policy(o[1])`` -- evaluated on the observation the step (or the reset) just returned -- drives
car 1 on the NEXT step, and only agent 0's observation / reward / done / info come out -- ``d[0]``: an episode ends when
car 0 is done or gym's TimeLimit hits; an opponent that finishes first just stays in the world, frozen
(``done_policy="car0"`` of the car context; pinned by tests/golden/car_wrappers.npz "competitive_k4").

Here the N envs are one HipCarVecEnv batch.  The opponent's actions stay on the device when
``batched=True`` (one call with the (N, K, 96, 96) tensor, returning (N, 2)); ``batched=False`` keeps the
reference's per-env call protocol (numpy (K, 96, 96) in, one action out), which costs a host round trip
per step.  As in the reference (make_competitive_car_racing.py:53), ``action_repeat`` is accepted but not
forwarded to the envs.
"""
import numpy as np
import torch

from . import spaces
from .vec_env import VecEnv, _EnvList
from .vec_env_car import HipCarVecEnv

__all__ = ["make_competitive_car_racing", "HipCompetitiveCarVecEnv"]


class _Agent0Infos:
    """``infos[i]`` = the base env's ``info[0]``: {"num_steps": k} (+ agent 0's terminal observation)."""

    def __init__(self, inner, k):
        self._inner, self._k = inner, k

    def __len__(self):
        return len(self._inner)

    def __getitem__(self, i):
        d = self._inner[i]
        out = {"num_steps": d[0]["num_steps"]}
        if "terminal_observation" in d:
            out["terminal_observation"] = d["terminal_observation"][:self._k]
        return out

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def copy(self):
        return self


class HipCompetitiveCarVecEnv(VecEnv):
    def __init__(self, opponent_policy, num_envs, seed=0, frame_stack=4, dones="dummy", batched=False, device=None, output="torch",
                 env_id_base=0):
        assert callable(opponent_policy)
        self.opponent_policy, self.batched = opponent_policy, bool(batched)
        self.env = HipCarVecEnv(num_envs, seed=seed, device=device, env_id_base=env_id_base, output="torch", dones=dones,
                                action_repeat=None, frame_stack=frame_stack, players=2, done_policy="car0")
        self.K, self.output, self.dones_kind = self.env.K, output, dones
        self.device = self.env.device
        VecEnv.__init__(self, num_envs, spaces.Box(0, 255, (self.K, 96, 96), dtype=np.uint8),
                        spaces.Box(-1, 1, (2,), dtype=np.float32))
        self._act = torch.zeros((num_envs, 2, 2), dtype=torch.float32, device=self.device)
        self.opponent_action = None
        self.envs = _EnvList(self)

    def _opponent(self, obs):
        theirs = obs[:, self.K:]
        if self.batched:
            a = self.opponent_policy(theirs)
            a = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, dtype=np.float32))
        else:  # the reference's protocol: one env's (K, 96, 96) array in, one action out
            host = theirs.cpu().numpy()
            a = torch.as_tensor(np.asarray([self.opponent_policy(host[i]) for i in range(self.num_envs)], dtype=np.float32))
        self.opponent_action = a.to(self.device, torch.float32).reshape(self.num_envs, -1)[:, :2]

    def _out(self, t):
        return t.cpu().numpy() if self.output == "numpy" else t

    def reset(self):
        obs = self.env.reset()
        self._opponent(obs)
        return self._out(obs[:, :self.K])

    def step_async(self, actions):
        a = actions if isinstance(actions, torch.Tensor) else torch.as_tensor(np.asarray(actions, dtype=np.float32))
        self._act[:, 0] = a.to(self.device, torch.float32).reshape(self.num_envs, -1)[:, :2]
        self._act[:, 1] = self.opponent_action

    def step_wait(self):
        self.env.step_async(self._act)
        obs, rew, done, infos = self.env.step_wait()
        self._opponent(obs)  # for envs that just ended this is the reset observation, as under DummyVecEnv
        return self._out(obs[:, :self.K]), self._out(rew), self._out(done), _Agent0Infos(infos, self.K)

    def seed(self, seed=None):
        return self.env.seed(seed)

    def close(self):
        self.env.close()

    def get_attr(self, attr_name, indices=None):
        return [getattr(self, attr_name)] * len(self._get_indices(indices))

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        raise NotImplementedError("per-env methods are not exposed by the GPU batch")

    def get_images(self, *a, **k):
        return self.env.get_images(*a, **k)


def make_competitive_car_racing(opponent_policy, seed=0, num_envs=3, asynchronous=False, frame_stack=4, action_repeat=None, *,
                                batched=False, device=None, output="torch", env_id_base=0):
    assert callable(opponent_policy)
    asynchronous = asynchronous and num_envs > 1
    return HipCompetitiveCarVecEnv(opponent_policy, num_envs, seed=seed, frame_stack=frame_stack,
                                   dones="subproc" if asynchronous else "dummy", batched=batched, device=device, output=output,
                                   env_id_base=env_id_base)
