"""TournamentEnvWrapper (reference competitive_rl/pong/competitive_pong_env.py:9-53).

Wraps a cPongDouble VecEnv and plays a built-in opponent on the right side, so the caller
sees a single-agent env.  Opponents (pong/builtin_policies.py:26-33,61-91): RANDOM, WEAK, MEDIUM,
RULE_BASED.  WEAK / MEDIUM are the reference's LightActorCritic checkpoints served by one HIP kernel
(policy_serving.Policy: frame in, action out, straight into the right-hand column of the device
action array -- no host round trip).  RULE_BASED is action 999, resolved by the step kernel
(= auto_action).  STRONG is listed by the reference but its checkpoint is not in the reference tree
(the reference's own wrapper asserts on the missing file), so it is not offered here; its MODEL (the
full-size ActorCritic) is served -- ``Policy(..., use_light_model=False)`` -- and ``add_agent`` puts
such a policy into the pool.
"""
import random

import numpy as np
import torch

from . import spaces
from .policy_serving import BUILTIN_CHECKPOINTS, Policy
from .vec_env import CHEAT_CODES

BUILTIN_AGENT_NAMES = ["RANDOM", "WEAK", "MEDIUM", "RULE_BASED"]

# Hard-coded in the reference too (builtin_policies.py:36-37)
single_obs_space = spaces.Box(0, 255, (1, 42, 42))
single_act_space = spaces.Discrete(3)


def get_builtin_agent_names():
    return BUILTIN_AGENT_NAMES


def get_rule_based_policy(num_envs=1):
    """pong/builtin_policies.py:44-48: the cheat code for one env, a list of it for several (the step kernel resolves 999 = auto_action)."""
    if num_envs == 1:
        return lambda _: CHEAT_CODES
    return lambda _: [CHEAT_CODES] * num_envs


def get_random_policy(num_envs=1):
    """pong/builtin_policies.py:51-58 (numpy's global generator, as there)."""
    if num_envs == 1:
        return lambda obs: np.random.randint(3)
    return lambda obs: [np.random.randint(3) for _ in range(num_envs)]


def get_compute_action_function(agent_name, num_envs=1, device=None):
    """pong/builtin_policies.py:61-91."""
    if agent_name in BUILTIN_CHECKPOINTS:
        return Policy(single_obs_space, single_act_space, num_envs, BUILTIN_CHECKPOINTS[agent_name], use_light_model=True,
                      device=device)
    if agent_name == "RANDOM":
        return get_random_policy(num_envs)
    if agent_name == "RULE_BASED":
        return get_rule_based_policy(num_envs)
    raise ValueError("Unknown agent name: {}".format(agent_name))


class TournamentEnvWrapper:
    """Same protocol as the reference's wrapper (competitive_pong_env.py:9-53: step / reset / reset_opponent / get_agent_names /
    seed, a single-agent view of a two-player env); the opponent's action never leaves the device: both columns of one int32
    (N, 2) action tensor are filled in place and handed to the env's device step."""

    def __init__(self, env, num_envs, agent_names=None):
        self.env, self.num_envs = env, num_envs
        device = getattr(env, "device", None)
        if device is None:
            raise RuntimeError("TournamentEnvWrapper needs the HIP vector env (there is no CPU fallback)")
        names = get_builtin_agent_names() if agent_names is None else list(agent_names)
        cnn = [n for n in names if n in BUILTIN_CHECKPOINTS]
        if cnn and getattr(env, "R", 42) != 42:
            raise ValueError(f"{cnn} are trained on 42x42 frames (builtin_policies.py:36); make the env with resized_dim=42 "
                             "or pass agent_names without them")
        self.agents = {name: get_compute_action_function(name, num_envs, device) for name in names}
        self.agent_names = list(self.agents)
        self.observation_space, self.action_space = env.observation_space[0], env.action_space[0]
        self.prev_opponent_obs = None  # what the opponent acts on: the right-hand view of the previous step / reset
        self._act = torch.zeros((num_envs, 2), dtype=torch.int32, device=device)
        self._select("RULE_BASED" if "RULE_BASED" in self.agents else self.agent_names[0])  # the reference starts with RULE_BASED

    def _stack_env(self):
        """The env a FrameStackTensor binds to (frame_stack.py): agent 0's observation of the wrapped env is this wrapper's."""
        return getattr(self.env, "_stack_env", lambda: None)()

    def done_host(self):
        """The last step's done flags on the host, ahead of the observation (HipPongVecEnv.done_host)."""
        return self.env.done_host()

    # ---- opponent
    def _select(self, name):
        self.current_agent_name, self.current_agent = name, self.agents[name]

    def get_agent_names(self):
        return self.agent_names

    def reset_opponent(self, agent_name=None):
        name = random.choice(self.agent_names) if agent_name is None else agent_name
        assert name in self.agent_names, self.agent_names
        self._select(name)

    def add_agent(self, name, compute_action):
        """Beyond the reference: add an opponent of one's own to the pool -- a ``policy_serving.Policy`` (e.g. a full-size
        ActorCritic checkpoint trained with the reference, ``use_light_model=False``; its actions stay on the device) or any
        callable obs -> (N,) actions on the host."""
        assert name not in self.agents, name
        if isinstance(compute_action, Policy):
            assert compute_action.num_envs == self.num_envs and compute_action.device == self._act.device
        self.agents[name] = compute_action
        self.agent_names.append(name)

    def _fill_actions(self, mine_i32):
        """column 0 <- the caller's actions, column 1 <- the current opponent's: written in place by the policy kernel
        (WEAK / MEDIUM), the constant 999 (RULE_BASED, resolved by the step kernel = auto_action), or a host callable (RANDOM)"""
        self._act[:, 0] = mine_i32
        agent = self.current_agent
        if isinstance(agent, Policy):
            agent.act_device(self.prev_opponent_obs, out=self._act[:, 1])
        elif self.current_agent_name == "RULE_BASED":
            self._act[:, 1] = CHEAT_CODES
        else:
            theirs = np.asarray(agent(self.prev_opponent_obs)).reshape(-1)
            self._act[:, 1] = torch.as_tensor(theirs, dtype=torch.int32).to(self._act.device)
        return self._act

    # ---- VecEnv protocol, agent 0's view
    def step(self, action):
        if not isinstance(action, torch.Tensor):
            action = torch.as_tensor(np.asarray(action).reshape(-1), dtype=torch.int32)
        obs, rew, done, info = self.env.step(self._fill_actions(action.to(self._act.device, torch.int32).reshape(-1)))
        self.prev_opponent_obs = obs[1]
        done = done[:, 0] if done.ndim == 2 else done
        return obs[0], rew[:, 0].reshape(-1, 1), done.reshape(-1, 1), info

    def step_device(self, actions_i32):
        """Hot-loop entry (no host work, no clones, no sync): ``actions_i32`` is an int32 (N,) device tensor;
        returns the env's device buffers (obs (N, 2, K, R, R) -- view 0 is the caller's --, rewards (N, 2),
        done (N,)) like HipPongVecEnv.step_device."""
        buf, rew, done = self.env.step_device(self._fill_actions(actions_i32))
        self.prev_opponent_obs = buf[:, 1]
        return buf, rew, done

    def reset(self, **kwargs):
        views = self.env.reset(**kwargs)
        self.prev_opponent_obs = views[1]
        return views[0]

    def seed(self, s):
        self.env.seed(s)

    def close(self):
        for a in self.agents.values():
            if isinstance(a, Policy):
                a.close()
        self.env.close()
