"""TournamentEnvWrapper (reference competitive_rl/pong/competitive_pong_env.py:9-53).

Wraps a cPongDouble VecEnv and plays a built-in opponent on the right side, so the caller
sees a single-agent env.  Built-in opponents served here: RULE_BASED (action 999, resolved by
the kernel = auto_action) and RANDOM.  The reference's CNN opponents (WEAK / MEDIUM / STRONG)
are torch policies outside this hot path and are not shipped.
"""
import random

import numpy as np
import torch

from .vec_env import CHEAT_CODES

BUILTIN_AGENT_NAMES = ["RANDOM", "RULE_BASED"]


def get_compute_action_function(agent_name, num_envs=1):
    """pong/builtin_policies.py:61-91 for the two rule-free opponents."""
    if agent_name == "RANDOM":
        return lambda obs: np.random.randint(0, 3, size=num_envs)
    if agent_name == "RULE_BASED":
        return lambda obs: np.full(num_envs, CHEAT_CODES, dtype=np.int64)
    raise ValueError("Unknown agent name: {}".format(agent_name))


class TournamentEnvWrapper:
    def __init__(self, env, num_envs):
        self.env = env
        self.agents = {name: get_compute_action_function(name, num_envs) for name in BUILTIN_AGENT_NAMES}
        self.agent_names = list(self.agents)
        self.prev_opponent_obs = None
        self.current_agent_name = "RULE_BASED"
        self.current_agent = self.agents[self.current_agent_name]
        self.observation_space = env.observation_space[0]
        self.action_space = env.action_space[0]
        self.num_envs = num_envs

    def get_agent_names(self):
        return self.agent_names

    def reset_opponent(self, agent_name=None):
        if agent_name is None:
            self.current_agent_name = random.choice(self.agent_names)
        else:
            assert agent_name in self.agent_names, self.agent_names
            self.current_agent_name = agent_name
        self.current_agent = self.agents[self.current_agent_name]

    def step(self, action):
        if isinstance(action, torch.Tensor):
            action = action.detach().cpu().numpy()
        tuple_action = np.stack([np.asarray(action).reshape(-1), np.asarray(self.current_agent(self.prev_opponent_obs)).reshape(-1)],
                                axis=1)
        obs, rew, done, info = self.env.step(tuple_action)
        self.prev_opponent_obs = obs[1]
        if done.ndim == 2:
            done = done[:, 0]
        return obs[0], rew[:, 0].reshape(-1, 1), done.reshape(-1, 1), info

    def reset(self, **kwargs):
        obs = self.env.reset(**kwargs)
        self.prev_opponent_obs = obs[1]
        return obs[0]

    def seed(self, s):
        self.env.seed(s)

    def close(self):
        self.env.close()
