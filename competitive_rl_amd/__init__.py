"""competitive_rl_amd -- MI355X-native vector-env backend for competitive-rl's env.step() hot path.

Importing the package does not touch the GPU; constructing an env loads libcrl_hip.so and
fails loudly if it is missing (no CPU fallback exists in this package).
"""
from .make_envs import DummyVecEnv, EnvThunk, SubprocVecEnv, make_car_racing, make_car_racing_double, make_env_a2c_atari, make_envs
from .vec_env import CHEAT_CODES, HipPongVecEnv, LazyInfos, VecEnv, VecEnvWrapper, tile_images
from .vec_env_car import HipCarVecEnv
from .frame_stack import FrameStackTensor
from .tournament import (TournamentEnvWrapper, get_builtin_agent_names, get_compute_action_function, get_random_policy,
                         get_rule_based_policy)
from .policy_serving import Policy
from .competitive_car import make_competitive_car_racing
from .utils import evaluate, step_envs
from .pong_evaluate import evaluate_two_policies, evaluate_two_policies_in_batch
from .sharding import ShardSpec, StepGather, all_gather_step, shard_of



def register_pong():
    """pong/register.py:8-27 registers cPong-v0 / cPongDouble-v0 / cPongTournament-v0 with gym.  Here ``make_envs`` serves the ids itself
    (gym is not a dependency of the backend): kept so that a script written against the reference's package imports unchanged."""


def register_car_racing():
    """car_racing/register.py:8-26 (cCarRacing-v0 / cCarRacingDouble-v0, TimeLimit 1000): see ``register_pong``."""


def register_competitive_envs():
    """register.py:5-7."""
    register_pong()
    register_car_racing()


__all__ = ["make_envs", "DummyVecEnv", "SubprocVecEnv", "make_env_a2c_atari", "make_car_racing", "make_car_racing_double", "EnvThunk", "HipPongVecEnv", "HipCarVecEnv", "VecEnv", "VecEnvWrapper", "tile_images", "LazyInfos", "FrameStackTensor", "TournamentEnvWrapper", "Policy", "make_competitive_car_racing", "step_envs", "evaluate", "evaluate_two_policies", "evaluate_two_policies_in_batch", "CHEAT_CODES", "get_builtin_agent_names", "get_compute_action_function", "get_random_policy", "get_rule_based_policy",
           "register_pong", "register_car_racing", "register_competitive_envs",
           "ShardSpec", "shard_of", "all_gather_step", "StepGather"]
