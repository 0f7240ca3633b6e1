"""competitive_rl_amd -- MI355X-native vector-env backend for competitive-rl's env.step() hot path.

Importing the package does not touch the GPU; constructing an env loads libcrl_hip.so and
fails loudly if it is missing (no CPU fallback exists in this package).
"""
from .make_envs import make_envs
from .vec_env import CHEAT_CODES, HipPongVecEnv, LazyInfos, VecEnv, VecEnvWrapper, tile_images
from .vec_env_car import HipCarVecEnv
from .frame_stack import FrameStackTensor
from .tournament import TournamentEnvWrapper
from .policy_serving import Policy
from .competitive_car import make_competitive_car_racing
from .utils import evaluate, step_envs
from .pong_evaluate import evaluate_two_policies, evaluate_two_policies_in_batch
from .sharding import ShardSpec, StepGather, all_gather_step, shard_of

__all__ = ["make_envs", "HipPongVecEnv", "HipCarVecEnv", "VecEnv", "VecEnvWrapper", "tile_images", "LazyInfos", "FrameStackTensor", "TournamentEnvWrapper", "Policy", "make_competitive_car_racing", "step_envs", "evaluate", "evaluate_two_policies", "evaluate_two_policies_in_batch", "CHEAT_CODES",
           "ShardSpec", "shard_of", "all_gather_step", "StepGather"]
