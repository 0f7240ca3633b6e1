"""``make_envs`` -- the drop-in boundary (reference competitive_rl/make_envs.py:67-118).

Same signature and argument meaning as the reference factory; the object returned for the
Pong ids is the HIP vector env instead of ``DummyVecEnv`` / ``SubprocVecEnv``.  Extra
keyword-only arguments select GPU-side options.
"""
import os

from .vec_env import HipPongVecEnv
from .vec_env_car import HipCarVecEnv

__all__ = ["make_envs"]

_HIP_IDS = ("cPongDouble-v0", "cPong-v0", "cPongTournament-v0", "cCarRacingDouble-v0", "cCarRacing-v0")


def make_envs(env_id="cPong-v0", seed=0, log_dir="data", num_envs=3, asynchronous=False, resized_dim=42,
              frame_stack=4, action_repeat=None, *, backend="hip", device=None, output="torch",
              obs_dtype="uint8", env_id_base=0, stack_planes=1, score_atlas=None):
    """Create a vectorised environment.

    :param env_id: one of the reference's ids served by the HIP backend: "cPongDouble-v0", "cPong-v0", "cPongTournament-v0",
        "cCarRacingDouble-v0", "cCarRacing-v0" (register.py / car_racing/register.py); anything else raises NotImplementedError.
    :param seed: random seed; env i is keyed by ``seed`` and its global index.
    :param log_dir: only created, as in the reference (make_envs.py:98-99).
    :param num_envs: number of concurrent environments (any size: one GPU lane per env).
    :param asynchronous: the reference picks SubprocVecEnv when True and num_envs > 1
        (make_envs.py:83,114-117).  On the GPU there are no worker processes; the flag only
        selects the Subproc return convention: ``dones`` of shape (N,) instead of (N, 2).
    :param resized_dim: observation is (1, resized_dim, resized_dim) per agent.
    :param frame_stack: must be None for cPongDouble-v0 (make_envs.py:105-106); FrameStack / MultipleFrameStack depth for the other ids.
    :param action_repeat: CarRacing only (car_racing_multi_players.py:576-603).
    :param stack_planes: GPU extra -- fuse FrameStackTensor's K-plane stack into the step
        (obs (N, K, R, R) per agent).
    :param score_atlas: GPU extra -- gray glyph images of the score band (default: the baked FreeSansBold atlas).
    """
    asynchronous = asynchronous and num_envs > 1
    if backend != "hip":
        raise ValueError("competitive_rl_amd only provides backend='hip'")
    if env_id not in _HIP_IDS:
        raise NotImplementedError(
            f"{env_id!r} is not served by the HIP backend yet (available: {_HIP_IDS})")
    if env_id == "cPongTournament-v0":  # make_envs.py:93-96
        from .tournament import TournamentEnvWrapper

        envs = make_envs("cPongDouble-v0", seed, log_dir, num_envs, asynchronous, resized_dim, None, backend=backend,
                         device=device, output=output, obs_dtype=obs_dtype, env_id_base=env_id_base, score_atlas=score_atlas)
        return TournamentEnvWrapper(envs, num_envs)
    if log_dir:
        os.makedirs(log_dir, exist_ok=True)
    if env_id == "cPong-v0":  # make_env_a2c_atari + FrameStack(frame_stack) (atari_wrappers.py:40-53)
        k = 1 if frame_stack is None else int(frame_stack)
        return HipPongVecEnv(num_envs, seed=seed, mode="wrapped", resized_dim=resized_dim, frame_stack=k, device=device,
                             env_id_base=env_id_base, output=output, obs_dtype=obs_dtype,
                             dones="subproc" if asynchronous else "dummy", single_player=True, stack_replicate=True,
                             score_atlas=score_atlas)
    if env_id == "cCarRacing-v0":  # make_car_racing (car_racing/register.py:29-40): FrameStack, one car
        return HipCarVecEnv(num_envs, seed=seed, device=device, env_id_base=env_id_base, output=output,
                            dones="subproc" if asynchronous else "dummy", action_repeat=action_repeat,
                            frame_stack=frame_stack, players=1)
    if env_id == "cCarRacingDouble-v0":
        return HipCarVecEnv(num_envs, seed=seed, device=device, env_id_base=env_id_base, output=output,
                            dones="subproc" if asynchronous else "dummy", action_repeat=action_repeat,
                            frame_stack=frame_stack)
    if env_id == "cPongDouble-v0":
        assert frame_stack is None
    return HipPongVecEnv(num_envs, seed=seed, mode="wrapped", resized_dim=resized_dim, frame_stack=stack_planes,
                         device=device, env_id_base=env_id_base, output=output, obs_dtype=obs_dtype,
                         dones="subproc" if asynchronous else "dummy", score_atlas=score_atlas)
