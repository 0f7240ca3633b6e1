"""``make_envs`` -- the drop-in boundary (reference competitive_rl/make_envs.py:67-118).

Same signature and argument meaning as the reference factory; the object returned for the
Pong ids is the HIP vector env instead of ``DummyVecEnv`` / ``SubprocVecEnv``.  Extra
keyword-only arguments select GPU-side options.
"""
import os

from .vec_env import HipPongVecEnv
from .vec_env_car import HipCarVecEnv

__all__ = ["make_envs", "make_env_a2c_atari", "make_car_racing", "make_car_racing_double", "DummyVecEnv", "SubprocVecEnv", "EnvThunk"]

_HIP_IDS = ("cPongDouble-v0", "cPong-v0", "cPongTournament-v0", "cCarRacingDouble-v0", "cCarRacing-v0")


def make_envs(env_id="cPong-v0", seed=0, log_dir="data", num_envs=3, asynchronous=False, resized_dim=42,
              frame_stack=4, action_repeat=None, *, backend="hip", device=None, output="torch",
              obs_dtype="uint8", env_id_base=0, stack_planes=1, score_atlas=None, dones=None):
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
    :param dones: GPU extra -- "dummy" / "subproc": the return convention whatever ``asynchronous`` and ``num_envs`` say (what the
        ``DummyVecEnv`` / ``SubprocVecEnv`` constructors below pass: a SubprocVecEnv of ONE env still returns (N,) dones).
    """
    if dones not in (None, "dummy", "subproc"):
        raise ValueError(f"dones must be None, 'dummy' or 'subproc', got {dones!r}")
    asynchronous = (asynchronous and num_envs > 1) if dones is None else dones == "subproc"
    if backend != "hip":
        raise ValueError("competitive_rl_amd only provides backend='hip'")
    if env_id not in _HIP_IDS:
        raise NotImplementedError(
            f"{env_id!r} is not served by the HIP backend yet (available: {_HIP_IDS})")
    if env_id == "cPongTournament-v0":  # make_envs.py:93-96
        from .tournament import TournamentEnvWrapper

        envs = make_envs("cPongDouble-v0", seed, log_dir, num_envs, asynchronous, resized_dim, None, backend=backend,
                         device=device, output=output, obs_dtype=obs_dtype, env_id_base=env_id_base, score_atlas=score_atlas, dones=dones)
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


class EnvThunk:
    """What the reference's per-env factories return -- ``make_env_a2c_atari(env_id, seed, rank, log_dir, resized_dim, frame_stack)``
    (utils/atari_wrappers.py:40-53), ``make_car_racing`` / ``make_car_racing_double`` (car_racing/register.py:29-53): a callable that
    builds env ``rank`` of a batch.  On the GPU a batch is ONE context, so the thunk only carries its arguments; ``DummyVecEnv`` /
    ``SubprocVecEnv`` below turn a list of them into that context, and calling one by itself gives the env as a batch of one
    (global env id = rank: the same serve / track randomness it has inside a batch)."""

    def __init__(self, env_id, seed, rank, log_dir=None, resized_dim=84, frame_stack=None, action_repeat=None):
        self.env_id, self.seed, self.rank, self.log_dir = env_id, int(seed), int(rank), log_dir
        self.resized_dim, self.frame_stack, self.action_repeat = resized_dim, frame_stack, action_repeat

    def key(self):
        return (self.env_id, self.seed, self.resized_dim, self.frame_stack, self.action_repeat)

    def __call__(self, **gpu_options):
        return make_envs(self.env_id, self.seed, self.log_dir, 1, False, self.resized_dim, self.frame_stack, self.action_repeat,
                         env_id_base=self.rank, **gpu_options)


def make_env_a2c_atari(env_id, seed, rank, log_dir, resized_dim=84, frame_stack=None):
    assert env_id in ("cPong-v0", "cPongDouble-v0"), env_id
    assert frame_stack is None or isinstance(frame_stack, int)
    return EnvThunk(env_id, seed, rank, log_dir, resized_dim, frame_stack)


def make_car_racing(env_id, seed, rank, frame_stack=None, action_repeat=None):
    assert "CarRacing" in env_id  # (car_racing/register.py:30)
    return EnvThunk(env_id, seed, rank, None, 84, frame_stack, action_repeat)


def make_car_racing_double(seed, rank, frame_stack=None, action_repeat=None):
    return EnvThunk("cCarRacingDouble-v0", seed, rank, None, 84, frame_stack, action_repeat)


def _from_thunks(env_fns, dones, gpu_options):
    fns = list(env_fns)
    if not fns or not all(isinstance(f, EnvThunk) for f in fns):
        raise TypeError("the HIP backend builds a batch from the thunks of make_env_a2c_atari / make_car_racing / make_car_racing_double "
                        "(arbitrary env constructors would run on the host: there is no CPU fallback)")
    if any(f.key() != fns[0].key() for f in fns):
        raise ValueError("one batch = one env id, seed, resized_dim, frame_stack and action_repeat")
    ranks = [f.rank for f in fns]
    if ranks != list(range(ranks[0], ranks[0] + len(fns))):
        raise ValueError(f"env i of a batch is rank r0 + i (consecutive global env ids); got ranks {ranks[:8]}")
    f = fns[0]
    return make_envs(f.env_id, f.seed, f.log_dir, len(fns), dones == "subproc", f.resized_dim, f.frame_stack, f.action_repeat,
                     env_id_base=ranks[0], dones=dones, **gpu_options)


def DummyVecEnv(env_fns, **gpu_options):
    """``DummyVecEnv([make_env_a2c_atari(...) for i in range(n)])`` (utils/dummy_vec_env.py:26-46) as one HIP context with the
    DummyVecEnv return conventions ((N, A) dones, ...)."""
    return _from_thunks(env_fns, "dummy", gpu_options)


def SubprocVecEnv(env_fns, start_method=None, **gpu_options):
    """``SubprocVecEnv(env_fns)`` (utils/subproc_vec_env.py:50-118): the same context with the worker convention ((N,) dones); no
    processes are started (``start_method`` is accepted and ignored)."""
    return _from_thunks(env_fns, "subproc", gpu_options)
