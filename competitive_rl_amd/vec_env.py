"""HIP vector-env backend behind the reference's VecEnv protocol.

Mirrors, for cPongDouble-v0, what ``DummyVecEnv([make_env_a2c_atari(...)]*N)`` does in the
reference (competitive_rl/utils/dummy_vec_env.py:10-133, base_vec_env.py:63-252,
atari_wrappers.py:40-53) -- same method names, argument meaning, return structure and
error behaviour -- but one ``step`` is two HIP kernel launches over all N envs, and the
returned arrays are PyTorch-ROCm tensors that live in HBM.

PyTorch is plumbing here (device memory, stream handle, torch.distributed); the
compute is libcrl_hip.so (include/crl.h) reached through ctypes.  No CPU fallback.
"""
import ctypes as C
from abc import ABC, abstractmethod

import numpy as np
import torch

from . import _native as N
from . import spaces

CHEAT_CODES = 999  # pong/base_pong_env.py:9


class VecEnv(ABC):
    """The reference's abstract vectorised env (utils/base_vec_env.py:63-252)."""

    metadata = {"render.modes": ["human", "rgb_array"]}

    def __init__(self, num_envs, observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.action_space = action_space

    @abstractmethod
    def reset(self):
        pass

    @abstractmethod
    def step_async(self, actions):
        pass

    @abstractmethod
    def step_wait(self):
        pass

    @abstractmethod
    def close(self):
        pass

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def get_images(self, *args, **kwargs):
        raise NotImplementedError

    def render(self, mode="rgb_array", *args, **kwargs):
        """utils/base_vec_env.py:173-192 and DummyVecEnv.render (dummy_vec_env.py:87-102): one env -> its own image, several ->
        all of them tiled into one picture (``tile_images``).  mode="human" would open a window (cv2.imshow): there is no display
        on a GPU node."""
        imgs = self.get_images(*args, **kwargs)
        if mode == "rgb_array":
            return imgs[0] if self.num_envs == 1 else tile_images(imgs)
        raise NotImplementedError("only mode='rgb_array' is available on the GPU backend")

    @property
    def unwrapped(self):
        return self.venv.unwrapped if isinstance(self, VecEnvWrapper) else self

    def getattr_depth_check(self, name, already_found):
        """Name of this class when it holds ``name`` and a wrapper above already did (base_vec_env.py:201-212)."""
        if hasattr(self, name) and already_found:
            return "{0}.{1}".format(type(self).__module__, type(self).__name__)
        return None

    def _get_indices(self, indices):
        if indices is None:
            indices = range(self.num_envs)
        elif isinstance(indices, int):
            indices = [indices]
        return indices


def tile_images(img_nhwc):
    """N images of one size as ONE picture of P x Q cells, row-major, P = ceil(sqrt(N)) rows and Q = ceil(N / P) columns, the
    unused cells black (what ``VecEnv.render`` shows for several envs, utils/base_vec_env.py:10-38).  Gray (N, H, W) frames --
    CarRacing's -- give an (P H, Q W) picture."""
    imgs = np.asarray(img_nhwc)
    gray = imgs.ndim == 3
    if gray:
        imgs = imgs[..., None]
    n, h, w, c = imgs.shape
    rows = int(np.ceil(np.sqrt(n)))
    cols = int(np.ceil(float(n) / rows))
    sheet = np.zeros((rows * cols, h, w, c), imgs.dtype)
    sheet[:n] = imgs
    big = sheet.reshape(rows, cols, h, w, c).swapaxes(1, 2).reshape(rows * h, cols * w, c)
    return big[..., 0] if gray else big


class VecEnvWrapper(VecEnv):
    """Base class of wrappers around a vector env (utils/base_vec_env.py:255-374): forwards ``step_async``, ``seed``, ``close``,
    ``render``, ``get_images``, ``get_attr`` / ``set_attr`` / ``env_method`` to ``venv``; a subclass provides ``reset`` and
    ``step_wait``.  Attributes the wrapper does not have are looked up down the chain of wrappers; one that a wrapper AND something
    below it define is refused as ambiguous, as in the reference."""

    def __init__(self, venv, observation_space=None, action_space=None):
        self.venv = venv
        VecEnv.__init__(self, num_envs=venv.num_envs, observation_space=observation_space or venv.observation_space,
                        action_space=action_space or venv.action_space)
        self.class_attributes = {k for klass in type(self).__mro__ for k in vars(klass)}

    def step_async(self, actions):
        self.venv.step_async(actions)

    @abstractmethod
    def reset(self):
        pass

    @abstractmethod
    def step_wait(self):
        pass

    def seed(self, seed=None):
        return self.venv.seed(seed)

    def close(self):
        return self.venv.close()

    def render(self, *args, **kwargs):
        return self.venv.render(*args, **kwargs)

    def get_images(self, *args, **kwargs):
        return self.venv.get_images(*args, **kwargs)

    def get_attr(self, attr_name, indices=None):
        return self.venv.get_attr(attr_name, indices)

    def set_attr(self, attr_name, value, indices=None):
        return self.venv.set_attr(attr_name, value, indices)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        return self.venv.env_method(method_name, *method_args, indices=indices, **method_kwargs)

    def _own(self, name):
        return name in self.__dict__ or name in self.class_attributes

    def __getattr__(self, name):  # (only reached when normal lookup failed)
        if name in ("venv", "class_attributes"):
            raise AttributeError(name)
        blocked = self.getattr_depth_check(name, already_found=False)
        if blocked is not None:
            raise AttributeError("Error: Recursive attribute lookup for {0} from {1}.{2} is ambiguous and hides attribute from {3}".format(
                name, type(self).__module__, type(self).__name__, blocked))
        return self.getattr_recursive(name)

    def getattr_recursive(self, name):
        if self._own(name):
            return getattr(self, name)
        if hasattr(self.venv, "getattr_recursive"):
            return self.venv.getattr_recursive(name)
        return getattr(self.venv, name)

    def getattr_depth_check(self, name, already_found):
        if self._own(name) and already_found:
            return "{0}.{1}".format(type(self).__module__, type(self).__name__)
        return self.venv.getattr_depth_check(name, already_found or self._own(name))


class LazyInfos:
    """``infos`` of one step without building N dicts (65 536 Python dicts per step would
    cost more than the simulation).  Behaves like the reference's list of dicts:
    ``infos[i]`` -> ``{"real_reward": [l, r], "num_steps": k}`` plus
    ``"terminal_observation"`` on the step that ended env i's episode
    (atari_wrappers.py:179-180, dummy_vec_env.py:55-57).  Host copies happen on first use; the
    terminal observations of ALL envs that finished in this step are drawn by one library call
    (device index list, no per-env round trip).

    Validity: the scalar fields are snapshots and stay valid; ``terminal_observation`` is drawn
    from the context's record of each env's most recent terminal frames and (with a frame stack)
    the previous observation buffer, so it must be read before the env is stepped again --
    afterwards it raises instead of returning a later episode's frames.
    """

    def __init__(self, env, wrapped, done, real_reward, num_steps, single=False):
        self._env, self._wrapped, self._single = env, wrapped, single
        self._done_dev, self._rr_dev, self._ns_dev = done, real_reward, num_steps
        self._host = None
        self._serial = env._serial
        self._term = None  # env index -> terminal observation, filled by the first request

    def __len__(self):
        return self._env.num_envs

    def _sync(self):
        if self._host is None:
            self._host = (self._done_dev.cpu().numpy(), self._rr_dev.cpu().numpy(), self._ns_dev.cpu().numpy())
        return self._host

    def _terminal(self, i):
        if self._term is None:
            if self._env._serial != self._serial:
                raise RuntimeError("terminal_observation of a past step: read infos[i] before stepping the env again")
            idx = torch.nonzero(self._done_dev).reshape(-1)
            self._term = dict(zip(idx.cpu().tolist(), self._env.terminal_observation(idx)))
        return self._term[i]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        done, rr, ns = self._sync()
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        d = {}
        if self._wrapped:
            d["real_reward"] = float(rr[i, 0]) if self._single else [float(rr[i, 0]), float(rr[i, 1])]
            d["num_steps"] = int(ns[i])
        if done[i]:
            d["terminal_observation"] = self._terminal(int(i))
        return d

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def copy(self):
        return self


class _EnvHandle:
    """What ``DummyVecEnv.envs[i]`` offers to the reference's scripts (vis.py:28-46, test/test_pong.py:13): ``close()``, ``render()`` and
    -- for a batch of ONE env, which is what vis.py builds -- the gym calls ``reset()`` / ``step(action)`` of that env: the observation
    without the batch axis (a tuple of the two agents' for Pong), the reward row, a scalar ``done``, the env's info dict.  The vector env
    restarts a finished env by itself; the handle returns the episode's LAST observation with ``done`` (gym's contract) and its next
    ``reset()`` hands out the restarted episode's first observation instead of resetting a second time -- the env sees exactly the
    resets a gym loop (``while not done: step``, then ``reset``) would issue."""

    def __init__(self, venv, idx):
        self._venv, self._idx = venv, idx
        self.observation_space = venv.observation_space
        self.action_space = venv.action_space

    def close(self):
        return None

    def render(self, mode="rgb_array", **_):
        return self._venv.get_images()[self._idx]

    def seed(self, seed=None):
        return None

    def _alone(self):
        if self._venv.num_envs != 1:
            raise NotImplementedError("a GPU batch steps as a whole (envs.reset() / envs.step(actions)); envs.envs[i].reset() / .step() exist for "
                                      "a batch of ONE env, make_envs(..., num_envs=1)")
        return self._venv

    @staticmethod
    def _first(obs):
        return tuple(o[0] for o in obs) if isinstance(obs, tuple) else obs[0]

    def reset(self):
        v = self._alone()
        pending = getattr(v, "_handle_restarted", None)
        v._handle_restarted = None
        if pending is not None and pending[0] == v._serial:  # the episode ended on the last step: the vector env has already restarted it
            return self._first(pending[1])
        return self._first(v.reset())

    def step(self, action):
        v = self._alone()
        space = v.action_space
        shape = (len(space),) if hasattr(space, "spaces") else tuple(space.shape or ())
        scalar = lambda x: x.reshape(-1)[0].item() if isinstance(x, torch.Tensor) else np.asarray(x).reshape(-1)[0]  # noqa: E731
        a = np.asarray([scalar(x) for x in action] if hasattr(space, "spaces") else (action.cpu().numpy() if isinstance(action, torch.Tensor) else action))
        obs, rew, done, info = v.step(a.reshape((1,) + shape))
        d = bool(np.asarray(done.cpu() if isinstance(done, torch.Tensor) else done).reshape(-1).all())
        i0 = info[0]
        ob = self._first(obs)
        if d:
            v._handle_restarted = (v._serial, obs)
            ob = i0["terminal_observation"]
        return ob, rew[0], d, i0


class HipPongVecEnv(VecEnv):
    """N cPongDouble-v0 envs stepped on one MI355X.

    mode="wrapped": make_env_a2c_atari semantics -- obs tuple of two (N, K, R, R) tensors
    (K = frame_stack planes, 1 by default as in the reference where cPongDouble forbids
    FrameStack, make_envs.py:105-106), rewards (N, 2) float32 in {-1,0,1}, dones (N, 2)
    bool (DummyVecEnv) or (N,) (dones="subproc").
    mode="raw": the unwrapped env under a DummyVecEnv -- obs tuple of two
    (N, 210, 160, 3) uint8 tensors, 1 step = 1 frame.
    Observations are views into a double buffer: valid until the next-but-one step().
    """

    def __init__(self, num_envs, seed=0, mode="wrapped", resized_dim=84, frame_stack=1, device=None,
                 env_id_base=0, output="torch", obs_dtype="uint8", dones="dummy", score_atlas=None,
                 single_player=False, stack_replicate=False):
        if not torch.cuda.is_available():
            raise RuntimeError("HipPongVecEnv needs a ROCm GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        assert mode in ("wrapped", "raw") and output in ("torch", "numpy") and dones in ("dummy", "subproc")
        assert obs_dtype in ("uint8", "float32", "float32_ref")
        if obs_dtype == "float32_ref" and mode != "wrapped":
            raise ValueError('obs_dtype="float32_ref" is the wrapped (WarpFrame) observation of the reference\'s float32 step path')
        self._L = N.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.mode, self.R, self.K = mode, int(resized_dim), int(frame_stack)
        self.output, self.obs_dtype, self.dones_kind = output, obs_dtype, dones
        self.closed = False
        self.single, self.V = bool(single_player), 1 if single_player else 2
        self.stack_replicate = bool(stack_replicate)
        opts = N.CrlOpts(env_kind=N.CRL_ENV_PONG_SINGLE if single_player else N.CRL_ENV_PONG_DOUBLE,
                         obs_mode=N.CRL_OBS_GRAY_RESIZED if mode == "wrapped" else N.CRL_OBS_RAW_RGB,
                         resized_dim=self.R if mode == "wrapped" else 0, frame_stack=self.K if mode == "wrapped" else 1,
                         num_envs=int(num_envs), env_id_base=int(env_id_base), seed=int(seed) & (2 ** 64 - 1),
                         device=self.device.index or 0, flags=N.CRL_FLAG_STACK_REPLICATE if stack_replicate else 0,
                         obs_dtype=(N.CRL_OBS_F32_REF if obs_dtype == "float32_ref" else N.CRL_OBS_F32) if (obs_dtype != "uint8" and mode == "wrapped")
                         else N.CRL_OBS_U8)
        self._atlas = N.load_score_atlas() if score_atlas is None else np.ascontiguousarray(score_atlas, np.uint8)
        assert self._atlas.size == N.ATLAS_BYTES
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            N.check(self._L.crl_create(C.byref(opts), self._atlas.ctypes.data_as(C.c_void_p), C.byref(h)))
        self._h = h
        n = int(num_envs)
        if mode == "wrapped":
            self._obs_shape = (n, self.V, self.K, self.R, self.R)
            box = spaces.Box(0, 255, (self.K, self.R, self.R), dtype=np.float32)
        else:
            self._obs_shape = (n, self.V, 210, 160, 3)
            box = spaces.Box(0, 255, (210, 160, 3), dtype=np.float32)
        if self.single:  # cPong-v0: Box / Discrete(3) (pong/base_pong_env.py:22-25)
            VecEnv.__init__(self, n, box, spaces.Discrete(3))
        else:
            VecEnv.__init__(self, n, spaces.Tuple([box, box]), spaces.Tuple([spaces.Discrete(3), spaces.Discrete(3)]))
        dev = self.device
        # float32 observations (DummyVecEnv's buffer dtype): the wrapped raster stores them directly; the raw RGB frames are
        # uint8 in the library and widened by torch on request (a 53 GB tensor at 65 536 envs -- not a hot path)
        # "float32_ref": the values the reference's step() itself returns under old gym's float32 Box -- unrounded INTER_AREA averages
        # of the float gray frame on step(), rounded on reset() (include/crl.h CRL_OBS_F32_REF); "float32": the uint8 values, widened
        self._buf_dtype = torch.float32 if (obs_dtype != "uint8" and mode == "wrapped") else torch.uint8
        self._obs = [torch.empty(self._obs_shape, dtype=self._buf_dtype, device=dev) for _ in range(2)]
        self._flip = 0
        self._serial = 0  # steps + resets so far: lazy infos check it before drawing terminal observations
        self._rew = torch.zeros((n,) if self.single else (n, 2), dtype=torch.float32, device=dev)
        self._done = torch.zeros((n,), dtype=torch.uint8, device=dev)
        self._actions = torch.zeros((n,) if self.single else (n, 2), dtype=torch.int32, device=dev)
        self.envs = _EnvList(self)
        self.waiting = False
        # the done flags on the host while the step's draw still runs (crl_set_flags_event; step_envs reads them every step)
        self._flags_event = None
        self._flags_side = None
        self._flags_host = None
        self._flags_serial = -1
        self._flags_armed = -1
        # a FrameStackTensor bound to this env (frame_stack.py): step / reset then draw its next state with the observation
        self._bound_stack = None       # weakref
        self._last_kind = None         # "reset" | "step": what produced the newest observation
        self._learner_obs = None       # agent 0's newest observation as handed out (torch output only)

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _format_obs(self, buf, learner=None):
        views = [buf[:, v] for v in range(self.V)]
        if learner is not None:
            views[0] = learner  # (a bound stack's newest plane stands in for agent 0's tile)
        if self.obs_dtype == "float32" and buf.dtype != torch.float32:  # raw mode only
            views = [v.float() for v in views]
        if self.output == "numpy":
            views = [v.cpu().numpy() for v in views]
        return views[0] if self.single else tuple(views)

    def _check_open(self):
        if self.closed:
            raise RuntimeError("VecEnv is closed")

    # ------------------------------------------------------------------ early done flags
    def done_host(self):
        """The last step's done flags as a host bool array (N,), WITHOUT waiting for the step's observation: the library records an
        event right behind the kernel that writes them, a side stream copies them into pinned memory behind that event, and the host
        waits for the copy only -- the 0.2-1.5 ms draw of the observation (and of a bound frame stack) keeps running under the
        caller's bookkeeping.  What ``step_envs`` walks after every step (reference utils/utils.py:33-42)."""
        self._check_open()
        if self._last_kind != "step":
            raise RuntimeError("done_host() reads the flags of the last step(): step the env first")
        if self._flags_serial != self._serial:
            with torch.cuda.device(self.device):
                if self._flags_event is None:
                    self._flags_host = torch.empty((self.num_envs,), dtype=torch.uint8, pin_memory=True)
                    self._flags_side = torch.cuda.Stream(device=self.device)
                cur = torch.cuda.current_stream(self.device)
                if self._flags_event is None or self._flags_armed != self._serial:
                    # the event was not armed for this step (first use): order the copy behind everything enqueued so far
                    self._flags_side.wait_stream(cur)
                else:
                    self._flags_side.wait_event(self._flags_event)
                with torch.cuda.stream(self._flags_side):
                    self._flags_host.copy_(self._done, non_blocking=True)
                self._flags_side.synchronize()
                if self._flags_event is None:  # from now on every step records it
                    self._flags_event = torch.cuda.Event()
                    self._flags_event.record(cur)  # (creates the handle)
                    handle = self._flags_event.cuda_event
                    N.check(self._L.crl_set_flags_event(self._h, C.c_void_p(getattr(handle, "value", handle))))
            self._flags_serial = self._serial
        return self._flags_host.numpy().astype(bool)

    # ------------------------------------------------------------------ bound FrameStackTensor (frame_stack.py)
    def _stack_env(self):
        return self

    def _can_draw_stack(self, fst):
        """crl_step_stack's conditions (include/crl.h): FrameStackTensor's history rule, one (1, R, R) plane per agent and step."""
        return (self.mode == "wrapped" and not self.stack_replicate and self.K == 1 and self.output == "torch" and not self.closed
                and fst.device == self.device and fst.num_envs == self.num_envs and fst.num_channels == 1 and 1 <= fst.frame_stack <= 4
                and fst.plane_shape == (self.R, self.R) and (fst.dtype == torch.float32 or self._buf_dtype == torch.uint8))

    def _stack_alias(self, fst):
        """The stack's newest plane IS agent 0's observation when both are one element type: that tile is then written once."""
        return self.K == 1 and fst.dtype == self._buf_dtype

    def _stack_predraw(self, kind, alias_ok=True):
        fst = self._bound_stack() if self._bound_stack is not None else None
        if fst is None:
            return None, None, None
        pre = fst._predraw(self, kind)
        if pre is None:
            return None, None, None
        buf, desc = pre
        if not alias_ok:
            desc.alias_newest = 0
        return fst, buf, desc

    def _draw_stack_into(self, desc):
        with torch.cuda.device(self.device):
            N.check(self._L.crl_draw_stack(self._h, None, C.byref(desc), self._stream()))

    def _is_latest_learner_obs(self, obs):
        mine = self._learner_obs
        return (mine is not None and isinstance(obs, torch.Tensor) and obs.data_ptr() == mine.data_ptr() and obs.shape == mine.shape
                and obs.dtype == mine.dtype and obs.stride() == mine.stride())

    def _latest_learner_obs(self):
        if self._learner_obs is None:
            raise RuntimeError("the env has produced no observation yet (reset() it first)")
        return self._learner_obs

    def _note_obs(self, kind, buf, stack_buf=None, fst=None, aliased=False):
        """Books after a reset / step that drew into `buf` (and, for a bound stack, its next state into `stack_buf`)."""
        self._last_kind = kind
        if self.output != "torch":
            self._learner_obs = None
            return None
        k = fst.frame_stack if fst is not None else 0
        learner = stack_buf[:, k - 1:k] if aliased else buf[:, 0]
        self._learner_obs = learner
        if fst is not None:
            fst._predrawn(self, stack_buf, kind)
        return learner

    # ------------------------------------------------------------------ VecEnv protocol
    def seed(self, seed=None):
        """Env i gets seed + i (dummy_vec_env.py:65-69): here the counter-based serve
        sampler is keyed by (seed, global env id), which is the same partition."""
        self._check_open()
        N.check(self._L.crl_seed(self._h, int(seed or 0) & (2 ** 64 - 1)))
        return [None] * self.num_envs

    def reset(self):
        self._check_open()
        buf = self._obs[self._flip]
        fst, sbuf, desc = self._stack_predraw("reset")
        N.check(self._L.crl_reset(self._h, None if desc is not None else C.c_void_p(buf.data_ptr()), self._stream()))
        if desc is not None:  # the first observation and the bound stack's first state in one launch
            N.check(self._L.crl_draw_stack(self._h, C.c_void_p(buf.data_ptr()), C.byref(desc), self._stream()))
        self._flip ^= 1
        self._serial += 1
        return self._format_obs(buf, self._note_obs("reset", buf, sbuf, fst, bool(desc is not None and desc.alias_newest)))

    def step_async(self, actions):
        self._check_open()
        if isinstance(actions, torch.Tensor):
            a = actions.to(device=self.device, dtype=torch.int32)  # checked on the device (CRL_EACTION)
        else:
            host = np.asarray(actions)
            # `assert self.action_space.contains(action)` (pong/base_pong_env.py:42; 999 is decoded first, :116-134)
            if not np.isin(host, (0, 1, 2, CHEAT_CODES)).all():
                raise AssertionError(f"actions outside the action space {{0, 1, 2, {CHEAT_CODES}}}: {np.unique(host)[:8]}")
            a = torch.as_tensor(host, dtype=torch.int32).to(self.device)
        want = (self.num_envs,) if self.single else (self.num_envs, 2)
        if self.single and a.dim() == 2 and a.shape[1] == 1:
            a = a[:, 0]
        if tuple(a.shape) != want:
            raise AssertionError(f"actions must have shape {want}, got {tuple(a.shape)}")
        self._actions = a.contiguous()
        self.waiting = True

    def check(self):
        """Synchronises and raises if an action outside the action space reached a step kernel since the last
        check (device-resident actions are validated by the kernel; the bat of such an action does not move)."""
        N.check(self._L.crl_check(self._h, self._stream()))

    def step_wait(self):
        self._check_open()
        buf = self._obs[self._flip]
        self.waiting = False
        # (the library first: a refused call -- e.g. the report of an earlier out-of-range action -- has not stepped the envs,
        # so the buffer flip and the serial that lazy infos check stay where they are)
        fst, sbuf, desc = self._stack_predraw("step")
        N.check(self._L.crl_step_stack(self._h, C.c_void_p(self._actions.data_ptr()), C.c_void_p(buf.data_ptr()),
                                       C.c_void_p(self._rew.data_ptr()), C.c_void_p(self._done.data_ptr()),
                                       None if desc is None else C.byref(desc), self._stream()))
        self._flip ^= 1
        self._serial += 1
        self._flags_armed = self._serial if self._flags_event is not None else -1
        learner = self._note_obs("step", buf, sbuf, fst, bool(desc is not None and desc.alias_newest))
        done = self._done.bool()
        if self.dones_kind == "dummy":  # scalar done broadcast over the agents (dummy_vec_env.py:39-40)
            done_out = done[:, None].expand(-1, self.V)
        else:
            done_out = done
        rr = torch.empty((self.num_envs, 2), dtype=torch.float32, device=self.device)
        ns = torch.empty((self.num_envs,), dtype=torch.int32, device=self.device)
        N.check(self._L.crl_copy_info(self._h, C.c_void_p(rr.data_ptr()), C.c_void_p(ns.data_ptr()), self._stream()))
        infos = LazyInfos(self, self.mode == "wrapped", self._done.clone(), rr, ns, single=self.single)
        rew = self._rew.clone()
        if self.single and self.dones_kind == "dummy":
            rew = rew[:, None]  # buf_rews is (N, multi_agent = 1)
        self._prev_buf = self._obs[self._flip]  # the observation before this step (still intact)
        if self.output == "numpy":
            return self._format_obs(buf), rew.cpu().numpy(), done_out.cpu().numpy().copy(), infos
        return self._format_obs(buf, learner), rew, done_out.clone(), infos

    def close(self):
        if self.closed:
            return
        self.closed = True
        fst = self._bound_stack() if self._bound_stack is not None else None
        if fst is not None:
            fst.unbind()  # (the stack lives on, on the generic kernel)
        torch.cuda.synchronize(self.device)
        self._L.crl_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_attr(self, attr_name, indices=None):
        return [getattr(self.envs[i], attr_name) for i in self._get_indices(indices)]

    def set_attr(self, attr_name, value, indices=None):
        for i in self._get_indices(indices):
            setattr(self.envs[i], attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        return [getattr(self.envs[i], method_name)(*args, **kwargs) for i in self._get_indices(indices)]

    def get_images(self, *args, **kwargs):
        """Raw RGB view of agent 0 for every env (render(mode='rgb_array'))."""
        st = self.get_state()
        frames = np.zeros(self.num_envs, N.FRAME_DT)
        for k in ("ball_x", "ball_y", "bat_l_y", "bat_r_y", "score_l", "score_r"):
            frames[k] = st[k]
        return list(self.render_frames(frames)[:, 0].cpu().numpy())

    # ------------------------------------------------------------------ extras (parity tests, checkpoint)
    def terminal_observation(self, env_indices):
        """Observation the episode of each listed env ended on (its most recent done step).  ``env_indices`` may be a
        device tensor (e.g. ``torch.nonzero(done)``): the whole batch is then gathered and drawn on the device by one
        call, without a host round trip."""
        if isinstance(env_indices, torch.Tensor):
            idx = env_indices.to(device=self.device, dtype=torch.int64).reshape(-1).contiguous()
            m = idx.numel()
        else:
            host_idx = np.ascontiguousarray(env_indices, np.int64).reshape(-1)
            if ((host_idx < 0) | (host_idx >= self.num_envs)).any():
                raise IndexError(f"env index out of range: {host_idx}")
            idx = torch.as_tensor(host_idx).to(self.device)
            m = len(host_idx)
        shape = (m, self.V, 210, 160, 3) if self.mode == "raw" else (m, self.V, self.R, self.R)
        out = torch.empty(shape, dtype=self._buf_dtype, device=self.device)
        if m:
            N.check(self._L.crl_terminal_observation_dev(self._h, C.c_void_p(idx.data_ptr()), m, C.c_void_p(out.data_ptr()),
                                                         self._stream()))
        if self.obs_dtype == "float32" and out.dtype != torch.float32:
            out = out.float()
        stacked = self.mode != "raw" and self.stack_replicate and self.K > 1
        if stacked:
            # FrameStack wrapper: the terminal observation is the whole stack = the K-1 newest
            # planes of the previous observation + the terminal frame (atari_wrappers.py:249-252)
            prev = self._prev_buf[idx][:, :, 1:]
            out = torch.cat([prev, out[:, :, None]], dim=2)
        if self.output == "numpy":
            out = out.cpu().numpy()
        res = []
        for k in range(m):
            if self.mode == "raw" or stacked:
                pair = tuple(out[k, v] for v in range(self.V))
            else:
                pair = tuple(out[k, v][None] for v in range(self.V))  # (1, R, R) each, WrapPyTorch layout
            res.append(pair[0] if self.single else pair)
        return res

    def get_state(self):
        st = np.zeros(self.num_envs, N.STATE_DT)
        N.check(self._L.crl_get_state(self._h, st.ctypes.data_as(C.c_void_p), 0, self.num_envs, self._stream()))
        return st

    def set_state(self, st):
        st = np.ascontiguousarray(st, N.STATE_DT)
        assert len(st) == self.num_envs
        N.check(self._L.crl_set_state(self._h, st.ctypes.data_as(C.c_void_p), 0, self.num_envs, self._stream()))

    def state_dict(self):
        """Checkpoint of the whole batch in torch's idiom (SURVEY section 5 "checkpoint / resume"): a plain dict that ``torch.save`` /
        ``numpy.save`` can store -- the structured per-env state (ball, bats, scores, f64 speeds, serve counters, the kept frames and
        the frame-stack history) plus what the context was created with.  ``load_state_dict`` puts it back; the next step then
        produces the same bytes as the original run would have (tests/test_hip_pong_parity.py)."""
        return {"kind": "cPong", "num_envs": self.num_envs, "mode": self.mode, "env_state": self.get_state()}

    def load_state_dict(self, sd):
        st = sd["env_state"] if isinstance(sd, dict) else sd  # (a bare get_state() array is accepted too)
        if isinstance(sd, dict) and (sd.get("kind") != "cPong" or int(sd.get("num_envs", -1)) != self.num_envs):
            raise ValueError(f"state_dict of a {sd.get('kind')} batch of {sd.get('num_envs')} envs does not fit this env")
        self.set_state(st)

    def set_replay(self, u, bx, by):
        """Replay mode of the serve sampler: arrays [N, per_env]."""
        if u is None:
            N.check(self._L.crl_set_replay(self._h, None, None, None, 0))
            return
        u = np.ascontiguousarray(u, np.float64).reshape(self.num_envs, -1)
        bx = np.ascontiguousarray(bx, np.uint8).reshape(self.num_envs, -1)
        by = np.ascontiguousarray(by, np.uint8).reshape(self.num_envs, -1)
        N.check(self._L.crl_set_replay(self._h, u.ctypes.data_as(C.c_void_p), bx.ctypes.data_as(C.c_void_p),
                                       by.ctypes.data_as(C.c_void_p), u.shape[1]))

    def render_frames(self, frames):
        """Raw (count, 2, 210, 160, 3) render of explicit frame descriptors (raw mode only)."""
        frames = np.ascontiguousarray(frames, N.FRAME_DT)
        out = torch.empty((len(frames), 2, 210, 160, 3), dtype=torch.uint8, device=self.device)
        if self.mode != "raw":
            tmp = HipPongVecEnv(1, mode="raw", device=self.device)
            try:
                return tmp.render_frames(frames)
            finally:
                tmp.close()
        N.check(self._L.crl_render_raw(self._h, frames.ctypes.data_as(C.c_void_p), len(frames),
                                       C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def obs_descriptors(self, out=None):
        """The frame descriptors the current observation was drawn from: int64 (8, N) on the device, 64 bytes per env (ring
        layout of include/crl.h).  ``render_descriptors`` turns descriptors -- this shard's or any other's -- back into pixels."""
        if out is None:
            out = torch.empty((8, self.num_envs), dtype=torch.int64, device=self.device)
        assert out.is_contiguous() and out.numel() == 8 * self.num_envs and out.element_size() == 8 and out.device == self.device
        N.check(self._L.crl_obs_descriptors(self._h, C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def render_descriptors(self, desc, out=None):
        """Observations of ``count`` envs from their descriptors (int64 (8, count) on the device): same layout as step_device's
        observation buffer, (count, V, ...)."""
        assert desc.is_contiguous() and desc.element_size() == 8 and desc.device == self.device and desc.numel() % 8 == 0
        count = desc.numel() // 8
        if out is None:
            out = torch.empty((count,) + tuple(self._obs_shape[1:]), dtype=self._buf_dtype, device=self.device)
        assert out.is_contiguous() and out.device == self.device
        N.check(self._L.crl_render_frames_dev(self._h, C.c_void_p(desc.data_ptr()), count, C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def step_device(self, actions_i32, render=True, obs_out=None):
        """Hot-loop entry for training/bench code: `actions_i32` is an int32 (N, 2) tensor
        already on the device; returns device tensors, no host work, no clones, no sync.  ``obs_out``: draw the observation
        straight into the caller's tensor (e.g. a slice of a collective's send buffer) instead of the env's double buffer."""
        if not (actions_i32.is_contiguous() and actions_i32.dtype == torch.int32 and actions_i32.device == self.device):
            raise AssertionError("step_device needs a contiguous int32 tensor on the env's device")
        if obs_out is not None:
            if not (obs_out.is_contiguous() and obs_out.dtype == self._buf_dtype and obs_out.device == self.device
                    and obs_out.numel() == self._obs[0].numel()):
                raise AssertionError("obs_out must be a contiguous tensor of the observation buffer's size and dtype on the env's device")
            fst, sbuf, desc = self._stack_predraw("step", alias_ok=False)
            N.check(self._L.crl_step_stack(self._h, C.c_void_p(actions_i32.data_ptr()), C.c_void_p(obs_out.data_ptr()),
                                           C.c_void_p(self._rew.data_ptr()), C.c_void_p(self._done.data_ptr()),
                                           None if desc is None else C.byref(desc), self._stream()))
            self._serial += 1
            self._flags_armed = self._serial if self._flags_event is not None else -1
            out = obs_out.view(self._obs[0].shape)
            self._note_obs("step", out, sbuf, fst)
            return out, self._rew, self._done
        buf = self._obs[self._flip]
        # (a bound FrameStackTensor is drawn too -- FrameStackTensor.update_from_env then swaps it in; the whole observation
        # buffer is written here, no tile is left to the stack: the caller gets `buf` itself)
        fst, sbuf, desc = self._stack_predraw("step", alias_ok=False) if render else (None, None, None)
        N.check(self._L.crl_step_stack(self._h, C.c_void_p(actions_i32.data_ptr()),
                                       C.c_void_p(buf.data_ptr()) if render else None,
                                       C.c_void_p(self._rew.data_ptr()), C.c_void_p(self._done.data_ptr()),
                                       None if desc is None else C.byref(desc), self._stream()))
        self._prev_buf = self._obs[self._flip ^ 1]  # (after the call: a refused call has not stepped the envs)
        self._flip ^= 1
        self._serial += 1
        self._flags_armed = self._serial if self._flags_event is not None else -1
        if render:
            self._note_obs("step", buf, sbuf, fst)
        else:
            self._last_kind, self._learner_obs = "step", None
        return buf, self._rew, self._done

    def kernel_timing(self, enable=True):
        N.check(self._L.crl_kernel_timing(self._h, int(enable)))

    def kernel_time_ms(self, which):
        ms, cnt = C.c_double(), C.c_int64()
        N.check(self._L.crl_kernel_time_ms(self._h, which, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def kernel_time_stats(self, which):
        """(total ms, launches, longest launch in ms) of timing slot `which` (include/crl.h crl_kernel_time_stats; 2 = CarRacing's touching solve)"""
        ms, cnt, mx = C.c_double(), C.c_int64(), C.c_double()
        N.check(self._L.crl_kernel_time_stats(self._h, which, C.byref(ms), C.byref(cnt), C.byref(mx)))
        return ms.value, cnt.value, mx.value


class _EnvList:
    def __init__(self, venv):
        self._venv = venv

    def __len__(self):
        return self._venv.num_envs

    def __getitem__(self, i):
        if not -len(self) <= i < len(self):
            raise IndexError(i)
        return _EnvHandle(self._venv, i % len(self))

    def __iter__(self):
        return (self[i] for i in range(len(self)))
