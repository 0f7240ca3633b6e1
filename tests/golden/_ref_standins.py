"""Throw-away stand-ins for the third-party packages the reference imports.

CONTAINER-ONLY TOOLING.  Used exclusively by the ``gen_*_golden.py`` scripts in
this directory to import the reference's own ``.py`` files (from
``/root/reference``, never copied) and record golden input/output vectors.
Nothing here is imported by the product, the tests or the bench.

The reference needs ``gym``, ``pygame`` and ``cv2``; none is installed and
there is no network.  What the stand-ins assert about those packages is the
part of the golden vectors that is NOT pinned by the reference itself:

* ``pygame.Rect``: C ``int`` fields; assigning a float truncates toward zero;
  ``right = x + w``, ``bottom = y + h``, ``centerx = x + w // 2``.
* ``pygame.draw.rect`` / ``Surface.fill`` on positive-size rects: ordinary
  half-open clipping fill.  ``font.render`` / ``blit`` are no-ops, so golden
  frames carry NO score text (the text band is unpinned).
* ``gym``: old-gym behaviour where ``env.step`` forwards to ``_step`` etc.
"""
import sys
import types

import numpy as np


# --------------------------------------------------------------------------- pygame
class Rect:
    __slots__ = ("_x", "_y", "w", "h")

    def __init__(self, x, y, w, h):
        self._x, self._y, self.w, self.h = int(x), int(y), int(w), int(h)

    # int() truncates toward zero, like pygame's (int)double conversion
    x = property(lambda s: s._x, lambda s, v: setattr(s, "_x", int(v)))
    y = property(lambda s: s._y, lambda s, v: setattr(s, "_y", int(v)))
    left = x
    top = y
    right = property(lambda s: s._x + s.w, lambda s, v: setattr(s, "_x", int(v) - s.w))
    bottom = property(lambda s: s._y + s.h, lambda s, v: setattr(s, "_y", int(v) - s.h))
    centerx = property(lambda s: s._x + (s.w >> 1))
    centery = property(lambda s: s._y + (s.h >> 1))
    width = property(lambda s: s.w)
    height = property(lambda s: s.h)

    def _set_topleft(self, v):
        self.x, self.y = v

    topleft = property(lambda s: (s._x, s._y), _set_topleft)


class Surface:
    """numpy-backed (W, H) RGB surface: fill, draw.rect, array3d only."""

    def __init__(self, size, *a, **k):
        self.size = (int(size[0]), int(size[1]))
        self.px = np.zeros((self.size[0], self.size[1], 3), np.uint8)

    def fill(self, color):
        self.px[:, :] = color

    def blit(self, src, dest):  # text is not reproduced
        return None

    def get_rect(self):
        return Rect(0, 0, *self.size)


def _draw_rect(surface, color, rect, width=0):
    W, H = surface.size
    x0, y0 = max(rect.x, 0), max(rect.y, 0)
    x1, y1 = min(rect.x + rect.w, W), min(rect.y + rect.h, H)
    if x1 > x0 and y1 > y0:
        surface.px[x0:x1, y0:y1] = color


class _Font:
    def __init__(self, *a, **k):
        pass

    def render(self, text, aa, color):
        return Surface((1, 1))


def make_pygame():
    pg = types.ModuleType("pygame")
    pg.Rect = Rect
    pg.Surface = Surface
    pg.init = lambda: None
    pg.quit = lambda: None
    pg.sprite = types.SimpleNamespace(Sprite=object)
    pg.draw = types.SimpleNamespace(rect=_draw_rect)
    pg.font = types.SimpleNamespace(Font=_Font)
    pg.display = types.SimpleNamespace(quit=lambda: None)
    pg.surfarray = types.SimpleNamespace(array3d=lambda s: s.px.copy())
    return pg


# --------------------------------------------------------------------------- gym
class _Space:
    shape = None
    dtype = None


class Box(_Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)


class Discrete(_Space):
    def __init__(self, n):
        self.n = n
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def contains(self, x):
        return 0 <= int(x) < self.n


class Tuple(_Space):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]


class Dict(_Space):
    def __init__(self, spaces):
        self.spaces = spaces


class Env:
    metadata = {}
    observation_space = None
    action_space = None

    # old-gym compat patch: public names forward to the underscored ones
    def step(self, action):
        return self._step(action)

    def reset(self, **kw):
        return self._reset()

    def seed(self, seed=None):
        return self._seed(seed)

    def render(self, mode="human", **kw):
        return self._render(mode)

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.observation_space = env.observation_space
        self.action_space = env.action_space
        self.metadata = env.metadata

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kw):
        return self.env.reset(**kw)

    def seed(self, seed=None):
        return self.env.seed(seed)

    def close(self):
        return None

    @property
    def unwrapped(self):
        return self.env.unwrapped


class ObservationWrapper(Wrapper):
    def reset(self, **kw):
        return self.observation(self.env.reset(**kw))

    def step(self, action):
        o, r, d, i = self.env.step(action)
        return self.observation(o), r, d, i


class TimeLimit(Wrapper):
    """gym.wrappers.TimeLimit as ``gym.make`` applies it for ``register(max_episode_steps=...)`` [from memory of
    gym 0.10-0.21: count steps since reset; at the limit set info["TimeLimit.truncated"] = not done and done = True]."""

    def __init__(self, env, max_episode_steps):
        Wrapper.__init__(self, env)
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = None

    def step(self, action):
        observation, reward, done, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            info["TimeLimit.truncated"] = not done
            done = True
        return observation, reward, done, info

    def reset(self, **kw):
        self._elapsed_steps = 0
        return self.env.reset(**kw)


_REGISTRY = {}


def make_gym():
    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")
    for c in (Box, Discrete, Tuple, Dict):
        setattr(spaces, c.__name__, c)
    gym.spaces = spaces
    gym.Env, gym.Wrapper, gym.ObservationWrapper = Env, Wrapper, ObservationWrapper
    gym.logger = types.SimpleNamespace(set_level=lambda *_: None)
    gym.error = types.SimpleNamespace(Error=Exception)

    def register(id, entry_point, kwargs=None, max_episode_steps=None, **_):
        _REGISTRY[id] = (entry_point, kwargs or {}, max_episode_steps)

    def make(id, **kw):
        ep, k, limit = _REGISTRY[id]
        env = ep(**{**k, **kw})
        return env if limit is None else TimeLimit(env, limit)

    gym.make = make
    envs = types.ModuleType("gym.envs")
    reg = types.ModuleType("gym.envs.registration")
    reg.register = register
    envs.registration = reg
    gym.envs = envs
    gym.register = register
    return gym, {"gym": gym, "gym.spaces": spaces, "gym.envs": envs, "gym.envs.registration": reg}


def install(cv2_module=None):
    """Put the stand-ins into sys.modules (idempotent)."""
    sys.dont_write_bytecode = True  # never drop __pycache__ into /root/reference
    sys.modules["pygame"] = make_pygame()
    _, mods = make_gym()
    sys.modules.update(mods)
    if cv2_module is not None:
        sys.modules["cv2"] = cv2_module
    if not hasattr(np, "bool"):  # the reference predates numpy 1.24
        np.bool = bool
    if not hasattr(np, "float"):
        np.float = float


def load_ref(name, relpath, root="/root/reference/competitive_rl"):
    """Import one reference file by path under its package-qualified name."""
    import importlib.util

    parts = name.split(".")
    for i in range(1, len(parts)):
        pkg = ".".join(parts[:i])
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
    spec = importlib.util.spec_from_file_location(name, f"{root}/{relpath}")
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class ServeStream:
    """Replaces the ``random`` module inside the reference's pong file so the
    serve draws become an explicit, recorded input stream:
    one draw = (u in [0,1), bit_x, bit_y); ``choice([-s, +s])`` picks index ``bit``.
    """

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.u, self.bx, self.by = [], [], []
        self._pending = 0

    def uniform(self, a, b):
        u = float(self.rs.random_sample())
        self.u.append(u)
        self._pending = 0
        return a + (b - a) * u  # CPython's random.uniform formula

    def choice(self, seq):
        bit = int(self.rs.randint(0, 2))
        (self.bx if self._pending == 0 else self.by).append(bit)
        self._pending += 1
        return seq[bit]
