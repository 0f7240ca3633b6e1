"""Minimal gym.spaces look-alikes (gym is not a dependency of the backend).

Same constructor arguments and attributes as the gym classes the reference builds its
spaces from (pong/base_pong_env.py:91-101, utils/atari_wrappers.py:12-23,196-201)."""
import numpy as np


class Space:
    shape = None
    dtype = None


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def contains(self, x):
        return 0 <= int(x) < self.n

    def sample(self):
        return int(np.random.randint(self.n))

    def __repr__(self):
        return f"Discrete({self.n})"


class Tuple(Space):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def sample(self):
        return tuple(s.sample() for s in self.spaces)

    def __repr__(self):
        return "Tuple(" + ", ".join(map(repr, self.spaces)) + ")"


class Dict(Space):
    """gym.spaces.Dict as the reference uses it for the two cars' actions (car_racing_multi_players.py:245: ``{0: Box, 1: Box}``)."""

    def __init__(self, spaces):
        self.spaces = dict(spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, k):
        return self.spaces[k]

    def __iter__(self):
        return iter(self.spaces)

    def keys(self):
        return self.spaces.keys()

    def contains(self, x):
        return isinstance(x, dict) and x.keys() == self.spaces.keys() and all(s.contains(x[k]) for k, s in self.spaces.items())

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}

    def __repr__(self):
        return "Dict(" + ", ".join(f"{k}:{s!r}" for k, s in self.spaces.items()) + ")"
