"""Stand-ins for Box2D / matplotlib / the pygame calls of the CarRacing files, and a loader for the
reference's car modules.  CONTAINER-ONLY TOOLING for the ``gen_car_*_golden.py`` scripts (see
``_ref_standins.py``); nothing here is imported by the product, the tests or the bench.

The Box2D stand-in does NO physics: bodies are attribute bags whose positions / velocities a
generator script sets itself, ``ApplyForceToCenter`` and ``joint.motorSpeed`` are recorded, and
``World.Step`` calls an optional ``on_step(world)`` hook -- which is how a generator scripts what the
third-party engine "did" (moved bodies, raised Begin/EndContact on the reference's own listener)
while every line of bookkeeping around it stays the reference's.
"""
import math
import sys
import types

import numpy as np

import _ref_standins as S


class Vec2(tuple):
    def __new__(cls, x, y=None):
        if y is None:
            x, y = x
        return tuple.__new__(cls, (float(x), float(y)))

    x = property(lambda s: s[0])
    y = property(lambda s: s[1])

    def __add__(self, o):
        return Vec2(self[0] + o[0], self[1] + o[1])

    def __sub__(self, o):
        return Vec2(self[0] - o[0], self[1] - o[1])


class Shape:
    def __init__(self, vertices=None):
        self.vertices = list(vertices or [])


class FixtureDef:
    def __init__(self, shape=None, **kw):
        self.shape = shape
        self.__dict__.update(kw)


class Fixture:
    def __init__(self, shape):
        self.shape = Shape(shape.vertices)
        self.sensor = False


class Body:
    def __init__(self, position=(0, 0), angle=0.0, fixtures=None):
        self.position = Vec2(position)
        self.angle = float(angle)
        self.linearVelocity = Vec2(0, 0)
        self.angularVelocity = 0.0
        fx = fixtures if isinstance(fixtures, list) else [fixtures]
        self.fixtures = [Fixture(f.shape) for f in fx if f is not None]
        self.forces = []
        self.userData = None

    def GetWorldVector(self, v):  # float32 rotation like b2Rot, returned as python floats
        s, c = np.float32(math.sin(self.angle)), np.float32(math.cos(self.angle))
        x, y = np.float32(v[0]), np.float32(v[1])
        return Vec2(float(c * x - s * y), float(s * x + c * y))

    def ApplyForceToCenter(self, f, wake):
        self.forces.append((float(f[0]), float(f[1])))


class Joint:
    def __init__(self):
        self.angle = 0.0
        self.motorSpeed = 0.0


class World:
    on_step = None  # generator hook: called with the world on every Step

    def __init__(self, *a, contactListener=None, **k):
        self.static = []
        self.listener = contactListener
        self.steps = 0

    def CreateStaticBody(self, fixtures=None):
        b = Body(fixtures=fixtures)
        self.static.append(b)
        return b

    def CreateDynamicBody(self, **kw):
        return Body(**kw)

    def CreateJoint(self, jd):
        return Joint()

    def DestroyBody(self, b):
        pass

    def Step(self, *a):
        self.steps += 1
        if World.on_step is not None:
            World.on_step(self)


class Transform:
    position = (0, 0)
    angle = 0.0


def install_box2d():
    b2 = types.ModuleType("Box2D")
    b2.b2World = World
    b2.b2Transform = Transform
    b2.b2Vec2 = Vec2
    sub = types.ModuleType("Box2D.b2")
    sub.fixtureDef = FixtureDef
    sub.polygonShape = Shape
    sub.revoluteJointDef = lambda **kw: types.SimpleNamespace(**kw)
    sub.contactListener = type("contactListener", (), {"__init__": lambda self: None})
    b2.b2 = sub
    sys.modules["Box2D"], sys.modules["Box2D.b2"] = b2, sub


class Draws:
    """np_random stand-in: uniform(a, b) = a + (b - a) * u with u from an explicit, recorded stream."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.u = []

    def uniform(self, a, b):
        u = float(self.rs.random_sample())
        self.u.append(u)
        return a + (b - a) * u


class LazySurface(S.Surface):
    """pygame.Surface stand-in that does not allocate pixels (CarRacing builds a 10000 x 10000 one)."""

    def __init__(self, size, *a, **k):
        self.size = (int(size[0]), int(size[1]))
        self.px = None

    def fill(self, color):
        pass

    def subsurface(self, *a, **k):
        return self


def load_car_reference(seed_stream=lambda seed: Draws(0)):
    """Installs every stand-in and imports the reference's car modules by path.
    Returns (car_dynamics, car_racing_multi_players)."""
    S.install()
    install_box2d()
    gym = sys.modules["gym"]
    utils = types.ModuleType("gym.utils")
    utils.seeding = types.SimpleNamespace(np_random=lambda seed=None: (seed_stream(seed), seed))
    utils.EzPickle = type("EzPickle", (), {"__init__": lambda self, *a, **k: None})
    gym.utils = utils
    sys.modules["gym.utils"] = utils
    sys.modules["matplotlib"] = types.ModuleType("matplotlib")
    sys.modules["matplotlib.pyplot"] = types.ModuleType("matplotlib.pyplot")
    pg = sys.modules["pygame"]
    pg.Surface = LazySurface
    pg.font.init = lambda: None
    pg.image = types.SimpleNamespace(load=lambda p: LazySurface((30, 52)))
    pg.transform = types.SimpleNamespace(scale=lambda im, sz: LazySurface(sz), rotate=lambda im, a: im)
    pg.draw.polygon = lambda *a, **k: None

    # spaces.Box(np.array, np.array, dtype=) form used by CarRacing.__init__
    Box = sys.modules["gym.spaces"].Box

    def box_init(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.asarray(low).shape
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.low, self.high = np.broadcast_to(low, self.shape).astype(dtype), np.broadcast_to(high, self.shape).astype(dtype)

    Box.__init__ = box_init
    Dict = sys.modules["gym.spaces"].Dict
    Dict.__getitem__ = lambda self, k: self.spaces[k]
    S.load_ref("competitive_rl.car_racing.pygame_rendering", "car_racing/pygame_rendering.py")
    cd = S.load_ref("competitive_rl.car_racing.car_dynamics", "car_racing/car_dynamics.py")
    cr = S.load_ref("competitive_rl.car_racing.car_racing_multi_players", "car_racing/car_racing_multi_players.py")
    return cd, cr
