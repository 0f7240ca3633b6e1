"""Stand-ins that let the reference's OWN observation code run in this container:
``CarRacing.get_observation`` -> ``camera_update`` / ``camera_view`` / ``render`` /
``Car.draw_for_pygame`` / ``render_indicators_for_pygame`` / ``render_road_for_observation_map``
(car_racing/car_racing_multi_players.py:622-863, car_dynamics.py:265-298, pygame_rendering.py).

CONTAINER-ONLY TOOLING for ``gen_car_obs_golden.py`` (see ``_ref_standins.py``); nothing here is
imported by the product, the tests or the bench.

What the stand-ins assert about the two third-party packages is the part of ``car_obs.npz`` that is
NOT pinned by the reference itself [all from memory of the published sources]:

* pygame 1.9.6 (setup.py:7): ``Surface.fill`` / ``subsurface`` / ``blit`` / ``get_rect().center``;
  float -> int conversions truncate toward zero (``Rect`` from floats, polygon points, blit position,
  colour components); ``draw.polygon`` = draw.c ``draw_fillpoly`` (integer scanline crossings
  ``(y - y1) * (x2 - x1) / (y2 - y1) + x1`` with C division, edges counted for ``y1 <= y < y2``, on the
  last scanline also ``y1 < y <= y2``, sorted, inclusive spans); ``draw.rect`` = that polygon over
  ``(l, t), (r, t), (r, b), (l, b)`` with ``r = x + w - 1``, ``b = y + h - 1``; ``transform.rotate`` =
  transform.c ``surf_rotate`` (angle parsed as a C float; multiples of 90 go through ``rotate90``;
  otherwise the destination is the int-truncated bounding box and every destination pixel is
  inverse-mapped in 16.16 fixed point, nearest neighbour, background = the source's first pixel);
  ``font.render(text, False, colour)`` = 1-bit glyphs (taken from the PIL-baked atlas the product
  ships, competitive_rl_amd/assets/car_reward_text.npz).
* box2d-py 2.3.x: ``b2Vec2`` / ``b2Transform`` arithmetic in float32 (``b2Mul(T, v) = (q.c v.x - q.s v.y) + p.x, ...``),
  ``b2Rot(angle) = (sinf(angle), cosf(angle))`` -- evaluated with THIS host's libm through ctypes,
  like the real extension module would.
"""
import ctypes
import math
import os
import sys
import types

import numpy as np

import _ref_standins as S

_libm = ctypes.CDLL("libm.so.6")
_libm.sinf.restype = _libm.cosf.restype = ctypes.c_float
_libm.sinf.argtypes = _libm.cosf.argtypes = [ctypes.c_float]
f32 = np.float32


# ------------------------------------------------------------------------------------ Box2D (float32 value types)
class Vec2:
    """b2Vec2: two float32; every operation rounds to float32 once per arithmetic step."""

    __slots__ = ("x", "y")

    def __init__(self, x=0.0, y=None):
        if y is None:
            x, y = x[0], x[1]
        self.x, self.y = f32(x), f32(y)

    def __getitem__(self, i):  # what pygame / the reference read: python floats holding float32 values
        return float((self.x, self.y)[i])

    def __len__(self):
        return 2

    def __iter__(self):
        return iter((float(self.x), float(self.y)))

    def __add__(self, o):
        o = o if isinstance(o, Vec2) else Vec2(o)
        return Vec2(self.x + o.x, self.y + o.y)

    def __sub__(self, o):
        o = o if isinstance(o, Vec2) else Vec2(o)
        return Vec2(self.x - o.x, self.y - o.y)

    def __rmul__(self, a):  # float * b2Vec2
        a = f32(a)
        return Vec2(self.x * a, self.y * a)

    __mul__ = __rmul__

    def __neg__(self):
        return Vec2(-self.x, -self.y)


class Transform:
    """b2Transform: position p and rotation q = (sinf(angle), cosf(angle))."""

    def __init__(self):
        self.p = Vec2(0, 0)
        self.s, self.c = f32(0), f32(1)
        self._angle = f32(0)

    position = property(lambda s: s.p, lambda s, v: setattr(s, "p", v if isinstance(v, Vec2) else Vec2(v)))

    def _set_angle(self, a):
        self._angle = f32(a)
        self.s, self.c = f32(_libm.sinf(self._angle)), f32(_libm.cosf(self._angle))

    angle = property(lambda s: float(s._angle), _set_angle)

    def __mul__(self, v):  # b2Mul(T, v)
        v = v if isinstance(v, Vec2) else Vec2(v)
        return Vec2((self.c * v.x - self.s * v.y) + self.p.x, (self.s * v.x + self.c * v.y) + self.p.y)


class Shape:
    def __init__(self, vertices=None):
        self._v = [Vec2(v) for v in (vertices or [])]

    vertices = property(lambda s: [(float(v.x), float(v.y)) for v in s._v], lambda s, vs: setattr(s, "_v", [Vec2(v) for v in vs]))


class FixtureDef:
    def __init__(self, shape=None, **kw):
        self.shape = shape
        self.__dict__.update(kw)


class Fixture:
    def __init__(self, shape, body):
        self.shape = Shape(shape.vertices)
        self.body = body
        self.sensor = False


class Body:
    def __init__(self, position=(0, 0), angle=0.0, fixtures=None):
        self._t = Transform()
        self._t.position = position
        self._t.angle = angle
        self.linearVelocity = Vec2(0, 0)
        self.angularVelocity = 0.0
        fx = fixtures if isinstance(fixtures, list) else [fixtures]
        self.fixtures = [Fixture(f.shape, self) for f in fx if f is not None]
        self.userData = None

    position = property(lambda s: s._t.p, lambda s, v: setattr(s._t, "position", v))
    angle = property(lambda s: s._t.angle, lambda s, v: setattr(s._t, "angle", v))
    transform = property(lambda s: s._t)

    def set_pose(self, x, y, a):
        self._t.position = (x, y)
        self._t.angle = a


class Joint:
    def __init__(self, a, b):
        self.bodyA, self.bodyB = a, b
        self.motorSpeed = 0.0

    @property
    def angle(self):  # b2RevoluteJoint::GetJointAngle: aB - aA - referenceAngle in float32
        return float(f32(self.bodyB._t._angle - self.bodyA._t._angle) - f32(0))


class World:
    def __init__(self, *a, contactListener=None, **k):
        self.static = []

    def CreateStaticBody(self, fixtures=None):
        b = Body(fixtures=fixtures)
        self.static.append(b)
        return b

    def CreateDynamicBody(self, **kw):
        return Body(position=kw.get("position", (0, 0)), angle=kw.get("angle", 0.0), fixtures=kw.get("fixtures"))

    def CreateJoint(self, jd):
        return Joint(jd.bodyA, jd.bodyB)

    def DestroyBody(self, b):
        pass

    def Step(self, *a):
        pass


def install_box2d():
    b2 = types.ModuleType("Box2D")
    b2.b2World, b2.b2Transform, b2.b2Vec2 = World, Transform, Vec2
    sub = types.ModuleType("Box2D.b2")
    sub.fixtureDef, sub.polygonShape = FixtureDef, Shape
    sub.revoluteJointDef = lambda **kw: types.SimpleNamespace(**kw)
    sub.contactListener = type("contactListener", (), {"__init__": lambda self: None})
    b2.b2 = sub
    sys.modules["Box2D"], sys.modules["Box2D.b2"] = b2, sub


# ------------------------------------------------------------------------------------ pygame (numpy raster)
def _ci(v):
    """C (int) conversion of a Python number: truncation toward zero"""
    return int(v)


class Rect:
    def __init__(self, x, y, w, h):
        self.x, self.y, self.w, self.h = _ci(x), _ci(y), _ci(w), _ci(h)

    center = property(lambda s: (s.x + (s.w >> 1), s.y + (s.h >> 1)))


class Surface:
    """32-bit surface as a (W, H, 3) uint8 array view; a subsurface shares the parent's pixels."""

    def __init__(self, size=None, *a, px=None, **k):
        self.px = np.zeros((_ci(size[0]), _ci(size[1]), 3), np.uint8) if px is None else px
        self.mask = None  # set on text surfaces: only these pixels are copied by blit (colour key)

    size = property(lambda s: (s.px.shape[0], s.px.shape[1]))

    def get_rect(self):
        return Rect(0, 0, *self.size)

    def fill(self, color):
        self.px[:, :] = [_ci(c) for c in color[:3]]

    def subsurface(self, rect):
        r = rect if isinstance(rect, Rect) else Rect(*rect)
        W, H = self.size
        if r.x < 0 or r.y < 0 or r.x + r.w > W or r.y + r.h > H:
            raise ValueError("subsurface rectangle outside surface area")
        return Surface(px=self.px[r.x:r.x + r.w, r.y:r.y + r.h])

    def blit(self, src, dest):
        dx, dy = _ci(dest[0]), _ci(dest[1])
        W, H = self.size
        w, h = src.size
        x0, y0, x1, y1 = max(dx, 0), max(dy, 0), min(dx + w, W), min(dy + h, H)
        if x1 <= x0 or y1 <= y0:
            return
        s = src.px[x0 - dx:x1 - dx, y0 - dy:y1 - dy]
        if src.mask is None:
            self.px[x0:x1, y0:y1] = s
        else:
            m = src.mask[x0 - dx:x1 - dx, y0 - dy:y1 - dy]
            self.px[x0:x1, y0:y1][m] = s[m]


def _hline(surface, color, x1, y, x2):  # draw.c drawhorzlineclip; the clip rect is the whole surface
    W, H = surface.size
    if y < 0 or y >= H:
        return
    if x2 < x1:
        x1, x2 = x2, x1
    x1, x2 = max(x1, 0), min(x2, W - 1)
    if x2 < 0 or x1 >= W:
        return
    surface.px[x1:x2 + 1, y] = color


def _cdiv(a, b):  # C integer division truncates toward zero
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def draw_polygon(surface, color, points, width=0):
    assert width == 0
    color = [_ci(c) for c in color[:3]]
    vx = [_ci(p[0]) for p in points]
    vy = [_ci(p[1]) for p in points]
    n = len(vx)
    miny, maxy = min(vy), max(vy)
    if miny == maxy:
        _hline(surface, color, min(vx), miny, max(vx))
        return
    for y in range(miny, maxy + 1):
        xs = []
        for i in range(n):
            i1, i2 = (n - 1, 0) if i == 0 else (i - 1, i)
            y1, y2 = vy[i1], vy[i2]
            if y1 < y2:
                x1, x2 = vx[i1], vx[i2]
            elif y1 > y2:
                y2, y1 = vy[i1], vy[i2]
                x2, x1 = vx[i1], vx[i2]
            else:
                continue
            if (y1 <= y < y2) or (y == maxy and y1 < y <= y2):
                xs.append(_cdiv((y - y1) * (x2 - x1), y2 - y1) + x1)
        xs.sort()
        for i in range(0, len(xs) - 1, 2):
            _hline(surface, color, xs[i], y, xs[i + 1])


def draw_rect(surface, color, rect, width=0):
    r = rect if isinstance(rect, Rect) else Rect(*rect)
    l, t, rr, b = r.x, r.y, r.x + r.w - 1, r.y + r.h - 1
    draw_polygon(surface, color, [(l, t), (rr, t), (rr, b), (l, b)], width)


def _c_int(v):
    """(int) of a C double"""
    return int(v)


def transform_rotate(surf, angle):
    angle = float(f32(angle))  # PyArg_ParseTuple "f"
    sw, sh = surf.size
    if math.fmod(angle, 90.0) == 0:  # rotate90(surf, (int)angle)
        q = _cdiv(_c_int(angle), 90)
        turns = q - 4 * _cdiv(q, 4)  # C: (angle / 90) % 4
        if turns < 0:
            turns += 4
        src = surf.px
        if turns == 0:
            out = src.copy()
        elif turns == 1:  # dst(x, y) = src(w - 1 - y, x)
            out = np.ascontiguousarray(src[::-1].transpose(1, 0, 2))
        elif turns == 2:
            out = np.ascontiguousarray(src[::-1, ::-1])
        else:  # dst(x, y) = src(y, h - 1 - x)
            out = np.ascontiguousarray(src[:, ::-1].transpose(1, 0, 2))
        return Surface(px=out)
    radangle = angle * .01745329251994329
    sangle, cangle = math.sin(radangle), math.cos(radangle)
    x, y = float(sw), float(sh)
    cx, cy, sx, sy = cangle * x, cangle * y, sangle * x, sangle * y
    nxmax = _c_int(max(abs(cx + sy), abs(cx - sy), abs(-cx + sy), abs(-cx - sy)))
    nymax = _c_int(max(abs(sx + cy), abs(sx - cy), abs(-sx + cy), abs(-sx - cy)))
    bg = surf.px[0, 0].copy()
    # rotate(src, dst, bgcolor, sangle, cangle)
    dcy = nymax // 2
    xd, yd = (sw - nxmax) * 32768, (sh - nymax) * 32768
    isin, icos = _c_int(sangle * 65536), _c_int(cangle * 65536)
    ax = (nxmax << 15) - _c_int(cangle * ((nxmax - 1) << 15))
    ay = (nymax << 15) - _c_int(sangle * ((nxmax - 1) << 15))
    xmaxval, ymaxval = (sw << 16) - 1, (sh << 16) - 1
    yy, xx = np.meshgrid(np.arange(nymax, dtype=np.int64), np.arange(nxmax, dtype=np.int64))  # [x, y] indexing
    dx = (ax + isin * (dcy - yy)) + xd + icos * xx
    dy = (ay - icos * (dcy - yy)) + yd + isin * xx
    inside = (dx >= 0) & (dy >= 0) & (dx <= xmaxval) & (dy <= ymaxval)
    out = np.empty((nxmax, nymax, 3), np.uint8)
    out[:] = bg
    out[inside] = surf.px[(dx >> 16)[inside], (dy >> 16)[inside]]
    return Surface(px=out)


class Font:
    """font.render(text, False, colour): 1-bit glyphs of the five-character reward read-out from the baked atlas"""

    atlas = None

    def __init__(self, path, size):
        self.size = size

    def render(self, text, antialias, color):
        assert not antialias
        if self.size != 5 or Font.atlas is None:
            return Surface((1, 1))
        bits, r_min = Font.atlas
        idx = 3000 if text == "-0000" else int(text) - r_min
        rows = bits[idx]
        s = Surface((32, len(rows)))
        s.px[:, :] = [_ci(c) for c in color[:3]]
        s.mask = np.zeros((32, len(rows)), bool)
        for r, word in enumerate(rows):
            for c in range(32):
                s.mask[c, r] = bool((int(word) >> c) & 1)
        return s


def install_pygame(atlas_path):
    pg = types.ModuleType("pygame")
    pg.Rect, pg.Surface = Rect, Surface
    pg.init = pg.quit = lambda: None
    pg.sprite = types.SimpleNamespace(Sprite=object)
    pg.draw = types.SimpleNamespace(polygon=draw_polygon, rect=draw_rect)
    pg.font = types.SimpleNamespace(Font=Font, init=lambda: None)
    pg.display = types.SimpleNamespace(quit=lambda: None)
    pg.surfarray = types.SimpleNamespace(array3d=lambda s: s.px.copy())
    pg.image = types.SimpleNamespace(load=lambda p: Surface((30, 52)))
    pg.transform = types.SimpleNamespace(rotate=transform_rotate, scale=lambda im, sz: Surface(sz))
    pg.SRCALPHA = 0
    sys.modules["pygame"] = pg
    a = np.load(atlas_path)
    Font.atlas = (a["bits"], int(a["r_min"]))
    return pg


def load_car_reference(seed_stream, atlas_path):
    """Installs the stand-ins and imports the reference's car modules by path -> (car_dynamics, car_racing_multi_players)."""
    S.install()
    install_pygame(atlas_path)
    install_box2d()
    gym = sys.modules["gym"]
    utils = types.ModuleType("gym.utils")
    utils.seeding = types.SimpleNamespace(np_random=lambda seed=None: (seed_stream(seed), seed))
    utils.EzPickle = type("EzPickle", (), {"__init__": lambda self, *a, **k: None})
    gym.utils = utils
    sys.modules["gym.utils"] = utils
    sys.modules["matplotlib"] = types.ModuleType("matplotlib")
    sys.modules["matplotlib.pyplot"] = types.ModuleType("matplotlib.pyplot")
    Box = sys.modules["gym.spaces"].Box

    def box_init(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.asarray(low).shape
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.low, self.high = np.broadcast_to(low, self.shape).astype(dtype), np.broadcast_to(high, self.shape).astype(dtype)

    Box.__init__ = box_init
    sys.modules["gym.spaces"].Dict.__getitem__ = lambda self, k: self.spaces[k]
    S.load_ref("competitive_rl.car_racing.pygame_rendering", "car_racing/pygame_rendering.py")
    cd = S.load_ref("competitive_rl.car_racing.car_dynamics", "car_racing/car_dynamics.py")
    cr = S.load_ref("competitive_rl.car_racing.car_racing_multi_players", "car_racing/car_racing_multi_players.py")
    return cd, cr
