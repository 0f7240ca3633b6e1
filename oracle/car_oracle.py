"""ctypes front-end of the CarRacing CPU oracle (oracle/car_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from . import pong_oracle as _po

MAX_TILES = 512

TRACK_DT = np.dtype([("n", "<i4"), ("pad", "<i4"), ("track", "<f8", (MAX_TILES, 4)), ("tile", "<f8", (MAX_TILES, 5, 2)),
                     ("border_poly", "<f8", (MAX_TILES, 4, 2)), ("border", "u1", (MAX_TILES,))])
BODY_DT = np.dtype([(k, "<f4") for k in ("cx", "cy", "a", "vx", "vy", "w", "fx", "fy")])
CAR_DT = np.dtype([("hull", BODY_DT), ("wheel", BODY_DT, (4,)), ("imp", "<f4", (4, 3)), ("motor_imp", "<f4", (4,)),
                   ("motor_speed", "<f4", (4,)), ("limit_state", "<i4", (4,)), ("gas", "<f8", (4,)), ("brake", "<f8", (4,)),
                   ("steer", "<f8", (4,)), ("phase", "<f8", (4,)), ("omega", "<f8", (4,)), ("sleep_time", "<f4", (5,)), ("pad_s", "<f4")])
CONTACT_DT = np.dtype([("pair", "<i4"), ("count", "<i4"), ("type", "<i4"), ("ln", "<f4", (2,)), ("lp", "<f4", (2,)),
                       ("pt", "<f4", (2, 2)), ("id", "<u4", (2,)), ("nimp", "<f4", (2,)), ("timp", "<f4", (2,))])
ENV_DT = np.dtype([("trk", TRACK_DT), ("tile32", "<f4", (MAX_TILES, 5, 2)), ("tile_aabb", "<f4", (MAX_TILES, 4)),
                   ("car", CAR_DT, (2,)), ("wheel_tiles", "<u4", (2, 4, MAX_TILES // 32)), ("visited", "<u4", (2, MAX_TILES // 32)),
                   ("tile_visited_count", "<i4", (2,)), ("last_block", "<i4", (2,)), ("done", "<i4", (2,)),
                   ("reward", "<f8", (2,)), ("prev_reward", "<f8", (2,)), ("t", "<f8"), ("step_count", "<i4"), ("inv_dt0", "<f4"),
                   ("n_contact", "<i4"), ("contacts_enabled", "<i4"), ("contact", CONTACT_DT, (8,))],
                  align=True)
CONSTS_DT = np.dtype([("hull_poly", "<f4", (4, 8, 2)), ("hull_n", "<i4", (4,)), ("wheel_poly", "<f4", (4, 2)),
                      ("hull_mass", "<f4"), ("hull_inv_mass", "<f4"), ("hull_I", "<f4"), ("hull_inv_I", "<f4"), ("hull_lc", "<f4", (2,)),
                      ("wheel_mass", "<f4"), ("wheel_inv_mass", "<f4"), ("wheel_I", "<f4"), ("wheel_inv_I", "<f4"),
                      ("anchor", "<f4", (4, 2))])

MAP_ORG, MAP_W = 4392, 1216  # window of the 10000 x 10000 observation map the oracle keeps (car_oracle.h)
PAL_GRAY = np.array([161, 176, 101, 103, 107, 255, 76, 0], np.uint8)

_libs = {}


def _configure(L):
    vp, i32, f64 = C.c_void_p, C.c_int, C.c_double
    L.car_oracle_create_track.argtypes = [vp, vp]
    L.car_oracle_create_track.restype = i32
    L.car_oracle_consts.restype = vp
    L.car_oracle_process_action.argtypes = [vp, vp]
    L.car_oracle_controls.argtypes = [vp, f64, f64, f64]
    L.car_oracle_wheel.argtypes = [f64] * 9 + [i32, vp, vp, vp, vp]
    L.car_oracle_place.argtypes = [vp, f64, f64, f64, i32]
    L.car_oracle_reset.argtypes = [vp, vp, i32, i32]
    L.car_oracle_reset.restype = i32
    L.car_oracle_step.argtypes = [vp, vp, vp, vp]
    L.car_oracle_step_repeat.argtypes = [vp, vp, i32, vp, vp]
    L.car_oracle_step_batch.argtypes = [vp, C.c_long, vp, vp, vp]
    L.car_oracle_collide_batch.argtypes = [vp, C.c_long]
    L.car_oracle_world_step.argtypes = [vp]
    L.car_oracle_contact_event.argtypes = [vp, i32, i32, i32, i32]
    L.car_oracle_hull_position.argtypes = [vp, i32, vp]
    L.car_oracle_wheel_on_road.argtypes = [vp, i32, i32]
    L.car_oracle_wheel_on_road.restype = i32
    L.car_oracle_build_map.argtypes = [vp, vp, i32, i32]
    L.car_oracle_build_map.restype = C.c_long
    L.car_oracle_map_vertices.argtypes = [vp, vp]
    L.car_oracle_view_sources.argtypes = [vp, i32, vp, vp]
    L.car_oracle_render.argtypes = [vp, vp, i32, i32, i32, vp]
    L.car_oracle_render_analytic.argtypes = [vp, i32, vp]
    L.car_oracle_set_text.argtypes = [vp]
    L.car_oracle_f64.argtypes = [i32, vp, vp, vp, C.c_long]
    L.car_oracle_env_size.restype = i32
    assert L.car_oracle_env_size() == ENV_DT.itemsize, (L.car_oracle_env_size(), ENV_DT.itemsize)
    return L


def lib(libm=False):
    """liboracle.so (sin / cos / atan2 shared with the HIP kernels); libm=True: liboracle_libm.so (the host libm, like the
    reference); libm="fma": liboracle_fma.so (liboracle.so with the island solver's iterations in fused multiply-adds, the
    checker of CRL_FLAG_CAR_FMA contexts).  Same sources, see oracle/Makefile."""
    key = libm if libm in ("fma", "norb") else bool(libm)
    if key not in _libs:
        L = _po.lib()  # builds the first three
        if key == "norb":  # the default build without the wheel joints' exactly-zero rB terms (oracle/Makefile): a counting variant
            import os
            import subprocess

            subprocess.check_call(["make", "-C", _po.HERE, "norb"], stdout=subprocess.DEVNULL)
            L = C.CDLL(os.path.join(_po.HERE, "liboracle_norb.so"))
        _libs[key] = _configure(L if key in (False, "norb") else C.CDLL(_po.LIB_FMA) if key == "fma" else C.CDLL(_po.LIB_LIBM))
        assert _libs[key].car_oracle_fma() == (1 if key == "fma" else 0)
    return _libs[key]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


_text = None


def set_text(bits):
    """Reward read-out bitmaps u32 [3001, 10] (kept alive here); None turns the text off."""
    global _text
    _text = None if bits is None else np.ascontiguousarray(bits, np.uint32)
    for libm in (False, True, "fma"):
        lib(libm).car_oracle_set_text(None if _text is None else _p(_text))


def f64(fn, a, b=None, libm=False):
    """sin (fn=0) / cos (1) / atan2(a, b) (2) as the chosen oracle build evaluates them."""
    a = np.ascontiguousarray(a, np.float64)
    b = np.zeros_like(a) if b is None else np.ascontiguousarray(b, np.float64)
    out = np.zeros_like(a)
    lib(libm).car_oracle_f64(int(fn), _p(a), _p(b), _p(out), a.size)
    return out


def consts():
    ptr = lib().car_oracle_consts()
    return np.frombuffer((C.c_char * CONSTS_DT.itemsize).from_address(ptr), CONSTS_DT)[0]


def create_track(u24, libm=False):
    u = np.ascontiguousarray(u24, np.float64)
    trk = np.zeros(1, TRACK_DT)
    ok = lib(libm).car_oracle_create_track(_p(u), _p(trk))
    return bool(ok), trk[0]


def process_action(a):
    a = np.ascontiguousarray(a, np.float64)
    out = np.zeros(3)
    lib().car_oracle_process_action(_p(a), _p(out))
    return out


def wheel(dt, steer, gas, brake, joint_angle, q_sin, q_cos, vx, vy, on_road, omega, phase):
    om, ph, ms, f = np.array([omega], np.float64), np.array([phase], np.float64), np.zeros(1), np.zeros(2)
    lib().car_oracle_wheel(dt, steer, gas, brake, joint_angle, q_sin, q_cos, vx, vy, int(on_road), _p(om), _p(ph), _p(ms), _p(f))
    return om[0], ph[0], ms[0], f


class CarEnv:
    """One cCarRacingDouble env (2 cars).  libm=True: the build that calls the host libm (oracle/Makefile)."""

    def __init__(self, libm=False):
        self.buf = np.zeros(1, ENV_DT)
        self.e = self.buf[0]
        self.L = lib(libm)
        self._map = None
        self._map_key = None

    def reset(self, u, shuffle_swap=0):
        u = np.ascontiguousarray(u, np.float64).reshape(-1)
        self._map = None
        return self.L.car_oracle_reset(_p(self.buf), _p(u), len(u) // 24, int(shuffle_swap))

    def step(self, actions):
        rew, done = np.zeros(2), np.zeros(2, np.int32)
        if actions is None:
            self.L.car_oracle_step(_p(self.buf), None, _p(rew), _p(done))
        else:
            a = np.ascontiguousarray(actions, np.float64).reshape(2, 2)
            self.L.car_oracle_step(_p(self.buf), _p(a), _p(rew), _p(done))
        return rew, done

    def world_step(self):
        """world.Step alone (Collide of the cars + island solve), no Car.step"""
        self.L.car_oracle_world_step(_p(self.buf))

    def step_repeat(self, actions, repeat):
        rew, done = np.zeros(2), np.zeros(2, np.int32)
        a = np.ascontiguousarray(actions, np.float64).reshape(2, 2)
        self.L.car_oracle_step_repeat(_p(self.buf), _p(a), int(repeat), _p(rew), _p(done))
        return rew, done

    def contact_event(self, c, w, t, begin):
        self.L.car_oracle_contact_event(_p(self.buf), c, w, t, int(begin))

    def wheel_on_road(self, c, w):
        return bool(self.L.car_oracle_wheel_on_road(_p(self.buf), c, w))

    def build_map(self, org=MAP_ORG, w=MAP_W):
        """render_road_for_observation_map: palette map (w, w) u8 of the window [org, org + w)^2 and the number
        of polygon pixels that fell outside it."""
        m = np.zeros((w, w), np.uint8)
        dropped = self.L.car_oracle_build_map(_p(self.buf), _p(m), int(org), int(w))
        return m, int(dropped)

    def map(self):
        """The windowed map of the current track (cached until the next reset / invalidate_map())."""
        if self._map is None:
            self._map, dropped = self.build_map()
            assert dropped == 0, "a track polygon leaves the map window"
        return self._map

    def invalidate_map(self):
        self._map = None

    def map_vertices(self):
        n = int(self.e["trk"]["n"])
        out = np.zeros((n, 9, 2), np.int32)
        self.L.car_oracle_map_vertices(_p(self.buf), _p(out))
        return out

    def view_sources(self, viewer):
        src, rect = np.zeros((96, 96, 2), np.int32), np.zeros(2, np.int32)
        self.L.car_oracle_view_sources(_p(self.buf), int(viewer), _p(src), _p(rect))
        return src, rect

    def render(self, viewer, map_=None, org=MAP_ORG):
        """CarRacing.get_observation(viewer): (96, 96) u8"""
        m = self.map() if map_ is None else map_
        out = np.zeros((96, 96), np.uint8)
        self.L.car_oracle_render(_p(self.buf), _p(m), int(org), int(m.shape[0]), int(viewer), _p(out))
        return out

    def render_analytic(self, viewer):
        """rounds 1-2: background classified analytically at pixel centres (kept for comparison)"""
        out = np.zeros((96, 96), np.uint8)
        self.L.car_oracle_render_analytic(_p(self.buf), int(viewer), _p(out))
        return out

    def hull_position(self, c):
        out = np.zeros(3, np.float32)
        self.L.car_oracle_hull_position(_p(self.buf), c, _p(out))
        return out


class CarBatch:
    """n oracle envs in ONE structured array, stepped together over the host's cores (car_oracle_step_batch): the long parity
    soaks.  ``self.E[i]`` is env i's car_env; ``view(i)`` a CarEnv over the same memory (render, reset, map)."""

    def __init__(self, n, libm=False):
        self.n = int(n)
        self.E = np.zeros(self.n, ENV_DT)
        self.L = lib(libm)
        self.libm = libm
        self._views = {}

    def view(self, i):
        v = self._views.get(i)
        if v is None:
            v = CarEnv.__new__(CarEnv)
            v.buf, v.L, v._map, v._map_key = self.E[i:i + 1], self.L, None, None
            v.e = v.buf[0]
            self._views[i] = v
        return v

    def step(self, actions):
        """actions (n, 2, 2) or None (the action-less step a reset ends with) -> step rewards (n, 2) f64, per-car done (n, 2) i32"""
        rew, done = np.zeros((self.n, 2)), np.zeros((self.n, 2), np.int32)
        a = None if actions is None else np.ascontiguousarray(actions, np.float64).reshape(self.n, 2, 2)
        self.L.car_oracle_step_batch(_p(self.E), self.n, None if a is None else _p(a), _p(rew), _p(done))
        return rew, done


def collide_variant(which):
    """liboracle_b2v<which>.so: the oracle with another published form of b2CollidePolygons' edge search (car_oracle.c
    CRL_B2_COLLIDE = 1: Box2D 2.3.1+, 2: Box2D 2.3.0); built on demand (make -C oracle variants).  Only for counting disagreements."""
    import os
    import subprocess

    key = ("b2v", int(which))
    if key not in _libs:
        subprocess.check_call(["make", "-C", _po.HERE, "variants"], stdout=subprocess.DEVNULL)
        L = _configure(C.CDLL(os.path.join(_po.HERE, "liboracle_b2v%d.so" % int(which))))
        assert L.car_oracle_collide_variant() == int(which)
        _libs[key] = L
    return _libs[key]
