"""ctypes front-end of the CPU oracle (oracle/pong_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# CRL_ORACLE_SUFFIX=_asan: the sanitizer builds of oracle/Makefile (`make asan`), for tests/test_oracle_sanitizers.py
_SUFFIX = os.environ.get("CRL_ORACLE_SUFFIX", "")
LIB = os.path.join(HERE, "liboracle%s.so" % _SUFFIX)
LIB_LIBM = os.path.join(HERE, "liboracle_libm%s.so" % _SUFFIX)  # same sources with -DCRL_LIBM: the host libm's sin / cos / atan2
LIB_FMA = os.path.join(HERE, "liboracle_fma.so")  # -DCRL_FMA: the island solver's iterations in fused multiply-adds (CRL_FLAG_CAR_FMA contexts)

RAW, GRAY = 0, 1

FRAME_DT = np.dtype([("ball_x", "<i2"), ("ball_y", "<i2"), ("bat_l_y", "u1"), ("bat_r_y", "u1"),
                     ("score_l", "u1"), ("score_r", "u1")])
STATE_DT = np.dtype([
    ("speed_x", "<f8"), ("speed_y", "<f8"), ("ball_x", "<i4"), ("ball_y", "<i4"),
    ("bat_l_y", "<i4"), ("bat_r_y", "<i4"), ("score_l", "<i4"), ("score_r", "<i4"),
    ("num_rounds", "<i4"), ("num_steps", "<i4"), ("serve_ctr", "<u4"), ("wrap_steps", "<i4"),
    ("keep", FRAME_DT, (2,)), ("hist", FRAME_DT, (3, 2)),
])
assert STATE_DT.itemsize == 120 and FRAME_DT.itemsize == 8


def build(force=False):
    """(Re)builds liboracle.so / liboracle_libm.so when the sources changed.  Staleness is decided by a content hash kept
    beside each library (file times do not survive a copy of the tree), and the whole check runs under a file lock: the
    subprocess CPU baseline starts one worker per host thread, and all of them come through here at once."""
    import fcntl
    import hashlib
    inc = os.path.join(os.path.dirname(HERE), "include")
    srcs = [os.path.join(HERE, f) for f in ("pong_oracle.c", "car_oracle.c", "car_oracle.h", "Makefile")]
    srcs += [os.path.join(inc, f) for f in ("crl.h", "crl_rot.h", "crl_f64.h")]
    h = hashlib.sha256()
    for f in srcs:
        h.update(open(f, "rb").read())
    want = h.hexdigest()
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        for lib_ in (LIB, LIB_LIBM) + (() if _SUFFIX else (LIB_FMA,)):
            stamp = lib_ + ".stamp"
            have = open(stamp).read().strip() if os.path.exists(stamp) else ""
            if force or not os.path.exists(lib_) or have != want:
                subprocess.check_call(["make", "-C", HERE, "-B", os.path.basename(lib_)], stdout=subprocess.DEVNULL)
                with open(stamp, "w") as f:
                    f.write(want)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        vp, i64, u64, i32 = C.c_void_p, C.c_int64, C.c_uint64, C.c_int
        L.pong_oracle_create.restype = vp
        L.pong_oracle_create.argtypes = [i64, i32, i32, i32, u64, i64, vp]
        L.pong_oracle_destroy.argtypes = [vp]
        L.pong_oracle_set_threads.argtypes = [vp, i32]
        L.pong_oracle_seed.argtypes = [vp, u64]
        L.pong_oracle_set_mode.argtypes = [vp, i32, i32]
        L.pong_oracle_set_replay.argtypes = [vp, vp, vp, vp, i64]
        L.pong_oracle_state.restype = vp
        L.pong_oracle_state.argtypes = [vp]
        L.pong_oracle_real_reward.restype = vp
        L.pong_oracle_real_reward.argtypes = [vp]
        L.pong_oracle_num_steps.restype = vp
        L.pong_oracle_num_steps.argtypes = [vp]
        L.pong_oracle_terminal_frames.restype = vp
        L.pong_oracle_terminal_frames.argtypes = [vp]
        L.pong_oracle_reset.argtypes = [vp, vp]
        L.pong_oracle_step.argtypes = [vp, vp, vp, vp, vp]
        L.pong_oracle_terminal_observation.argtypes = [vp, i64, vp]
        L.pong_oracle_render_raw.argtypes = [vp, i64, vp, vp]
        L.pong_oracle_render_gray.argtypes = [vp, vp, vp, i32, i32, vp]
        L.pong_oracle_render_gray_f32.argtypes = [vp, vp, vp, i32, i32, i32, vp]
        L.pong_oracle_set_f32ref.argtypes = [vp, i32]
        L.pong_oracle_area_table.restype = i32
        L.pong_oracle_area_table.argtypes = [i32, i32, vp, vp, vp]
        L.pong_oracle_state_size.restype = i32
        assert L.pong_oracle_state_size() == STATE_DT.itemsize
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def area_table(ssize, dsize):
    di = np.zeros(2 * ssize, np.int32)
    si = np.zeros(2 * ssize, np.int32)
    al = np.zeros(2 * ssize, np.float32)
    n = lib().pong_oracle_area_table(ssize, dsize, _p(di), _p(si), _p(al))
    return di[:n], si[:n], al[:n]


def render_raw(frames, atlas):
    frames = np.ascontiguousarray(frames, dtype=FRAME_DT)
    out = np.empty((len(frames), 2, 210, 160, 3), np.uint8)
    lib().pong_oracle_render_raw(_p(frames), len(frames), _p(atlas), _p(out))
    return out


def render_gray_f32(fa, fb, atlas, view, R, rounded=False):
    """the reference's float32 step path for one frame pair (unrounded), or the uint8 path as floats (rounded=True)"""
    fa = np.ascontiguousarray(fa, dtype=FRAME_DT).reshape(1)
    fb = np.ascontiguousarray(fb, dtype=FRAME_DT).reshape(1)
    out = np.empty((R, R), np.float32)
    lib().pong_oracle_render_gray_f32(_p(fa), _p(fb), _p(atlas), view, R, int(rounded), _p(out))
    return out


def render_gray(fa, fb, atlas, view, R):
    fa = np.ascontiguousarray(fa, dtype=FRAME_DT).reshape(1)
    fb = np.ascontiguousarray(fb, dtype=FRAME_DT).reshape(1)
    out = np.empty((R, R), np.uint8)
    lib().pong_oracle_render_gray(_p(fa), _p(fb), _p(atlas), view, R, _p(out))
    return out


class PongOracle:
    """Batch of cPongDouble envs with VecEnv.step/reset semantics (auto-reset)."""

    def __init__(self, num_envs, atlas, obs_mode=RAW, resized_dim=84, frame_stack=1, seed=0, env_id_base=0,
                 single=False, replicate=False, obs_dtype="uint8"):
        self.n, self.mode, self.R, self.K = int(num_envs), obs_mode, int(resized_dim), int(frame_stack)
        self.atlas = np.ascontiguousarray(atlas, np.uint8)
        assert self.atlas.size == 22 * 22 * 34 * 160
        self.h = lib().pong_oracle_create(self.n, obs_mode, self.R, self.K, seed, env_id_base, _p(self.atlas))
        self._replay = None
        self.single, self.V = bool(single), 1 if single else 2
        lib().pong_oracle_set_mode(self.h, int(single), int(replicate))
        shape = (self.n, self.V, 210, 160, 3) if obs_mode == RAW else (self.n, self.V, self.K, self.R, self.R)
        assert obs_dtype in ("uint8", "float32_ref")
        self.f32ref = obs_dtype == "float32_ref"  # the reference's unrounded float32 step path (pong_oracle_render_gray_f32)
        assert not (self.f32ref and obs_mode == RAW)
        lib().pong_oracle_set_f32ref(self.h, int(self.f32ref))
        self.obs = np.zeros(shape, np.float32 if self.f32ref else np.uint8)
        self.rew = np.zeros((self.n,) if single else (self.n, 2), np.float32)
        self.done = np.zeros((self.n,), np.uint8)

    def close(self):
        if self.h and lib is not None:  # (module globals are gone when this runs as __del__ at interpreter exit)
            lib().pong_oracle_destroy(self.h)
            self.h = None

    __del__ = close

    def set_threads(self, t):
        lib().pong_oracle_set_threads(self.h, int(t))

    def seed(self, seed):
        lib().pong_oracle_seed(self.h, int(seed))

    def set_replay(self, u, bx, by):
        """u, bx, by: arrays [n, per_env]."""
        u = np.ascontiguousarray(u, np.float64).reshape(self.n, -1)
        bx = np.ascontiguousarray(bx, np.uint8).reshape(self.n, -1)
        by = np.ascontiguousarray(by, np.uint8).reshape(self.n, -1)
        self._replay = (u, bx, by)  # keep alive
        lib().pong_oracle_set_replay(self.h, _p(u), _p(bx), _p(by), u.shape[1])

    @property
    def state(self):
        """Live structured view of the oracle's state array (writable)."""
        ptr = lib().pong_oracle_state(self.h)
        buf = (C.c_char * (self.n * STATE_DT.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=STATE_DT)

    @property
    def real_reward(self):
        ptr = lib().pong_oracle_real_reward(self.h)
        return np.frombuffer((C.c_float * (2 * self.n)).from_address(ptr), np.float32).reshape(self.n, 2)

    @property
    def num_steps(self):
        ptr = lib().pong_oracle_num_steps(self.h)
        return np.frombuffer((C.c_int32 * self.n).from_address(ptr), np.int32)

    @property
    def terminal_frames(self):
        ptr = lib().pong_oracle_terminal_frames(self.h)
        return np.frombuffer((C.c_char * (16 * self.n)).from_address(ptr), FRAME_DT).reshape(self.n, 2)

    def reset(self, render=True):
        lib().pong_oracle_reset(self.h, _p(self.obs) if render else None)
        return self.obs

    def step(self, actions, render=True):
        a = np.ascontiguousarray(actions, np.int32).reshape((self.n,) if self.single else (self.n, 2))
        lib().pong_oracle_step(self.h, _p(a), _p(self.obs) if render else None, _p(self.rew), _p(self.done))
        return self.obs, self.rew, self.done

    def terminal_observation(self, i):
        shape = (self.V, 210, 160, 3) if self.mode == RAW else (self.V, self.R, self.R)
        out = np.empty(shape, np.float32 if self.f32ref else np.uint8)
        lib().pong_oracle_terminal_observation(self.h, int(i), _p(out))
        return out


def f32ref_ambiguous():
    """float32_ref observations so far for which "reset observation" (the explicit flag the oracle rounds by) and "the two kept frames
    are identical" (what the HIP kernel infers it from) disagreed -- must stay 0."""
    L = lib()
    L.pong_oracle_f32ref_ambiguous.restype = C.c_long
    return int(L.pong_oracle_f32ref_ambiguous())
