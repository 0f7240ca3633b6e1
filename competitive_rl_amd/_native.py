"""ctypes binding of libcrl_hip.so (include/crl.h).  There is NO fallback: if the HIP
library is missing or fails to load, importing the backend raises."""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
# CRL_LIB_VARIANT=<tag>: a profiling build made by `python -m competitive_rl_amd.build --variant <tag> <flags>` (e.g. -DCRL_ABLATION:
# phase cycle stamps, timing ablations); unset = the shipped library
LIB_PATH = os.path.join(PKG, "libcrl_hip_%s.so" % os.environ["CRL_LIB_VARIANT"] if os.environ.get("CRL_LIB_VARIANT") else "libcrl_hip.so")

CRL_ENV_PONG_DOUBLE, CRL_ENV_CAR_DOUBLE, CRL_ENV_PONG_SINGLE, CRL_ENV_CAR_SINGLE = 1, 2, 3, 4
CRL_FLAG_STACK_REPLICATE = 1
CAR_MAX_TILES = 512
CAR_MAP_ORG, CAR_MAP_W = 4392, 1216  # include/crl.h CRL_CAR_MAP_*
CRL_OBS_RAW_RGB, CRL_OBS_GRAY_RESIZED = 0, 1
PONG_FRAME_BYTES = 210 * 160 * 3
ATLAS_BYTES = 22 * 22 * 34 * 160

# every symbol include/crl.h declares (tests check the library exports all of them)
SYMBOLS = ["crl_create", "crl_destroy", "crl_seed", "crl_reset", "crl_step", "crl_render", "crl_info", "crl_copy_info",
           "crl_terminal_observation", "crl_get_state", "crl_set_state", "crl_set_replay", "crl_render_raw",
           "crl_obs_bytes_per_env", "crl_kernel_timing", "crl_kernel_time_ms", "crl_last_error", "crl_version",
           "crl_car_get_state", "crl_car_set_state", "crl_car_get_track", "crl_car_set_track", "crl_car_get_map", "crl_car_set_replay",
           "crl_policy_create", "crl_policy_create_full", "crl_policy_destroy", "crl_policy_reset", "crl_policy_act", "crl_policy_get_stack",
           "crl_policy_set_stack", "crl_terminal_observation_dev", "crl_check", "crl_car_info", "crl_car_copy_info", "crl_frame_stack_update", "crl_frame_stack_update_to", "crl_frame_stack_update_u8", "crl_ctx_last_error",
           "crl_obs_descriptors", "crl_render_frames_dev", "crl_car_cap_hits", "crl_selftest_sincosf", "crl_step_stack", "crl_draw_stack", "crl_set_flags_event", "crl_kernel_time_stats"]

FRAME_DT = np.dtype([("ball_x", "<i2"), ("ball_y", "<i2"), ("bat_l_y", "u1"), ("bat_r_y", "u1"),
                     ("score_l", "u1"), ("score_r", "u1")])
STATE_DT = np.dtype([
    ("speed_x", "<f8"), ("speed_y", "<f8"), ("ball_x", "<i4"), ("ball_y", "<i4"),
    ("bat_l_y", "<i4"), ("bat_r_y", "<i4"), ("score_l", "<i4"), ("score_r", "<i4"),
    ("num_rounds", "<i4"), ("num_steps", "<i4"), ("serve_ctr", "<u4"), ("wrap_steps", "<i4"),
    ("keep", FRAME_DT, (2,)), ("hist", FRAME_DT, (3, 2)),
])


CAR_BODY_DT = np.dtype([(k, "<f4") for k in ("cx", "cy", "a", "vx", "vy", "w")])
CAR_STATE_DT = np.dtype([
    ("hull", CAR_BODY_DT), ("wheel", CAR_BODY_DT, (4,)), ("imp", "<f4", (4, 3)), ("motor_imp", "<f4", (4,)),
    ("motor_speed", "<f4", (4,)), ("limit_state", "<i4", (4,)), ("gas", "<f8", (4,)), ("omega", "<f8", (4,)), ("phase", "<f8", (4,)),
    ("reward", "<f8"), ("prev_reward", "<f8"), ("tile_visited_count", "<i4"), ("last_block", "<i4"), ("done", "<i4"),
    ("step_count", "<i4"), ("first_step", "<i4"), ("pad", "<i4"),
    ("wheel_tiles", "<u4", (4, CAR_MAX_TILES // 32)), ("visited", "<u4", (CAR_MAX_TILES // 32,)),
    ("sleep_time", "<f4", (5,)), ("pad2", "<f4")], align=True)
CAR_CONTACT_DT = np.dtype([("pair", "<i4"), ("count", "<i4"), ("type", "<i4"), ("ln", "<f4", (2,)), ("lp", "<f4", (2,)),
                           ("pt", "<f4", (2, 2)), ("id", "<u4", (2,)), ("nimp", "<f4", (2,)), ("timp", "<f4", (2,))])
CAR_ENV_STATE_DT = np.dtype([("car", CAR_STATE_DT, (2,)), ("elapsed", "<i4"), ("episode", "<u4"), ("n_contact", "<i4"), ("coupled", "<i4"),
                             ("contact", CAR_CONTACT_DT, (8,))], align=True)
CRL_FLAG_CAR_NO_CONTACTS = 2
CRL_FLAG_CAR_FMA = 4
CRL_CAR_DONE_ANY, CRL_CAR_DONE_CAR0 = 0, 1
CRL_OBS_U8, CRL_OBS_F32, CRL_OBS_F32_REF = 0, 1, 2
CRL_EACTION = -5


class CrlOpts(C.Structure):
    _fields_ = [("env_kind", C.c_int32), ("obs_mode", C.c_int32), ("resized_dim", C.c_int32),
                ("frame_stack", C.c_int32), ("num_envs", C.c_int64), ("env_id_base", C.c_int64),
                ("seed", C.c_uint64), ("device", C.c_int32), ("flags", C.c_int32), ("action_repeat", C.c_int32),
                ("done_policy", C.c_int32), ("obs_dtype", C.c_int32), ("reserved", C.c_int32)]


class CrlStackDesc(C.Structure):
    """crl_stack_desc (include/crl.h): a FrameStackTensor drawn by the step."""
    _fields_ = [("stack_dev", C.c_void_p), ("planes", C.c_int32), ("dtype", C.c_int32), ("agent", C.c_int32),
                ("valid_planes", C.c_int32), ("alias_newest", C.c_int32), ("reserved", C.c_int32)]


class CrlError(RuntimeError):
    pass


class CrlActionError(CrlError, AssertionError):
    """An action outside the action space reached the step kernel (the reference's
    ``assert self.action_space.contains(action)``, pong/base_pong_env.py:42)."""


_lib = None


def load():
    """Load libcrl_hip.so or raise -- the product path never silently degrades."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m competitive_rl_amd.build` "
            "(hipcc, gfx950).  competitive_rl_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i64, u64, i32 = C.c_void_p, C.c_int64, C.c_uint64, C.c_int
    L.crl_create.argtypes = [C.POINTER(CrlOpts), vp, C.POINTER(vp)]
    L.crl_destroy.argtypes = [vp]
    L.crl_destroy.restype = None
    L.crl_seed.argtypes = [vp, u64]
    L.crl_reset.argtypes = [vp, vp, vp]
    L.crl_step.argtypes = [vp, vp, vp, vp, vp, vp]
    L.crl_step_stack.argtypes = [vp, vp, vp, vp, vp, C.POINTER(CrlStackDesc), vp]
    L.crl_draw_stack.argtypes = [vp, vp, C.POINTER(CrlStackDesc), vp]
    L.crl_set_flags_event.argtypes = [vp, vp]
    L.crl_info.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.crl_render.argtypes = [vp, vp, vp]
    L.crl_copy_info.argtypes = [vp, vp, vp, vp]
    L.crl_terminal_observation.argtypes = [vp, vp, i64, vp, vp]
    L.crl_terminal_observation_dev.argtypes = [vp, vp, i64, vp, vp]
    L.crl_check.argtypes = [vp, vp]
    L.crl_car_info.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.crl_car_copy_info.argtypes = [vp, vp, vp, vp, vp]
    L.crl_frame_stack_update.argtypes = [vp, vp, i32, i64, vp, i64, i32, i32, i64, vp]
    L.crl_frame_stack_update_to.argtypes = [vp, vp, vp, i32, i64, vp, i64, i32, i32, i64, vp]
    L.crl_frame_stack_update_u8.argtypes = [vp, vp, vp, i32, i64, vp, i64, i32, i32, i64, vp]
    L.crl_get_state.argtypes = [vp, vp, i64, i64, vp]
    L.crl_set_state.argtypes = [vp, vp, i64, i64, vp]
    L.crl_set_replay.argtypes = [vp, vp, vp, vp, i64]
    L.crl_render_raw.argtypes = [vp, vp, i64, vp, vp]
    L.crl_obs_descriptors.argtypes = [vp, vp, vp]
    L.crl_render_frames_dev.argtypes = [vp, vp, i64, vp, vp]
    L.crl_obs_bytes_per_env.argtypes = [vp]
    L.crl_obs_bytes_per_env.restype = i64
    L.crl_kernel_timing.argtypes = [vp, i32]
    L.crl_kernel_time_ms.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(i64)]
    L.crl_kernel_time_stats.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double)]
    L.crl_car_get_state.argtypes = [vp, vp, i64, i64, vp]
    L.crl_car_set_state.argtypes = [vp, vp, i64, i64, vp]
    L.crl_car_get_track.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp]
    L.crl_car_set_track.argtypes = [vp, i64, i32, vp, vp, vp, vp, vp]
    L.crl_car_get_map.argtypes = [vp, i64, vp, vp, vp]
    L.crl_car_cap_hits.argtypes = [vp, vp, vp]
    L.crl_selftest_sincosf.argtypes = [C.c_int32, u64, u64, vp]
    L.crl_car_set_replay.argtypes = [vp, vp, vp, i64]
    L.crl_policy_create.argtypes = [i32, i64, vp, vp, vp, vp, vp, vp, C.POINTER(vp)]
    L.crl_policy_create_full.argtypes = [i32, i64, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(vp)]
    L.crl_policy_destroy.argtypes = [vp]
    L.crl_policy_destroy.restype = None
    L.crl_policy_reset.argtypes = [vp, vp]
    L.crl_policy_act.argtypes = [vp, vp, i64, vp, i64, vp, vp]
    L.crl_policy_get_stack.argtypes = [vp, vp, vp]
    L.crl_policy_set_stack.argtypes = [vp, vp, vp]
    L.crl_last_error.restype = C.c_char_p
    L.crl_ctx_last_error.restype = C.c_char_p
    L.crl_ctx_last_error.argtypes = [vp]
    L.crl_version.restype = C.c_char_p
    for name in SYMBOLS:
        getattr(L, name)
        if name not in ("crl_destroy", "crl_policy_destroy", "crl_obs_bytes_per_env", "crl_last_error", "crl_ctx_last_error", "crl_version"):
            getattr(L, name).restype = i32
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = f"crl error {rc}: {load().crl_last_error().decode()}"
        raise (CrlActionError if rc == CRL_EACTION else CrlError)(msg)


def load_score_atlas():
    a = np.load(os.path.join(PKG, "assets", "pong_score_atlas.npz"))["atlas"]
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.size == ATLAS_BYTES
    return a


def load_car_text():
    b = np.load(os.path.join(PKG, "assets", "car_reward_text.npz"))["bits"]
    b = np.ascontiguousarray(b, dtype=np.uint32)
    assert b.shape == (3001, 10)
    return b
