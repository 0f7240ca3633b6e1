"""CPU-side checks of the C-ABI boundary: the library builds/loads and exports exactly
what include/crl.h declares.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "crl.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(crl_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from competitive_rl_amd import _native as N
    from competitive_rl_amd.build import build

    build()
    lib = ctypes.CDLL(N.LIB_PATH)
    names = header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in crl.h but not exported"
    assert sorted(N.SYMBOLS) == names, "ctypes binding and header disagree"
    lib.crl_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.crl_version()


def test_state_struct_layout_matches_header():
    from competitive_rl_amd import _native as N
    from oracle import pong_oracle as po

    assert N.STATE_DT.itemsize == 120 and N.FRAME_DT.itemsize == 8
    assert N.STATE_DT == po.STATE_DT
    assert po.lib().pong_oracle_state_size() == 120
    assert ctypes.sizeof(N.CrlOpts) == 64  # crl_opts: 48 bytes of round 1 + action_repeat, done_policy, obs_dtype, reserved


def test_create_rejects_bad_arguments_without_gpu():
    from competitive_rl_amd import _native as N

    L = N.load()
    h = ctypes.c_void_p()
    assert L.crl_create(None, None, ctypes.byref(h)) == -1
    assert b"null" in L.crl_last_error()
    atlas = N.load_score_atlas()
    o = N.CrlOpts(env_kind=7, obs_mode=0, resized_dim=0, frame_stack=1, num_envs=4, env_id_base=0, seed=0, device=0)
    assert L.crl_create(ctypes.byref(o), atlas.ctypes.data_as(ctypes.c_void_p), ctypes.byref(h)) == -1
    assert b"env_kind" in L.crl_last_error()
    o.env_kind, o.num_envs = 1, 0
    assert L.crl_create(ctypes.byref(o), atlas.ctypes.data_as(ctypes.c_void_p), ctypes.byref(h)) == -1
    o.num_envs, o.obs_mode, o.resized_dim, o.frame_stack = 4, 1, 85, 1
    assert L.crl_create(ctypes.byref(o), atlas.ctypes.data_as(ctypes.c_void_p), ctypes.byref(h)) == -1
    assert L.crl_step(None, None, None, None, None, None) == -1
    # options that belong to the other env family, or an unknown policy, are refused before any GPU call
    o = N.CrlOpts(env_kind=1, obs_mode=0, resized_dim=0, frame_stack=1, num_envs=4, action_repeat=4)
    assert L.crl_create(ctypes.byref(o), atlas.ctypes.data_as(ctypes.c_void_p), ctypes.byref(h)) == -1
    assert b"CarRacing options" in L.crl_last_error()
    o = N.CrlOpts(env_kind=2, num_envs=4, frame_stack=1, action_repeat=84)
    assert L.crl_create(ctypes.byref(o), None, ctypes.byref(h)) == -1 and b"action_repeat" in L.crl_last_error()
    o = N.CrlOpts(env_kind=2, num_envs=4, frame_stack=1, done_policy=3)
    assert L.crl_create(ctypes.byref(o), None, ctypes.byref(h)) == -1 and b"done_policy" in L.crl_last_error()
    o = N.CrlOpts(env_kind=1, obs_mode=0, num_envs=4, frame_stack=1, obs_dtype=1)
    assert L.crl_create(ctypes.byref(o), atlas.ctypes.data_as(ctypes.c_void_p), ctypes.byref(h)) == -1 and b"obs_dtype" in L.crl_last_error()
    o = N.CrlOpts(env_kind=1, obs_mode=0, num_envs=4, frame_stack=1, reserved=9)
    assert L.crl_create(ctypes.byref(o), atlas.ctypes.data_as(ctypes.c_void_p), ctypes.byref(h)) == -1 and b"reserved" in L.crl_last_error()
    assert L.crl_frame_stack_update(None, None, 0, 0, None, 1, 1, 1, 1, None) == -1
    assert L.crl_check(None, None) == -1 and L.crl_car_info(None, None, None) == -1


def test_product_has_no_oracle_dependency():
    """The product package must not import or link the oracle."""
    pkg = os.path.join(ROOT, "competitive_rl_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn)).read()
                assert "oracle" not in txt.lower() or fn == "gen_score_atlas.py", f"{fn} mentions the oracle"


def test_env_without_gpu_fails_loudly():
    import pytest
    import torch

    import competitive_rl_amd as crl

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        crl.make_envs("cPongDouble-v0", num_envs=2, frame_stack=None, log_dir=None)


def test_c_abi_demo_links_against_the_library():
    """examples/c_abi_demo.cpp (no Python, no torch) compiles against include/crl.h and links
    libcrl_hip.so; running it needs a GPU (tests/test_hip_pong_parity.py)."""
    from competitive_rl_amd.build import build_c_demo

    exe = build_c_demo()
    assert os.path.exists(exe) and os.access(exe, os.X_OK)


def test_shipped_library_carries_no_debug_switches_or_superseded_kernels():
    """wrong-output switches (CRL_*_DEBUG, skeleton sweeps, phase skips) and the superseded kernels (round 2's
    analytic CarRacing raster, the first Pong writers, the packed-FMA opponent network) exist only in the profiling variant
    (-DCRL_ABLATION -> libcrl_hip_abl.so); the shipped library contains neither their names nor their code objects."""
    import subprocess

    from competitive_rl_amd import _native as N
    from competitive_rl_amd.build import PKG, build

    build()
    lib = os.path.join(PKG, "libcrl_hip.so")
    out = subprocess.run(["strings", "-n", "6", lib], capture_output=True, text=True, check=True).stdout
    hdr = open(os.path.join(ROOT, "include", "crl.h")).read()
    switches = sorted(x for x in set(re.findall(r"\bCRL_[A-Z0-9_]{3,}\b", out)) if x not in hdr)  # (enum names occur in error texts)
    assert not [s for s in switches if "DEBUG" in s or "_ABL" in s or "STAMPS" in s or "ANALYTIC" in s], switches
    assert len(switches) <= 6, switches  # alternative CORRECT paths only
    kernels = set(re.findall(r"_ZN3crl\d+([a-z_0-9]+kernel)", out))
    dead = {"car_raster_kernel", "car_raster_list_kernel", "pong_raster_raw_kernel", "pong_raster_raw_linear_kernel",
            "pong_raster_gray_kernel", "pong_raster_gray_sweep_kernel", "pong_gray_sweep_skeleton_kernel", "pong_gray_header_kernel",
            "pong_policy_light_kernel"}
    assert kernels and not (kernels & dead), sorted(kernels & dead)
    assert {"car_obs_kernel", "car_touch_kernel", "pong_raster_raw_sweep_kernel", "pong_raster_gray_env_kernel", "pong_policy_mfma_kernel"} <= kernels
    assert os.path.basename(N.LIB_PATH) == "libcrl_hip.so" or os.environ.get("CRL_LIB_VARIANT")


def test_spaces_mirror_the_gym_classes_the_reference_builds():
    """Box / Discrete / Tuple / Dict: constructor arguments, ``sample`` / ``contains`` and indexing as the reference uses them
    (pong/base_pong_env.py:91-101; car_racing_multi_players.py:237-245: two cars' actions are a Dict {0: Box(2,), 1: Box(2,)})."""
    from competitive_rl_amd import spaces

    one = spaces.Box(-1, 1, (2,), dtype=np.float32)
    both = spaces.Dict({i: one for i in range(2)})
    a = both.sample()
    assert list(a.keys()) == [0, 1] and all(v.shape == (2,) and v.dtype == np.float32 for v in a.values())
    assert both.contains(a) and not both.contains({0: a[0]}) and not both.contains({0: a[0], 1: np.array([2.0, 0.0], np.float32)})
    assert both[0] is one and len(both) == 2 and list(both) == [0, 1]
    t = spaces.Tuple([spaces.Discrete(3), spaces.Discrete(3)])
    assert len(t) == 2 and t[1].n == 3 and all(0 <= x < 3 for x in t.sample())


def test_tile_images_and_vec_env_wrapper_follow_the_reference_contract():
    """``tile_images`` (utils/base_vec_env.py:10-38) and ``VecEnvWrapper`` (:255-374) on a toy vector env: the grid shape and cell order,
    forwarding, attribute lookup down the chain, the ambiguity error, ``unwrapped``."""
    import pytest

    from competitive_rl_amd import VecEnv, VecEnvWrapper, spaces, tile_images

    imgs = [np.full((2, 3, 3), k + 1, np.uint8) for k in range(5)]          # 5 images -> 3 rows x 2 columns, last cell black
    big = tile_images(imgs)
    assert big.shape == (6, 6, 3) and (big[:2, :3] == 1).all() and (big[:2, 3:] == 2).all() and (big[2:4, :3] == 3).all()
    assert (big[4:, :3] == 5).all() and (big[4:, 3:] == 0).all()
    assert tile_images(np.ones((4, 2, 2, 1))).shape == (4, 4, 1) and tile_images(np.ones((1, 2, 2, 3))).shape == (2, 2, 3)
    assert tile_images(np.ones((3, 96, 96), np.uint8)).shape == (192, 192)   # gray frames (CarRacing)

    class Toy(VecEnv):
        color = "red"

        def __init__(self):
            VecEnv.__init__(self, 3, spaces.Box(0, 255, (1, 4, 4), dtype=np.uint8), spaces.Discrete(3))
            self.sent, self.closed, self.depth = None, False, 0

        def reset(self):
            return np.zeros((3, 1, 4, 4), np.uint8)

        def step_async(self, actions):
            self.sent = actions

        def step_wait(self):
            return self.reset(), np.ones(3, np.float32), np.zeros(3, bool), [{} for _ in range(3)]

        def close(self):
            self.closed = True

        def seed(self, seed=None):
            return [seed + i for i in range(3)]

        def get_images(self):
            return [np.full((4, 4, 3), i, np.uint8) for i in range(3)]

        def get_attr(self, name, indices=None):
            return [getattr(self, name)] * len(self._get_indices(indices))

        def set_attr(self, name, value, indices=None):
            setattr(self, name, value)

        def env_method(self, name, *a, indices=None, **k):
            return [name] * len(self._get_indices(indices))

    class Scale(VecEnvWrapper):
        def __init__(self, venv, k):
            VecEnvWrapper.__init__(self, venv)
            self.k = k

        def reset(self):
            return self.venv.reset()

        def step_wait(self):
            o, r, d, i = self.venv.step_wait()
            return o, r * self.k, d, i

    toy = Toy()
    w = Scale(Scale(toy, 2.0), 3.0)
    assert w.num_envs == 3 and w.observation_space is toy.observation_space and w.action_space is toy.action_space
    _, r, _, _ = w.step([0, 1, 2])
    assert toy.sent == [0, 1, 2] and (r == 6.0).all()
    assert w.seed(5) == [5, 6, 7] and w.get_attr("color", 1) == ["red"] and w.env_method("foo", indices=[0, 2]) == ["foo", "foo"]
    assert w.render("rgb_array").shape == (8, 8, 3) and len(w.get_images()) == 3 and w.unwrapped is toy and toy.unwrapped is toy
    assert w.color == "red" and w.depth == 0          # found two levels down
    w.set_attr("depth", 4)
    assert toy.depth == 4
    with pytest.raises(AttributeError, match="ambiguous"):
        w.getattr_depth_check  # noqa: B018  (resolves normally: defined on the class)
        Scale.__getattr__(w, "k")                     # both wrappers own `k`: the lookup from above is refused
    with pytest.raises(AttributeError):
        w.no_such_thing
    w.close()
    assert toy.closed
    with pytest.raises(TypeError):
        VecEnvWrapper(toy)                            # abstract: reset / step_wait are the subclass's


def test_package_exports_the_reference_packages_top_level_names():
    """``import competitive_rl_amd as competitive_rl`` finds what ``competitive_rl/__init__.py:1-6`` exports (PrintConsole, the trainer's
    console printer, excepted: host bookkeeping outside the path), with the reference's calling conventions for the small policies."""
    import competitive_rl_amd as crl

    for name in ("make_envs", "get_random_policy", "get_rule_based_policy", "get_compute_action_function", "get_builtin_agent_names",
                 "evaluate_two_policies_in_batch", "evaluate_two_policies", "register_pong", "register_car_racing", "register_competitive_envs",
                 "FrameStackTensor", "make_competitive_car_racing", "TournamentEnvWrapper"):
        assert callable(getattr(crl, name)), name
    assert crl.register_competitive_envs() is None
    assert crl.get_rule_based_policy()(None) == crl.CHEAT_CODES == 999 and crl.get_rule_based_policy(3)(None) == [999] * 3
    a = crl.get_random_policy(5)(None)
    assert len(a) == 5 and all(x in (0, 1, 2) for x in a) and crl.get_random_policy()(None) in (0, 1, 2)
    assert set(crl.get_builtin_agent_names()) >= {"RANDOM", "WEAK", "MEDIUM", "RULE_BASED"}
    assert crl.get_compute_action_function("RULE_BASED", 2)(None) == [999, 999] and crl.get_compute_action_function("RULE_BASED")(None) == 999
    try:
        crl.get_compute_action_function("NOBODY")
    except ValueError as e:
        assert "Unknown agent name" in str(e)
    else:
        raise AssertionError("unknown agent accepted")


def test_thunk_factories_and_vec_env_constructors_without_a_gpu():
    """``DummyVecEnv([make_env_a2c_atari(...) for i in range(n)])`` -- the reference's own way to build a batch (make_envs.py:100-117) --
    validates its thunks on the host and then fails loudly without a GPU, like every env constructor of the package."""
    import pytest
    import torch

    import competitive_rl_amd as crl

    t = [crl.make_env_a2c_atari("cPongDouble-v0", 0, i, None, 42, None) for i in range(3)]
    assert [x.rank for x in t] == [0, 1, 2] and t[0].key() == t[2].key() and isinstance(t[0], crl.EnvThunk)
    with pytest.raises(ValueError, match="consecutive"):
        crl.DummyVecEnv([t[0], t[2]])
    with pytest.raises(ValueError, match="one batch"):
        crl.SubprocVecEnv([t[0], crl.make_env_a2c_atari("cPongDouble-v0", 0, 1, None, 84, None)])
    with pytest.raises(TypeError):
        crl.DummyVecEnv([lambda: None])
    with pytest.raises(TypeError):
        crl.DummyVecEnv([])
    with pytest.raises(AssertionError):
        crl.make_car_racing("cPong-v0", 0, 0)
    c = crl.make_car_racing_double(5, 2, frame_stack=4, action_repeat=2)
    assert (c.env_id, c.seed, c.rank, c.frame_stack, c.action_repeat) == ("cCarRacingDouble-v0", 5, 2, 4, 2)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="GPU"):
            crl.DummyVecEnv(t)
        with pytest.raises(RuntimeError, match="GPU"):
            t[1]()


def test_single_env_handle_follows_gyms_reset_step_contract_on_a_fake_batch_of_one():
    """``envs.envs[0].reset() / .step(action)`` of a one-env batch (vis.py's flow): the batch axis comes off, ``done`` is a scalar, the step that
    ends an episode returns the TERMINAL observation, and the ``reset()`` that follows it hands out the restarted episode's first
    observation without resetting the (already restarted) env again; a batch of several envs refuses per-env stepping."""
    import pytest

    from competitive_rl_amd import spaces
    from competitive_rl_amd.vec_env import _EnvHandle

    class Fake:
        def __init__(self, n=1):
            self.num_envs, self._serial, self.resets, self.t = n, 0, 0, 0
            self.observation_space = spaces.Tuple([spaces.Box(0, 255, (1, 2, 2), dtype=np.uint8)] * 2)
            self.action_space = spaces.Tuple([spaces.Discrete(3), spaces.Discrete(3)])
            self.sent = []

        def _obs(self, v):
            return tuple(np.full((self.num_envs, 1, 2, 2), v + k, np.int32) for k in range(2))

        def reset(self):
            self.resets += 1
            self.t = 0
            return self._obs(100 * self.resets)

        def step(self, actions):
            self.sent.append(np.asarray(actions).copy())
            self._serial += 1
            self.t += 1
            done = self.t == 3
            info = [{}]
            if done:  # the vector env restarts by itself and keeps the last observation in the infos
                info[0]["terminal_observation"] = tuple(o[0] for o in self._obs(50 + self.t))
                self.resets += 1
                self.t = 0
                obs = self._obs(100 * self.resets)
            else:
                obs = self._obs(10 + self.t)
            return obs, np.array([[1.0, -1.0]], np.float32), np.array([[done, done]]), info

    v = Fake()
    e = _EnvHandle(v, 0)
    o = e.reset()
    assert isinstance(o, tuple) and o[0].shape == (1, 2, 2) and o[0][0, 0, 0] == 100 and v.resets == 1
    for t in range(1, 4):
        o, r, d, i = e.step([np.array([2]), 1])
        assert v.sent[-1].shape == (1, 2) and v.sent[-1].tolist() == [[2, 1]] and r.tolist() == [1.0, -1.0] and d is (t == 3)
    assert o[0][0, 0, 0] == 53 and "terminal_observation" in i       # the episode's last observation, not the next episode's first
    o = e.reset()
    assert o[0][0, 0, 0] == 200 and v.resets == 2                     # the restart the vector env already did: no second reset
    o = e.reset()
    assert o[0][0, 0, 0] == 300 and v.resets == 3                     # (a reset that does not follow a finished episode is a reset)
    e.step([0, 0])
    assert e.reset()[0][0, 0, 0] == 400
    with pytest.raises(NotImplementedError):
        _EnvHandle(Fake(2), 0).reset()
    with pytest.raises(NotImplementedError):
        _EnvHandle(Fake(2), 1).step([0, 0])
