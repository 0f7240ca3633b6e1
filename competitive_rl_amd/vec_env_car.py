"""HIP vector env for cCarRacingDouble-v0 behind the reference's VecEnv protocol.

Mirrors ``DummyVecEnv([make_car_racing_double(seed, i, frame_stack, action_repeat)] * N)``
(reference competitive_rl/car_racing/register.py:43-53, utils/atari_wrappers.py:308-334,
utils/dummy_vec_env.py:10-133): observation (N, 2, 96, 96) uint8 (both agents' 96x96 gray views on
the channel axis, WrapPyTorch layout), actions (N, 2, 2) floats in [-1, 1], reward = agent 0's
(N, 1), done = any agent done or the 1000-step TimeLimit (N, 1); per-agent rewards and step
counts in ``infos``.  Auto-reset lays a fresh procedural track on the GPU.
"""
import ctypes as C

import numpy as np
import torch

from . import _native as N
from . import spaces
from .vec_env import VecEnv, _EnvList


class CarLazyInfos:
    """``infos[i]`` -> ``{0: {"num_steps": k, "reward": r0}, 1: {...}}`` (crmp:618-620,
    atari_wrappers.py:327-328) without building N dicts per step.  ``num_steps`` is ``CarRacing.step_count``
    read from the device state (it advances by ``action_repeat`` per step); ``TimeLimit.truncated`` appears on
    the step gym's TimeLimit ends (``not done`` of what the env returned: a dict there, hence False for two
    cars).  All terminal observations of a step are fetched by one library call; they must be read before the
    env is stepped again (the older planes of a stacked terminal observation live in the previous buffer)."""

    def __init__(self, env, rew, steps, done, done_car, elapsed):
        self._env, self._n = env, env.num_envs
        self._dev = (rew, steps, done, done_car, elapsed)
        self._host = None
        self._serial = env._serial
        self._term = None

    def __len__(self):
        return self._n

    def _terminal(self, i):
        if self._term is None:
            if self._env._serial != self._serial:
                raise RuntimeError("terminal_observation of a past step: read infos[i] before stepping the env again")
            idx = torch.nonzero(self._dev[2]).reshape(-1)
            self._term = dict(zip(idx.cpu().tolist(), self._env.terminal_observation(idx)))
        return self._term[i]

    def __getitem__(self, i):
        if self._host is None:
            self._host = tuple(t.cpu().numpy() for t in self._dev)
        r, st, dn, dc, el = self._host
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        if r.shape[1] == 1:  # cCarRacing-v0: info = {"num_steps": k} (crmp:616)
            d = {"num_steps": int(st[i])}
        else:
            d = {k: {"num_steps": int(st[i]), "reward": float(r[i, k])} for k in range(2)}
        if el[i] >= 1000:  # gym TimeLimit (max_episode_steps=1000, car_racing/register.py:15-26)
            d["TimeLimit.truncated"] = (not bool(dc[i, 0])) if r.shape[1] == 1 else False
        if dn[i]:
            d["terminal_observation"] = self._terminal(int(i))
        return d

    def __iter__(self):
        return (self[i] for i in range(self._n))

    def copy(self):
        return self


class HipCarVecEnv(VecEnv):
    def __init__(self, num_envs, seed=0, device=None, env_id_base=0, output="torch", dones="dummy", action_repeat=None,
                 frame_stack=None, players=2, car_contacts=True, done_policy="any", solver="box2d"):
        if not torch.cuda.is_available():
            raise RuntimeError("HipCarVecEnv needs a ROCm GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.action_repeat = 1 if action_repeat is None else int(action_repeat)
        assert 1 <= self.action_repeat <= 16
        assert output in ("torch", "numpy") and dones in ("dummy", "subproc") and done_policy in ("any", "car0")
        self._L = N.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.output, self.dones_kind, self.closed = output, dones, False
        self.K = 1 if frame_stack is None else int(frame_stack)
        assert players in (1, 2)
        # solver="fma": world.Step's island iterations in fused multiply-adds (include/crl.h CRL_FLAG_CAR_FMA): 0.46 x the instructions of
        # the step's longest chain; checked against its own build of the CPU checker (tolerance 0), velocities within 2.7e-5 of Box2D's roundings after one step
        assert solver in ("box2d", "fma")
        self.solver = solver
        self.P = int(players)  # 2 = cCarRacingDouble-v0, 1 = cCarRacing-v0
        opts = N.CrlOpts(env_kind=N.CRL_ENV_CAR_DOUBLE if players == 2 else N.CRL_ENV_CAR_SINGLE, obs_mode=0,
                         resized_dim=0, frame_stack=self.K, num_envs=int(num_envs),
                         env_id_base=int(env_id_base), seed=int(seed) & (2 ** 64 - 1), device=self.device.index or 0,
                         flags=(0 if car_contacts else N.CRL_FLAG_CAR_NO_CONTACTS) | (N.CRL_FLAG_CAR_FMA if solver == "fma" else 0), action_repeat=self.action_repeat,
                         done_policy=N.CRL_CAR_DONE_CAR0 if done_policy == "car0" else N.CRL_CAR_DONE_ANY)
        h = C.c_void_p()
        self._text = N.load_car_text()  # reward read-out bitmaps of the indicator strip
        with torch.cuda.device(self.device):
            N.check(self._L.crl_create(C.byref(opts), self._text.ctypes.data_as(C.c_void_p), C.byref(h)))
        self._h = h
        n = int(num_envs)
        obs_space = spaces.Box(0, 255, (self.P * self.K, 96, 96), dtype=np.uint8)
        # Box(-1, 1, (2,)) per car (car_racing_multi_players.py:237); the raw two-car env's Dict {0: Box, 1: Box} (:245) becomes
        # Box(-1, 1, (num_players, 2)) under FlattenMultiAgentObservation (utils/atari_wrappers.py:316), which make_car_racing_double applies
        act_space = spaces.Box(-1, 1, (2, 2) if self.P == 2 else (2,), dtype=np.float32)
        VecEnv.__init__(self, n, obs_space, act_space)
        dev = self.device
        self._obs = [torch.empty((n, self.P * self.K, 96, 96), dtype=torch.uint8, device=dev) for _ in range(2)]
        self._flip = 0
        self._rew = torch.zeros((n, self.P), dtype=torch.float32, device=dev)
        self._done = torch.zeros((n,), dtype=torch.uint8, device=dev)
        self._actions = torch.zeros((n, self.P, 2), dtype=torch.float32, device=dev)
        self._serial = 0
        self._prev_buf = self._cur_out = self._obs[1]
        self.envs = _EnvList(self)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check_open(self):
        if self.closed:
            raise RuntimeError("VecEnv is closed")

    def seed(self, seed=None):
        self._check_open()
        N.check(self._L.crl_seed(self._h, int(seed or 0) & (2 ** 64 - 1)))
        return [None] * self.num_envs

    def _out(self, t):
        return t.cpu().numpy() if self.output == "numpy" else t

    def reset(self):
        self._check_open()
        buf = self._obs[self._flip]
        N.check(self._L.crl_reset(self._h, C.c_void_p(buf.data_ptr()), self._stream()))
        self._flip ^= 1  # (only after the call went through: a failed one leaves get_images / lazy infos on the drawn buffer)
        self._serial += 1
        return self._out(buf)

    def step_async(self, actions):
        self._check_open()
        if isinstance(actions, torch.Tensor):
            a = actions.to(device=self.device, dtype=torch.float32)
        else:
            if self.P == 2 and len(actions) and isinstance(actions[0], dict):
                # the reference indexes an env's action by car key (utils/atari_wrappers.py:325, car_racing_multi_players.py:550-555): per-env dicts {0: a0, 1: a1} work too
                actions = [[a[k] for k in range(self.P)] for a in actions]
            a = torch.as_tensor(np.asarray(actions, dtype=np.float32)).to(self.device)
        if self.P == 1 and tuple(a.shape) == (self.num_envs, 2):
            a = a[:, None, :]
        if tuple(a.shape) != (self.num_envs, self.P, 2):
            raise AssertionError(f"actions must have shape ({self.num_envs}, {self.P}, 2), got {tuple(a.shape)}")
        self._actions = a.contiguous()

    def step_device(self, actions_f32, render=True, obs_out=None):
        """Hot-loop entry: float32 (N, 2, 2) device tensor in, device tensors out, no sync.  ``obs_out``: draw the observation straight
        into the caller's tensor (e.g. its slot of a collective's send buffer, sharding.StepGather.obs_slot) instead of the env's
        double buffer; the caller keeps it intact until the next step has been issued (a stacked env's terminal observations read it)."""
        if not (actions_f32.is_contiguous() and actions_f32.dtype == torch.float32 and actions_f32.device == self.device):
            raise AssertionError("step_device needs a contiguous float32 tensor on the env's device")
        if obs_out is not None:
            if not (obs_out.is_contiguous() and obs_out.dtype == torch.uint8 and obs_out.device == self.device
                    and obs_out.numel() == self._obs[0].numel()):
                raise AssertionError("obs_out must be a contiguous uint8 tensor of the observation buffer's size on the env's device")
            N.check(self._L.crl_step(self._h, C.c_void_p(actions_f32.data_ptr()), C.c_void_p(obs_out.data_ptr()),
                                     C.c_void_p(self._rew.data_ptr()), C.c_void_p(self._done.data_ptr()), self._stream()))
            self._serial += 1
            self._prev_buf, self._cur_out = self._cur_out, obs_out.view(self._obs[0].shape)
            return self._cur_out, self._rew, self._done
        buf = self._obs[self._flip]
        N.check(self._L.crl_step(self._h, C.c_void_p(actions_f32.data_ptr()), C.c_void_p(buf.data_ptr()) if render else None,
                                 C.c_void_p(self._rew.data_ptr()), C.c_void_p(self._done.data_ptr()), self._stream()))
        self._serial += 1
        if not render:  # nothing was drawn: the buffers stay as they are, and there is no observation to hand out
            return None, self._rew, self._done
        self._prev_buf = self._cur_out  # observation before this step (the stack's older planes)
        self._cur_out = buf
        self._flip ^= 1
        return buf, self._rew, self._done

    def _info_snapshot(self, elapsed=False):
        """Copies of the library's per-step info arrays: per-car done flags (N, P) u8, CarRacing.step_count (N,) i32 and, on
        request, gym TimeLimit's step count after the step (N,) i32 -- the device keeps that counter (it is part of the state)."""
        dc = torch.empty((self.num_envs, self.P), dtype=torch.uint8, device=self.device)
        ns = torch.empty((self.num_envs,), dtype=torch.int32, device=self.device)
        el = torch.empty((self.num_envs,), dtype=torch.int32, device=self.device) if elapsed else None
        N.check(self._L.crl_car_copy_info(self._h, C.c_void_p(dc.data_ptr()), C.c_void_p(ns.data_ptr()),
                                          C.c_void_p(el.data_ptr()) if elapsed else None, self._stream()))
        return (dc, ns, el) if elapsed else (dc, ns)

    def step_wait(self):
        self._check_open()
        buf, rew, done = self.step_device(self._actions)
        dc, ns, el = self._info_snapshot(elapsed=True)
        infos = CarLazyInfos(self, rew.clone(), ns, done.clone(), dc, el)
        r0 = rew[:, :1].clone()
        d = done.bool()
        d = d[:, None].clone() if self.dones_kind == "dummy" else d.clone()
        if self.dones_kind == "subproc":
            r0 = r0[:, 0]
        return self._out(buf), self._out(r0), self._out(d), infos

    def close(self):
        if self.closed:
            return
        self.closed = True
        torch.cuda.synchronize(self.device)
        self._L.crl_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # VecEnv.get_attr / set_attr / env_method / render (utils/base_vec_env.py:124-160, 195-217), as on HipPongVecEnv: per-env
    # handles stand in for DummyVecEnv.envs[i]
    def get_attr(self, attr_name, indices=None):
        return [getattr(self.envs[i], attr_name) for i in self._get_indices(indices)]

    def set_attr(self, attr_name, value, indices=None):
        for i in self._get_indices(indices):
            setattr(self.envs[i], attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        return [getattr(self.envs[i], method_name)(*args, **kwargs) for i in self._get_indices(indices)]

    def get_images(self, *a, **k):
        """Agent 0's newest 96 x 96 frame of every env (what the reference's get_observation(0) returned last)."""
        return list(self._obs[self._flip ^ 1][:, self.K - 1].cpu().numpy())

    def terminal_observation(self, env_indices):
        """Observation (P*K, 96, 96) each listed env's episode ended on, at its most recent done step.  ``env_indices``
        may be a device tensor (``torch.nonzero(done)``): one gather kernel, no host round trip."""
        if isinstance(env_indices, torch.Tensor):
            idx = env_indices.to(device=self.device, dtype=torch.int64).reshape(-1).contiguous()
        else:
            host_idx = np.ascontiguousarray(env_indices, np.int64).reshape(-1)
            if ((host_idx < 0) | (host_idx >= self.num_envs)).any():
                raise IndexError(f"env index out of range: {host_idx}")
            idx = torch.as_tensor(host_idx).to(self.device)
        m = idx.numel()
        out = torch.empty((m, self.P, 96, 96), dtype=torch.uint8, device=self.device)
        if m:
            N.check(self._L.crl_terminal_observation_dev(self._h, C.c_void_p(idx.data_ptr()), m, C.c_void_p(out.data_ptr()),
                                                         self._stream()))
        if self.K > 1:  # MultipleFrameStack / FrameStack: K-1 newest planes of the previous obs + the last frame
            prev = self._prev_buf[idx].view(m, self.P, self.K, 96, 96)
            out = torch.cat([prev[:, :, 1:], out[:, :, None]], dim=2).reshape(m, self.P * self.K, 96, 96)
        if self.output == "numpy":
            out = out.cpu().numpy()
        return [out[k] for k in range(m)]

    # ---- parity / checkpoint helpers
    def get_state(self):
        st = np.zeros(self.num_envs, N.CAR_ENV_STATE_DT)
        N.check(self._L.crl_car_get_state(self._h, st.ctypes.data_as(C.c_void_p), 0, self.num_envs, self._stream()))
        return st

    def set_state(self, st):
        st = np.ascontiguousarray(st, N.CAR_ENV_STATE_DT)
        N.check(self._L.crl_car_set_state(self._h, st.ctypes.data_as(C.c_void_p), 0, self.num_envs, self._stream()))

    def state_dict(self):
        """Checkpoint of the whole batch in torch's idiom: the structured per-env state (bodies, joints, wheel model, tile books,
        manifolds, TimeLimit and episode counters) as a plain dict.  The current TRACKS are not part of it: they are a function of
        (seed, global env id, episode) and are rebuilt by the next reset; restore into a context that is on the same episodes (or
        push tracks with set_track) when the cars' positions matter."""
        return {"kind": "cCarRacing", "num_envs": self.num_envs, "players": self.P, "solver": self.solver, "env_state": self.get_state()}

    def load_state_dict(self, sd):
        st = sd["env_state"] if isinstance(sd, dict) else sd  # (a bare get_state() array is accepted too)
        if isinstance(sd, dict) and (sd.get("kind") != "cCarRacing" or int(sd.get("num_envs", -1)) != self.num_envs or int(sd.get("players", self.P)) != self.P):
            raise ValueError(f"state_dict of a {sd.get('kind')} batch of {sd.get('num_envs')} envs does not fit this env")
        if isinstance(sd, dict) and sd.get("solver", self.solver) != self.solver:
            #  the two arithmetics of the island solver leave one state at different successors (velocities up to 2.7e-5 apart per
            # step): a run resumed in the other one silently leaves the recorded trajectory
            raise ValueError(f"state_dict was written by a solver={sd['solver']!r} env; this env runs solver={self.solver!r} "
                             "(HipCarVecEnv(solver=...)): the resumed run would not reproduce the original")
        self.set_state(st)

    def get_track(self, env):
        n = C.c_int32()
        tiles = np.zeros((N.CAR_MAX_TILES, 5, 2), np.float32)
        bpoly = np.zeros((N.CAR_MAX_TILES, 4, 2), np.float32)
        border = np.zeros(N.CAR_MAX_TILES, np.uint8)
        pose = np.zeros(3, np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        N.check(self._L.crl_car_get_track(self._h, int(env), C.byref(n), p(tiles), p(bpoly), p(border), p(pose), self._stream()))
        k = n.value
        return dict(n=k, tile_poly=tiles[:k], border_poly=bpoly[:k], border=border[:k], start_pose=pose)

    def set_track(self, env, tile_poly, border_poly, border, start_pose):
        """Replace one env's track: float64 polygons in the reference's vertex order (road_poly, _create_track
        car_racing_multi_players.py:400-441): tiles [n, 5, 2], border quads [n, 4, 2], border u8 [n] (0 none, 1 white, 2 red)."""
        tile_poly = np.ascontiguousarray(tile_poly, np.float64)
        border_poly = np.ascontiguousarray(border_poly, np.float64)
        border = np.ascontiguousarray(border, np.uint8)
        assert tile_poly.shape[1:] == (5, 2) and border_poly.shape == (len(tile_poly), 4, 2) and border.shape == (len(tile_poly),)
        start_pose = np.ascontiguousarray(start_pose, np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        N.check(self._L.crl_car_set_track(self._h, int(env), len(tile_poly), p(tile_poly), p(border_poly), p(border), p(start_pose),
                                          self._stream()))

    def get_map(self, env):
        """The env's pre-rastered observation map (render_road_for_observation_map): palette indices (1216, 1216) u8, and
        the number of polygon vertices that did not fit the window (0 for every track)."""
        m = np.zeros((N.CAR_MAP_W, N.CAR_MAP_W), np.uint8)
        ov = C.c_int32()
        N.check(self._L.crl_car_get_map(self._h, int(env), m.ctypes.data_as(C.c_void_p), C.byref(ov), self._stream()))
        return m, ov.value

    def cap_hits(self):
        """(wheel-tile slot overflows, car-car manifold overflows, 0, 0) since the env was created: all zero unless a fixed capacity
        of the device state was hit (which would be a deviation from the reference's unbounded sets / lists)."""
        out = np.zeros(4, np.int32)
        N.check(self._L.crl_car_cap_hits(self._h, out.ctypes.data_as(C.c_void_p), self._stream()))
        return tuple(int(x) for x in out)

    def set_replay(self, u, swap):
        """u: [N, attempts, 24] uniforms of _create_track attempts; swap: [N, attempts] birth-place bits."""
        if u is None:
            N.check(self._L.crl_car_set_replay(self._h, None, None, 0))
            return
        u = np.ascontiguousarray(u, np.float64).reshape(self.num_envs, -1, 24)
        swap = np.ascontiguousarray(swap, np.uint8).reshape(self.num_envs, -1)
        N.check(self._L.crl_car_set_replay(self._h, u.ctypes.data_as(C.c_void_p), swap.ctypes.data_as(C.c_void_p), u.shape[1]))

    def render_current(self):
        """Re-draw the current state without stepping (after set_state / set_track)."""
        buf = torch.empty((self.num_envs, 2, 96, 96), dtype=torch.uint8, device=self.device)
        N.check(self._L.crl_render(self._h, C.c_void_p(buf.data_ptr()), self._stream()))
        return buf

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
