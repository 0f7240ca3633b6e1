"""``FrameStackTensor``: the rolling (N, C*k, H, W) float32 observation stack a trainer keeps next to the
vector env (interface of reference competitive_rl/utils/utils.py:145-173).

Semantics pinned by tests/golden/step_envs.npz (recorded from the reference's class): an ``update(obs, mask)``
scales every env's history by its mask entry (0 where an episode just ended: the history is erased, not
replicated), drops the oldest C planes, and appends the new observation as the newest C planes.

Two ways the update runs on a GPU, same values:

* **bound to the env (round 6)** -- ``bind(envs)``, done by ``step_envs`` on its first call.  A wrapped Pong context keeps the
  descriptors of every env's last four planes with exactly this history rule, so ``envs.step`` DRAWS the stack's next state
  (``crl_step_stack``: the launch that draws the observation also writes agent 0's k planes into the stack's other buffer) and
  ``update`` is a pointer swap: the float32 stack is written once per step (7.4 GB at 65 536 x (4, 84, 84)) and never read,
  instead of rolled and appended by a pass of its own (13.4 GB).  The bound stack and the generic one are the same bytes
  (tests/test_hip_stack_fused.py: through episode ends, stack resets and re-binds).  What a bound ``update`` accepts without
  falling back: the env's newest learner observation with ``mask`` = 1 - done of that step, which is what ``step_envs`` passes
  (``_from_env`` marks its calls), or the observation of ``envs.reset()`` without a mask.  Anything else -- a foreign
  observation, a mask of the caller's own, two updates for one env step, an env step without an update -- is served by the
  generic kernel below and the binding is re-checked against the env's history (one comparison on the device) before it is
  used again.  ``dtype=torch.uint8`` (opt-in, not the reference's contract) keeps the stack as bytes: a quarter of the traffic.
* **generic** -- one kernel of libcrl_hip.so (csrc/frame_stack.hip): the observation is taken where it lies in HBM -- uint8 or
  float32, also a strided view such as agent 0's half of the (N, 2, K, R, R) buffer -- and widened on the fly.
  ``out_of_place=True`` (default): the reference's own data flow -- its ``self.current_obs = self.current_obs.roll(...)`` binds a
  NEW tensor on every update -- as a ping-pong of two buffers (a linear copy: 2.16 ms for 65 536 x (4, 84, 84) float32 = 0.78 of
  HBM, against 2.96 ms for the in-place column walk).  ``out_of_place=False``: one buffer, updated in place.

Lifetime of the returned tensor (both ways, two buffers): the tensor handed out by ``get()`` / ``update()`` stays intact through
the NEXT update and is recycled by the one after (the reference's old tensor lives as long as someone holds it; a rollout buffer
copies what it stores, as the reference's does).  The second buffer doubles the footprint (7.4 GB more at 65 536 x (4, 84, 84)
float32): it is allocated by the first update and released by ``reset()``; ``out_of_place=False`` never allocates it (and is
never bound).

``device="cpu"`` keeps the class usable with host-resident vector envs (numpy observations); that path is
plain tensor arithmetic and is not part of the GPU hot path.  A CUDA device without the library raises.
"""
import ctypes as C
import weakref

import numpy as np
import torch

from . import _native as N


_MASK_OF_LAST_STEP = object()  # update_from_env: "1 - done of the env's last step", built only if the generic kernel needs it


class FrameStackTensor:
    def __init__(self, num_envs, obs_shape, frame_stack, device, out_of_place=True, dtype=torch.float32):
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:  # ("cuda" names the current device: the env it may be bound to says cuda:<i>)
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.out_of_place = bool(out_of_place)
        if dtype not in (torch.float32, torch.uint8):
            raise ValueError("FrameStackTensor holds float32 (the reference's contract) or uint8 (opt-in)")
        self.dtype = dtype
        self._spare = None
        self.num_envs, self.frame_stack = int(num_envs), int(frame_stack)
        self.num_channels = int(obs_shape[0])
        self.plane_shape = tuple(int(d) for d in obs_shape[1:])
        self.obs_shape = (self.num_channels * self.frame_stack, *self.plane_shape)
        self.current_obs = torch.zeros((self.num_envs, *self.obs_shape), dtype=dtype, device=self.device)
        self._hw = int(np.prod(self.plane_shape)) if self.plane_shape else 1
        self._lib = N.load() if self.device.type == "cuda" else None
        # ---- binding to a HIP Pong env (see the module docstring)
        self._env = None          # weakref to the bound HipPongVecEnv
        self._bind_tried = False  # step_envs tries once
        self._age = 0             # updates since reset(), capped at frame_stack: the planes older than that are zeros
        self._zero = True         # the tensor is known to be all zeros (fresh, or reset())
        self._synced = False      # the env's plane history == this tensor's content (the fused draw may replace the generic update)
        self._env_serial = -1     # the env's step counter at the last update that consumed an env observation
        self._pre = None          # (env serial, buffer) of a stack state the env's last step / reset drew ahead
        self._verify_left = 8     # re-checks of an unsynced binding before it is given up
        self.fused_updates = 0    # (statistics: how many updates were pointer swaps)

    # ------------------------------------------------------------------ reference interface
    def reset(self):
        self.current_obs.zero_()
        self._spare = None  # the second buffer is re-allocated by the next update (reset() used to leave it alive and stale)
        self._age, self._zero, self._pre = 0, True, None
        if self._env is not None and self._env() is not None:
            self._synced, self._verify_left = True, 8  # an all-zero stack follows any history: only planes younger than now are drawn

    def get(self):
        return self.current_obs

    def update(self, obs, mask=None, _from_env=None):
        """``_from_env`` (private, set by ``step_envs``): the env whose ``step`` produced ``obs``, with ``mask`` = 1 - done of that step."""
        if self._try_commit(obs, mask, _from_env):
            return self.current_obs
        o, m = self._as_obs(obs), self._as_mask(mask)
        if self._lib is not None:
            self._update_hip(o, m)
        else:
            self._update_host(o, m)
        self._zero = False
        self._age = min(self._age + 1, self.frame_stack)
        self._note_generic_update(obs, mask, _from_env)
        return self.current_obs

    def update_from_env(self, envs):
        """For a training loop of one's own (``step_envs`` does this itself): push the newest learner observation of ``envs`` with the
        mask of its last step (1 - done; none after a ``reset()``).  On a bound stack this is the pointer swap."""
        env = getattr(envs, "_stack_env", lambda: None)()
        if env is None:
            raise TypeError("update_from_env needs the HIP Pong vector env (or a wrapper of it)")
        obs = env._latest_learner_obs()
        stepped = env._last_kind == "step"
        if self._try_commit(obs, _MASK_OF_LAST_STEP if stepped else None, envs if stepped else None):
            return self.current_obs
        mask = (env._done == 0).to(torch.float32) if stepped else None
        return self.update(obs, mask, _from_env=envs if stepped else None)

    # ------------------------------------------------------------------ binding
    def bind(self, envs):
        """Let ``envs.step`` draw this stack (module docstring).  Returns True when the env can (a HipPongVecEnv -- or a wrapper
        that forwards to one -- in wrapped mode, FrameStackTensor history, observation (1, R, R) per agent, at most 4 planes,
        same device, two buffers); False leaves the stack on the generic kernel.  Safe to call at any time: a stack whose
        content cannot be explained by the env's history stays generic until it can."""
        self._bind_tried = True
        env = getattr(envs, "_stack_env", lambda: None)()
        if env is None or not self.out_of_place:
            return False
        if not env._can_draw_stack(self):  # (the HIP env asks for its own device: a host-resident stack never binds to it)
            return False
        other = env._bound_stack() if env._bound_stack is not None else None
        if other is not None and other is not self:
            other.unbind()  # an env draws one stack
        self._env = weakref.ref(env)
        env._bound_stack = weakref.ref(self)
        self._pre, self._verify_left = None, 8
        self._synced = self._zero  # an all-zero stack needs no check; anything else is compared with the env's history on first use
        if not self._synced:
            self._verify(env)
        return True

    def unbind(self):
        env = self._env() if self._env is not None else None
        if env is not None and env._bound_stack is not None and env._bound_stack() is self:
            env._bound_stack = None
        self._env, self._pre, self._synced = None, None, False

    def _other_buffer(self):
        if self._spare is None:
            self._spare = torch.empty_like(self.current_obs)
        return self._spare

    def _stack_desc(self, buf, valid, alias):
        return N.CrlStackDesc(stack_dev=buf.data_ptr(), planes=self.frame_stack, dtype=N.CRL_OBS_F32 if self.dtype == torch.float32 else N.CRL_OBS_U8,
                              agent=0, valid_planes=int(valid), alias_newest=1 if alias else 0, reserved=0)

    def _predraw(self, env, kind):
        """Called by the env inside step / reset (before it counts the call): the buffer the launch should draw this stack's NEXT state
        into and its descriptor, or None (not synced: the env then draws its observation only)."""
        self._pre = None
        if not self._synced:
            return None
        # exactly one env step since the last update that consumed one -- or a stack without history, which follows any; an env reset
        # under a stack that was not reset leaves the old episode's planes in the trainer's tensor (the generic update keeps them)
        if not (self._zero or (env._serial == self._env_serial and kind == "step")):
            self._synced = False
            return None
        buf = self._other_buffer()
        return buf, self._stack_desc(buf, min(self._age + 1, self.frame_stack), env._stack_alias(self))

    def _predrawn(self, env, buf, kind):
        self._pre = (env._serial, buf, kind)

    def _try_commit(self, obs, mask, from_env):
        pre, self._pre = self._pre, None
        env = self._env() if self._env is not None else None
        if pre is None or env is None or env.closed or not self._synced:
            return False
        serial, buf, kind = pre
        if serial != env._serial or not env._is_latest_learner_obs(obs):
            return False
        if kind == "step":
            if from_env is None or getattr(from_env, "_stack_env", lambda: None)() is not env or mask is None:
                return False  # no mask, or a mask of the caller's own: the generic kernel applies it
        elif mask is not None or not self._zero:  # "reset": the first observation of a fresh / reset stack
            return False
        self._spare, self.current_obs = self.current_obs, buf
        self._age = min(self._age + 1, self.frame_stack)
        self._zero, self._env_serial = False, env._serial
        self.fused_updates += 1
        return True

    def _note_generic_update(self, obs, mask, from_env):
        """A generic update ran on a bound stack: the binding is valid again once the content is what the env's history draws."""
        env = self._env() if self._env is not None else None
        if env is None or env.closed:
            return
        if mask is not None and from_env is None:
            # the caller applies masks of its own: nothing a step could draw ahead (a draw that is thrown away costs a full write of the stack)
            self.unbind()
            return
        self._synced, self._env_serial = False, env._serial
        if not env._is_latest_learner_obs(obs):
            return  # a foreign observation is in the stack now: it has to roll out first (the next generic updates re-check)
        if self._verify_left <= 0:
            self.unbind()
            return
        self._verify(env)

    def _verify(self, env):
        """Is this tensor what the env's plane history draws for a stack of this age?  One draw into the other buffer + one comparison on
        the device (a host synchronisation: after a bind of a used stack or a generic update, at most eight times in a row)."""
        self._verify_left -= 1
        buf = self._other_buffer()
        env._draw_stack_into(self._stack_desc(buf, min(self._age, self.frame_stack), False))
        if bool(torch.equal(buf, self.current_obs)):
            self._synced, self._env_serial, self._verify_left = True, env._serial, 8
        return self._synced

    # ------------------------------------------------------------------ generic update
    def _as_mask(self, mask):
        """(N,) float32 on the stack's device, or None.  Accepts the (N, 1), (N, 1, 1, 1) ... shapes trainers pass."""
        if mask is None:
            return None
        m = torch.from_numpy(np.ascontiguousarray(mask)) if isinstance(mask, np.ndarray) else mask
        if m.numel() != self.num_envs:
            raise ValueError(f"mask must hold one value per env ({self.num_envs}), got shape {tuple(m.shape)}")
        return m.to(device=self.device, dtype=torch.float32).reshape(self.num_envs).contiguous()

    def _as_obs(self, obs):
        o = torch.from_numpy(obs) if isinstance(obs, np.ndarray) else obs
        want = (self.num_envs, self.num_channels, *self.plane_shape)
        if tuple(o.shape) != want:
            raise ValueError(f"observation must have shape {want}, got {tuple(o.shape)}")
        if o.dtype not in (torch.uint8, torch.float32):
            o = o.to(torch.float32)
        return o.to(self.device)

    def _update_hip(self, o, m):
        # the kernel walks planes of H*W contiguous elements; envs may be strided (a view of a wider buffer)
        inner = o[0]
        if not inner.is_contiguous() or (self.num_envs > 1 and o.stride(0) < inner.numel()):
            o = o.contiguous()
        stride = o.stride(0) if self.num_envs > 1 else o[0].numel()
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        tail = (N.CRL_OBS_F32 if o.dtype == torch.float32 else N.CRL_OBS_U8, int(stride), None if m is None else C.c_void_p(m.data_ptr()),
                self.num_envs, self.num_channels, self.frame_stack, self._hw, st)
        with torch.cuda.device(self.device):
            if self.dtype == torch.uint8:  # the opt-in byte stack: one entry point, in place (dst == src) or into the other buffer
                dst = self._other_buffer() if self.out_of_place else self.current_obs
                N.check(self._lib.crl_frame_stack_update_u8(C.c_void_p(dst.data_ptr()), C.c_void_p(self.current_obs.data_ptr()), C.c_void_p(o.data_ptr()), *tail))
                if self.out_of_place:
                    self._spare, self.current_obs = self.current_obs, dst
            elif self.out_of_place:
                # the reference's own data flow (`self.current_obs = self.current_obs.roll(...)`: a new tensor per update), as a ping-pong of two
                # buffers: the tensor handed out by the LAST update stays intact through this one and is recycled by the next
                dst = self._other_buffer()
                N.check(self._lib.crl_frame_stack_update_to(C.c_void_p(dst.data_ptr()), C.c_void_p(self.current_obs.data_ptr()),
                                                            C.c_void_p(o.data_ptr()), *tail))
                self._spare, self.current_obs = self.current_obs, dst
            else:
                N.check(self._lib.crl_frame_stack_update(C.c_void_p(self.current_obs.data_ptr()), C.c_void_p(o.data_ptr()), *tail))

    def _update_host(self, o, m):
        """Plain tensor arithmetic: host-resident stacks only (device="cpu", numpy observations)."""
        c, buf = self.num_channels, self.current_obs
        kept = buf[:, c:]
        if m is not None:
            mm = m.reshape(self.num_envs, *([1] * (buf.dim() - 1)))
            kept = kept * (mm if self.dtype == torch.float32 else (mm != 0).to(torch.uint8))
        else:
            kept = kept.clone()
        buf[:, :buf.shape[1] - c] = kept
        buf[:, buf.shape[1] - c:] = o.to(self.dtype)
