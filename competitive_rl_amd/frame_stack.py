"""``FrameStackTensor``: the rolling (N, C*k, H, W) float32 observation stack a trainer keeps next to the
vector env (interface of reference competitive_rl/utils/utils.py:145-173).

Semantics pinned by tests/golden/step_envs.npz (recorded from the reference's class): an ``update(obs, mask)``
scales every env's history by its mask entry (0 where an episode just ended: the history is erased, not
replicated), drops the oldest C planes, and appends the new observation as the newest C planes.

On a GPU the whole update is ONE kernel of libcrl_hip.so (csrc/frame_stack.hip): the observation is taken where the env
kernels left it in HBM -- uint8 or float32, also a strided view such as agent 0's half of the (N, 2, K, R, R) buffer -- and
widened on the fly; nothing crosses PCIe and the stack is read and written once.
``out_of_place=True`` (default since round 5, ``crl_frame_stack_update_to``): the reference's own data flow -- its
``self.current_obs = self.current_obs.roll(...)`` binds a NEW tensor on every update -- as a ping-pong of two buffers: the
kept planes are one contiguous run per env in the source and in the destination, so the shift is a linear copy (2.16 ms for
65 536 x (4, 84, 84) float32 = 0.78 of HBM, against 2.96 ms for the in-place column walk).  The tensor returned by ``get()`` /
``update()`` stays intact through the NEXT update and is recycled by the one after (the reference's old tensor lives as long as
someone holds it, with its finished envs zeroed by the mask multiply).  ``out_of_place=False``: one buffer, updated in place
(``crl_frame_stack_update``); the returned tensor is the live buffer and changes with the next ``update``.

``device="cpu"`` keeps the class usable with host-resident vector envs (numpy observations); that path is
plain tensor arithmetic and is not part of the GPU hot path.  A CUDA device without the library raises.
"""
import ctypes as C

import numpy as np
import torch

from . import _native as N


class FrameStackTensor:
    def __init__(self, num_envs, obs_shape, frame_stack, device, out_of_place=True):
        self.device = torch.device(device)
        self.out_of_place = bool(out_of_place)
        self._spare = None
        self.num_envs, self.frame_stack = int(num_envs), int(frame_stack)
        self.num_channels = int(obs_shape[0])
        self.plane_shape = tuple(int(d) for d in obs_shape[1:])
        self.obs_shape = (self.num_channels * self.frame_stack, *self.plane_shape)
        self.current_obs = torch.zeros((self.num_envs, *self.obs_shape), dtype=torch.float32, device=self.device)
        self._hw = int(np.prod(self.plane_shape)) if self.plane_shape else 1
        self._lib = N.load() if self.device.type == "cuda" else None

    def reset(self):
        self.current_obs.zero_()

    def get(self):
        return self.current_obs

    # ------------------------------------------------------------------ update
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

    def update(self, obs, mask=None):
        o, m = self._as_obs(obs), self._as_mask(mask)
        if self._lib is not None:
            self._update_hip(o, m)
        else:
            self._update_host(o, m)
        return self.current_obs

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
            if self.out_of_place:
                # the reference's own data flow (`self.current_obs = self.current_obs.roll(...)`: a new tensor per update), as a ping-pong of two
                # buffers: the tensor handed out by the LAST update stays intact through this one and is recycled by the next
                if self._spare is None:
                    self._spare = torch.empty_like(self.current_obs)
                dst = self._spare
                N.check(self._lib.crl_frame_stack_update_to(C.c_void_p(dst.data_ptr()), C.c_void_p(self.current_obs.data_ptr()),
                                                            C.c_void_p(o.data_ptr()), *tail))
                self._spare, self.current_obs = self.current_obs, dst
            else:
                N.check(self._lib.crl_frame_stack_update(C.c_void_p(self.current_obs.data_ptr()), C.c_void_p(o.data_ptr()), *tail))

    def _update_host(self, o, m):
        c, buf = self.num_channels, self.current_obs
        kept = buf[:, c:]
        if m is not None:
            kept = kept * m.reshape(self.num_envs, *([1] * (buf.dim() - 1)))
        else:
            kept = kept.clone()
        buf[:, :buf.shape[1] - c] = kept
        buf[:, buf.shape[1] - c:] = o.to(torch.float32)
