"""Built-in CNN opponents on the device (reference competitive_rl/utils/policy_serving.py:10-66,
utils/network.py:73-93, pong/builtin_policies.py:61-91).

``Policy`` has the reference's constructor and call protocol.  The forward pass -- the policy's own
four-frame stack, the network, argmax -- is hand-written HIP behind the C ABI (``crl_policy_*`` in
include/crl.h); there is no torch model and no CPU path in this module.

``use_light_model=True``: LightActorCritic (WEAK, MEDIUM; one fused kernel).  ``use_light_model=False``: the
full-size ActorCritic (network.py:14-50; three fp32 MFMA GEMM kernels, csrc/pong_policy_full.hip) for checkpoints a
user trained with the reference -- the reference tree itself ships none for it (STRONG / ALPHA_PONG's files
are missing), so without a checkpoint it starts from the reference's orthogonal initialisation.
"""
import ctypes as C
import logging
import os

import numpy as np
import torch

from . import _native as N

ASSETS = os.path.join(N.PKG, "assets")
BUILTIN_CHECKPOINTS = {"WEAK": os.path.join(ASSETS, "pong_policy_weak.npz"),
                       "MEDIUM": os.path.join(ASSETS, "pong_policy_medium.npz")}
_KEYS = ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "actor_w", "actor_b")
_SHAPES = {"conv1_w": (16, 4, 4, 4), "conv1_b": (16,), "conv2_w": (16, 16, 2, 2), "conv2_b": (16,), "actor_w": (3, 1600),
           "actor_b": (3,)}


def load_light_weights(checkpoint_path):
    """``.npz`` written by assets/gen_policy_weights.py, or a reference checkpoint
    (``torch.save({"model": state_dict, ...})``, policy_serving.py:33-36)."""
    assert os.path.isfile(checkpoint_path), checkpoint_path
    if checkpoint_path.endswith(".npz"):
        z = np.load(checkpoint_path)
        w = {k: z[k] for k in _KEYS}
    else:
        sd = torch.load(checkpoint_path, map_location="cpu", weights_only=False)["model"]
        w = {"conv1_w": sd["conv1.weight"], "conv1_b": sd["conv1.bias"], "conv2_w": sd["conv2.weight"], "conv2_b": sd["conv2.bias"],
             "actor_w": sd["actor_linear.weight"], "actor_b": sd["actor_linear.bias"]}
        w = {k: v.detach().cpu().numpy() for k, v in w.items()}
    w = {k: np.ascontiguousarray(v, np.float32) for k, v in w.items()}
    for k in _KEYS:
        if w[k].shape != _SHAPES[k]:
            raise ValueError(f"{checkpoint_path}: {k} has shape {w[k].shape}, LightActorCritic on (4, 42, 42) needs {_SHAPES[k]}")
    return w


_FULL_KEYS = ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "conv3_w", "conv3_b", "actor_w", "actor_b")
_FULL_SHAPES = {"conv1_w": (16, 4, 4, 4), "conv1_b": (16,), "conv2_w": (32, 16, 4, 4), "conv2_b": (32,), "conv3_w": (256, 32, 11, 11),
                "conv3_b": (256,), "actor_w": (3, 256), "actor_b": (3,)}


def load_full_weights(checkpoint_path):
    """ActorCritic tensors from an ``.npz`` with the keys of ``_FULL_KEYS`` or from a reference checkpoint."""
    assert os.path.isfile(checkpoint_path), checkpoint_path
    if checkpoint_path.endswith(".npz"):
        z = np.load(checkpoint_path)
        w = {k: z[k] for k in _FULL_KEYS}
    else:
        sd = torch.load(checkpoint_path, map_location="cpu", weights_only=False)["model"]
        names = {"conv1": "conv1", "conv2": "conv2", "conv3": "conv3", "actor": "actor_linear"}
        w = {}
        for short, long in names.items():
            w[short + "_w"], w[short + "_b"] = sd[long + ".weight"].detach().cpu().numpy(), sd[long + ".bias"].detach().cpu().numpy()
    return check_full_weights(w, checkpoint_path)


def check_full_weights(w, origin="weights"):
    w = {k: np.ascontiguousarray(w[k], np.float32) for k in _FULL_KEYS}
    for k in _FULL_KEYS:
        if w[k].shape != _FULL_SHAPES[k]:
            raise ValueError(f"{origin}: {k} has shape {w[k].shape}, ActorCritic on (4, 42, 42) needs {_FULL_SHAPES[k]}")
    return w


def _random_full_weights():
    """No checkpoint: orthogonal weights (gain sqrt 2 for the convolutions, 0.01 for the actor), zero biases (network.py:18-38)."""
    w = {}
    for k, gain in (("conv1", 2.0 ** 0.5), ("conv2", 2.0 ** 0.5), ("conv3", 2.0 ** 0.5), ("actor", 0.01)):
        t = torch.empty(_FULL_SHAPES[k + "_w"])
        torch.nn.init.orthogonal_(t, gain=gain)
        w[k + "_w"], w[k + "_b"] = t.numpy(), np.zeros(_FULL_SHAPES[k + "_b"], np.float32)
    return check_full_weights(w)


def _random_light_weights():
    """No checkpoint: the reference builds the model with torch's default initialisation."""
    c1, c2, fc = torch.nn.Conv2d(4, 16, 4, 2), torch.nn.Conv2d(16, 16, 2, 2), torch.nn.Linear(1600, 3)
    w = {"conv1_w": c1.weight, "conv1_b": c1.bias, "conv2_w": c2.weight, "conv2_b": c2.bias, "actor_w": fc.weight, "actor_b": fc.bias}
    return {k: np.ascontiguousarray(v.detach().numpy(), np.float32) for k, v in w.items()}


class Policy:
    def __init__(self, single_obs_space, single_action_space, num_envs, checkpoint_path="", frame_stack=4, use_light_model=False,
                 device=None, weights=None):
        """``weights``: optional dict of float32 arrays in torch layout instead of a checkpoint file (tests)."""
        if not torch.cuda.is_available():
            raise RuntimeError("competitive_rl_amd.Policy needs a ROCm GPU; there is no CPU fallback")
        self.num_envs = int(num_envs)
        self.obs_shape = tuple(single_obs_space.shape)
        if self.obs_shape != (1, 42, 42) or frame_stack != 4 or single_action_space.n != 3:
            raise ValueError("the built-in opponents are trained on (1, 42, 42) frames, a stack of 4 and 3 actions")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.use_light_model = bool(use_light_model)
        if weights is not None:
            self.weights = ({k: np.ascontiguousarray(weights[k], np.float32) for k in _KEYS} if use_light_model
                            else check_full_weights(weights))
        elif checkpoint_path:
            self.weights = load_light_weights(checkpoint_path) if use_light_model else load_full_weights(checkpoint_path)
        else:
            logging.warning("Loading a policy without checkpoint!")
            self.weights = _random_light_weights() if use_light_model else _random_full_weights()
        self._L = N.load()
        h = C.c_void_p()
        ptr = [self.weights[k].ctypes.data_as(C.c_void_p) for k in (_KEYS if use_light_model else _FULL_KEYS)]
        create = self._L.crl_policy_create if use_light_model else self._L.crl_policy_create_full
        with torch.cuda.device(self.device):
            N.check(create(self.device.index or 0, self.num_envs, *ptr, C.byref(h)))
        self._h = h
        self._actions = torch.zeros((self.num_envs,), dtype=torch.int32, device=self.device)
        self._logits = torch.zeros((self.num_envs, 3), dtype=torch.float32, device=self.device)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "_h", None):
            self._L.crl_policy_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self):
        N.check(self._L.crl_policy_reset(self._h, self._stream()))

    def _frames(self, obs):
        if isinstance(obs, np.ndarray):
            obs = torch.from_numpy(np.ascontiguousarray(obs))
        obs = obs.to(self.device)
        if obs.dtype != torch.uint8:  # DummyVecEnv hands float32 arrays holding 0..255 integers
            obs = obs.to(torch.uint8)
        # (N, C, 42, 42): the newest plane is the last channel (C = 1 under the reference's cPongDouble wrappers)
        obs = obs.reshape(self.num_envs, 42, 42) if obs.dim() != 4 else obs[:, -1]
        if obs.stride(2) != 1 or obs.stride(1) != 42 or obs.stride(0) % 4 or obs.data_ptr() % 4:
            obs = obs.contiguous()
        return obs

    def act_device(self, obs, out=None, want_logits=False):
        """Push ``obs`` (N, 1, 42, 42) onto the stack and write the greedy actions (int32) into ``out``
        (any int32 device view with one element per env, e.g. ``actions[:, 1]`` of an (N, 2) tensor;
        default: an internal (N,) tensor).  No host synchronisation."""
        f = self._frames(obs)
        out = self._actions if out is None else out
        assert out.dtype == torch.int32 and out.device == self.device and out.shape[0] == self.num_envs and out.numel() == self.num_envs
        stride = out.stride(0) if self.num_envs > 1 else 1
        N.check(self._L.crl_policy_act(self._h, C.c_void_p(f.data_ptr()), f.stride(0), C.c_void_p(out.data_ptr()), stride,
                                       C.c_void_p(self._logits.data_ptr()) if want_logits else None, self._stream()))
        return out

    def logits(self):
        """Logits of the last ``act_device(..., want_logits=True)`` call, float32 (N, 3)."""
        return self._logits

    def get_stack(self):
        out = torch.empty((self.num_envs, 4, 42, 42), dtype=torch.uint8, device=self.device)
        N.check(self._L.crl_policy_get_stack(self._h, C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def set_stack(self, stack):
        s = torch.as_tensor(stack).to(self.device, torch.uint8).contiguous()
        assert tuple(s.shape) == (self.num_envs, 4, 42, 42)
        N.check(self._L.crl_policy_set_stack(self._h, C.c_void_p(s.data_ptr()), self._stream()))
        torch.cuda.current_stream(self.device).synchronize()

    def compute_action(self, obs, deterministic=True):
        """Reference protocol (policy_serving.py:48-56): actions (N, 1) for a whole STACKED observation (N, 4, 42, 42) -- greedy, or
        sampled from the softmax of the logits -- without touching the policy's own frame stack (which only ``__call__`` / ``act_device``
        advance).  The network runs as in ``act_device``: the stack is put into the kernel's ring shifted by one plane, the newest plane
        is pushed (the ring then holds exactly ``obs``), and the ring is restored afterwards.  Not a hot path (three small copies)."""
        o = torch.from_numpy(np.ascontiguousarray(obs)) if isinstance(obs, np.ndarray) else obs
        o = o.to(self.device)
        if tuple(o.shape) != (self.num_envs, 4, 42, 42):
            raise ValueError(f"compute_action takes the stacked observation ({self.num_envs}, 4, 42, 42), got {tuple(o.shape)}")
        if o.dtype != torch.uint8:
            o = o.to(torch.uint8)  # (0..255 integers, as FrameStackTensor holds them)
        saved = self.get_stack()
        self.set_stack(torch.roll(o, shifts=1, dims=1))          # planes 0..2 of `obs` in ring positions 1..3; position 0 rolls out
        greedy = self.act_device(o[:, 3].contiguous(), out=torch.empty((self.num_envs,), dtype=torch.int32, device=self.device), want_logits=True)
        if deterministic:
            actions = greedy.to(torch.int64)
        else:
            actions = torch.distributions.Categorical(logits=self._logits).sample()
        self.set_stack(saved)
        return actions.view(-1, 1)

    def __call__(self, obs):
        """Reference protocol (policy_serving.py:58-66): numpy (N, 1) int64, or an int for one env."""
        a = self.act_device(obs).to(torch.int64)
        if self.num_envs == 1:
            return a.item()
        return a.view(-1, 1).cpu().numpy()
