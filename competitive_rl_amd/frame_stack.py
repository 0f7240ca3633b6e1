"""Device-side FrameStackTensor (reference competitive_rl/utils/utils.py:145-173).

Same class name, constructor and methods as the reference; ``update`` accepts the device
tensors the HIP env returns (no numpy round trip, no H2D copy) as well as numpy arrays.
Semantics: ``current_obs *= mask`` (mask = 0 where the episode ended -> history erased),
roll by ``num_channels`` along dim 1, newest observation in the last channels.
"""
import numpy as np
import torch


class FrameStackTensor:
    def __init__(self, num_envs, obs_shape, frame_stack, device):
        self.num_channels = obs_shape[0]
        self.obs_shape = (obs_shape[0] * frame_stack, *obs_shape[1:])
        self.current_obs = torch.zeros(num_envs, *self.obs_shape, device=device, dtype=torch.float)
        self.mask_shape = [1] * self.current_obs.dim()
        self.mask_shape[0] = -1
        self.device = device

    def reset(self):
        self.current_obs.fill_(0)

    def update(self, obs, mask=None):
        if mask is not None:
            if isinstance(mask, np.ndarray):
                mask = torch.from_numpy(mask)
            mask = mask.to(self.current_obs.device, torch.float).reshape(self.mask_shape)
            self.current_obs *= mask
        self.current_obs = self.current_obs.roll(shifts=-self.num_channels, dims=1)
        if isinstance(obs, np.ndarray):
            obs = torch.from_numpy(obs.astype(np.float32))
        self.current_obs[:, -self.num_channels:] = obs.to(self.current_obs.device, torch.float)
        return self.current_obs

    def get(self):
        return self.current_obs
