"""``step_envs`` -- the step of every training loop of the reference (competitive_rl/utils/utils.py:23-60), SURVEY 8f N1.

Same signature, same bookkeeping, same return tuple.  The reference moves every observation host -> device each
step (``torch.from_numpy(obs.astype(float32)).to(device)`` inside ``FrameStackTensor.update``); with the HIP env the
observation is already a device tensor and goes into the (device) ``FrameStackTensor`` without leaving HBM.  What does
cross to the host per step is what the recorders need: the ``done`` flags (N bytes) and, when ``episode_rewards`` is a
numpy array as in the reference's trainers, the rewards (N floats).  Pass a device tensor as ``episode_rewards`` to
keep those on the device too.
"""
import numpy as np
import torch

__all__ = ["step_envs"]


def _to_numpy(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def step_envs(cpu_actions, envs, episode_rewards, frame_stack_tensor, reward_recorder, length_recorder, total_steps,
              total_episodes, device, test):
    """One ``envs.step`` plus the books a trainer keeps around it: running episode returns, finished-episode
    recorders, episode / step totals, and the frame stack (erased where an episode ended)."""
    obs, reward, done, info = envs.step(cpu_actions)
    on_device = isinstance(done, torch.Tensor)
    if isinstance(episode_rewards, torch.Tensor):
        r = reward if isinstance(reward, torch.Tensor) else torch.as_tensor(np.asarray(reward))
        episode_rewards += r.to(episode_rewards.device, episode_rewards.dtype).reshape(episode_rewards.shape)
    else:
        episode_rewards += _to_numpy(reward).reshape(episode_rewards.shape)
    episode_rewards_old_shape = episode_rewards.shape
    if on_device:
        done = done.all(dim=1) if done.dim() == 2 else done
        done_host = done.cpu().numpy()
    else:
        done = np.asarray(done)
        if done.ndim == 2:  # DummyVecEnv: (N, agents); ``not np.isscalar(done[0])`` in the reference
            done = np.all(done, axis=1)
        done_host = done
    for idx in np.nonzero(done_host)[0]:  # finished episodes only
        idx = int(idx)
        reward_recorder.append(_to_numpy(episode_rewards[idx]).copy())
        # envs without a step counter in their info (CartPole) record no length
        if "num_steps" in info[idx]:
            length_recorder.append(info[idx]["num_steps"])
        total_episodes += 1
    if isinstance(episode_rewards, torch.Tensor):
        dm = done if on_device else torch.as_tensor(done_host)
        episode_rewards *= (1.0 - dm.to(episode_rewards.device, episode_rewards.dtype)).reshape(-1, 1)
    else:
        episode_rewards *= (1. - done_host.astype(np.float32)).reshape(-1, 1)
    assert episode_rewards.shape == episode_rewards_old_shape

    first = obs[0] if isinstance(obs, tuple) else obs
    total_steps += first.shape[0]
    if on_device:
        masks = (1.0 - done.to(torch.float32)).to(device).view(-1, 1)
    else:
        masks = torch.from_numpy(1. - done_host.astype(np.float32)).to(device).view(-1, 1)
    # the stack forgets the history of envs that just restarted (mask 0)
    frame_stack_masks = masks.view(-1, 1) if test else masks.view(-1, 1, 1, 1)
    # two-agent Pong hands back a tuple: the learner is agent 0
    frame_stack_tensor.update(first, frame_stack_masks)
    return obs, reward, done, info, masks, total_episodes, total_steps, episode_rewards
