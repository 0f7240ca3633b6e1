"""Two-policy match loops over a two-agent Pong env: ``evaluate_two_policies_in_batch`` and ``evaluate_two_policies``
(interface and results of reference competitive_rl/pong/evaluate.py:53-88 and :6-50; pinned by tests/golden/pong_evaluate.npz,
recorded from the reference's functions).

Both return ``(gameResult0, gameResult1)``, each ``[wins, draws, losses, cumulative reward]`` of that side, an episode being
scored by the sign of side 0's return.  ``compute_action{0,1}(obs_of_that_side)`` are the reference's policy callables
(pong/builtin_policies.py: a ``Policy``, ``lambda obs: [CHEAT_CODES] * n``, a random policy ...); here they may also return a
device tensor (``Policy.act_device``-style), in which case the actions never leave HBM.

With the HIP env the loop keeps observations, rewards and the running returns on the device; per step the host sees the N done
flags (handed over behind the dynamics kernel, ``done_host``) and, on steps where an episode ended, the return rows of the
finished envs -- the books themselves are the reference's Python, env by env in index order.
"""
import time

import numpy as np
import torch

from .utils import _per_env_done

__all__ = ["evaluate_two_policies", "evaluate_two_policies_in_batch"]


def _score(result0, result1, r0, r1):
    """pong/evaluate.py:31-42 / :71-82: win / draw / loss by the sign of side 0's return, both sides' cumulative reward."""
    if r0 > 0.0:
        result0[0] += 1
        result1[2] += 1
    elif r0 == 0.0:
        result0[1] += 1
        result1[1] += 1
    else:
        result0[2] += 1
        result1[0] += 1
    result0[3] += r0
    result1[3] += r1


def _actions(a0, a1, device):
    """[num_envs, 2] from the two sides' answers (pong/evaluate.py:60-64); a device tensor when the env lives on one."""
    if device is not None:
        cols = [a.to(device=device, dtype=torch.int32).reshape(-1) if isinstance(a, torch.Tensor)
                else torch.as_tensor(np.asarray(a).reshape(-1).astype(np.int32), device=device) for a in (a0, a1)]
        return torch.stack(cols, dim=1)
    return np.stack([(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)).reshape(-1) for a in (a0, a1)], axis=1)


def evaluate_two_policies_in_batch(compute_action0, compute_action1, envs, num_episodes):
    gameResult0 = [0] * 4  # [0] Win [1] Draw [2] Lose [3] Cumulative Reward
    gameResult1 = [0] * 4
    total_episodes = 0
    obs = envs.reset()
    on_device = isinstance(obs[0], torch.Tensor) and obs[0].is_cuda
    device = obs[0].device if on_device else None
    episode_rewards = (torch.zeros((envs.num_envs, 2), dtype=torch.float64, device=device) if on_device
                       else np.zeros([envs.num_envs, 2], dtype=np.float64))
    early = getattr(envs, "done_host", None)
    while True:
        actions = _actions(compute_action0(obs[0]), compute_action1(obs[1]), device)
        obs, reward, done, info = envs.step(actions)
        ended = _per_env_done(done)
        if on_device:
            episode_rewards += (reward if isinstance(reward, torch.Tensor) else torch.as_tensor(np.asarray(reward))).to(device, torch.float64)
            ended_host = early() if early is not None else ended.cpu().numpy()
        else:
            episode_rewards += reward.detach().cpu().numpy() if isinstance(reward, torch.Tensor) else np.asarray(reward)
            ended_host = ended.cpu().numpy() if isinstance(ended, torch.Tensor) else ended
        finished = np.flatnonzero(ended_host)
        if len(finished):
            rows = episode_rewards[torch.as_tensor(finished, device=device)].cpu().numpy() if on_device else episode_rewards[finished]
            for r in rows:  # (index order, as the reference's loop over the batch)
                _score(gameResult0, gameResult1, r[0], r[1])
                total_episodes += 1
            if on_device:
                episode_rewards.masked_fill_((ended if isinstance(ended, torch.Tensor) else torch.as_tensor(ended_host)).to(device).reshape(-1, 1), 0)
            else:
                episode_rewards[finished] = 0
        if total_episodes >= num_episodes:
            break
    return gameResult0, gameResult1


def evaluate_two_policies(compute_action0, compute_action1, env, num_episode, render=False, print_console=None, env_name="",
                          render_interval=0.05):
    """The one-env loop.  ``env`` is a gym-style two-agent env (``reset() -> (obs0, obs1)``, ``step([a0, a1])`` with a scalar
    ``done``) or a vector env of ONE env from ``make_envs(..., num_envs=1)``: the policies then see the env's own observations
    (the batch axis is taken off, the actions go in as a batch of one) and, since a vector env restarts by itself, only the first
    episode calls ``reset()`` -- the later episodes begin where the reference's per-episode ``env.reset()`` would put them."""
    gameResult0 = [0] * 4
    gameResult1 = [0] * 4
    vec = hasattr(env, "num_envs")
    if vec and env.num_envs != 1:
        raise ValueError("evaluate_two_policies plays ONE env; use evaluate_two_policies_in_batch for a batch")
    early = getattr(env, "done_host", None) if vec else None
    obs = None
    for episode in range(num_episode):
        matchTotalReward = [0.0, 0.0]
        if obs is None or not vec:
            obs = env.reset()
        done = False
        if hasattr(compute_action0, "reset"):
            compute_action0.reset()
        elif hasattr(compute_action1, "reset"):
            compute_action1.reset()
        while not done:
            if vec:
                a0, a1 = compute_action0(obs[0][0]), compute_action1(obs[1][0])
                dev = obs[0].device if isinstance(obs[0], torch.Tensor) and obs[0].is_cuda else None
                obs, reward, d, _ = env.step(_actions(a0, a1, dev))
                done = bool(early()[0]) if (early is not None and dev is not None) else bool(np.asarray(_per_env_done(d).cpu() if isinstance(d, torch.Tensor) else _per_env_done(d))[0])
                reward = (reward.cpu().numpy() if isinstance(reward, torch.Tensor) else np.asarray(reward))[0]
            else:
                obs, reward, done, _ = env.step([compute_action0(obs[0]), compute_action1(obs[1])])
                if isinstance(reward, torch.Tensor):  # (envs.envs[0] of the HIP env hands out device tensors: the books are host floats, as the reference's)
                    reward = reward.detach().cpu().numpy()
            matchTotalReward[0] += reward[0]
            matchTotalReward[1] += reward[1]
            if render:
                time.sleep(render_interval)
                env.render(mode="human")
        _score(gameResult0, gameResult1, matchTotalReward[0], matchTotalReward[1])
        if print_console is None:
            continue
        print_console.printMatchInfo(env_name, episode, matchTotalReward[0])
    return gameResult0, gameResult1
