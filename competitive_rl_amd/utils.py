"""``step_envs``: one vector-env step plus the books a trainer keeps around it (interface of reference
competitive_rl/utils/utils.py:23-60).  Behaviour pinned by tests/golden/step_envs.npz, recorded from
the reference's function.

What the call does, in terms of its arguments (all updated in place or returned, as in the reference):

* ``episode_rewards`` (N, A): running return per env and agent -- the step's rewards are added, rows of envs whose
  episode ended are recorded and then cleared;
* ``reward_recorder`` / ``length_recorder``: one entry per finished episode (its return row; ``info["num_steps"]``
  when the env reports it -- CartPole-style envs do not);
* ``total_episodes`` / ``total_steps``: counters (episodes finished; N env-steps per call);
* ``frame_stack_tensor``: gets the learner's observation (agent 0's of a two-agent tuple) with a mask that
  erases the history of envs that just restarted;
* returns ``(obs, reward, done, info, masks, total_episodes, total_steps, episode_rewards)`` with ``done``
  reduced to one flag per env and ``masks`` = 1 - done as an (N, 1) float tensor on ``device``.

With the HIP env everything the step produced is already in HBM: the device ``FrameStackTensor`` is bound to the env on
the first call (``FrameStackTensor.bind``: ``envs.step`` then draws the stack's next state in the launch that draws the
observation, and the update here is a pointer swap; other envs / stacks: one kernel), masks are built on the device, and
``episode_rewards`` may itself be a device tensor.  The only host traffic is the N done flags (the recorders are host lists) and, for a numpy
``episode_rewards``, the N x A rewards.
"""
import numpy as np
import torch

__all__ = ["step_envs", "evaluate"]


def _per_env_done(done):
    """One flag per env: a vector env with A agents reports (N, A) (DummyVecEnv broadcasts the env's scalar),
    a SubprocVecEnv-style one (N,)."""
    if isinstance(done, torch.Tensor):
        return done.bool().all(dim=1) if done.dim() == 2 else done.bool()
    done = np.asarray(done)
    return done.astype(bool).all(axis=1) if done.ndim == 2 else done.astype(bool)


class _Returns:
    """The running ``episode_rewards`` array, host (numpy) or device (torch), behind one interface."""

    def __init__(self, arr):
        self.arr, self.on_device = arr, isinstance(arr, torch.Tensor)

    def add(self, reward):
        if self.on_device:
            r = reward if isinstance(reward, torch.Tensor) else torch.as_tensor(np.asarray(reward))
            self.arr += r.to(self.arr.device, self.arr.dtype).reshape(self.arr.shape)
        else:
            r = reward.detach().cpu().numpy() if isinstance(reward, torch.Tensor) else np.asarray(reward)
            self.arr += r.reshape(self.arr.shape)

    def rows(self, idx):
        """Copies of the listed rows as numpy arrays (what the reward recorder keeps)."""
        if len(idx) == 0:
            return []
        picked = self.arr[torch.as_tensor(idx, device=self.arr.device)].cpu().numpy() if self.on_device else self.arr[idx]
        return [np.array(row) for row in picked]

    def clear(self, ended, ended_host):
        if self.on_device:
            e = ended if isinstance(ended, torch.Tensor) else torch.as_tensor(ended_host)
            self.arr.masked_fill_(e.to(self.arr.device).reshape(-1, 1), 0)
        else:
            self.arr[ended_host] = 0


def step_envs(cpu_actions, envs, episode_rewards, frame_stack_tensor, reward_recorder, length_recorder, total_steps,
              total_episodes, device, test):
    shape_before = episode_rewards.shape
    if not getattr(frame_stack_tensor, "_bind_tried", True):
        frame_stack_tensor.bind(envs)  # (once: from now on envs.step draws the stack's next state, frame_stack.py)
    obs, reward, done, info = envs.step(cpu_actions)
    learner_obs = obs[0] if isinstance(obs, tuple) else obs  # two-agent Pong: the learner is agent 0
    num_envs = learner_obs.shape[0]

    ended = _per_env_done(done)
    early = getattr(envs, "done_host", None)  # the HIP Pong env hands the flags over while the step's draw still runs (vec_env.py)
    ended_host = early() if early is not None else (ended.cpu().numpy() if isinstance(ended, torch.Tensor) else ended)

    returns = _Returns(episode_rewards)
    returns.add(reward)
    finished = np.flatnonzero(ended_host)
    reward_recorder.extend(returns.rows(finished))
    for i in finished:
        entry = info[int(i)]
        if "num_steps" in entry:
            length_recorder.append(entry["num_steps"])
    total_episodes += len(finished)
    returns.clear(ended, ended_host)
    assert episode_rewards.shape == shape_before
    total_steps += num_envs

    if isinstance(ended, torch.Tensor):
        masks = (~ended).to(device=device, dtype=torch.float32).reshape(num_envs, 1)
    else:
        masks = torch.from_numpy((~ended_host).astype(np.float32)).to(device).reshape(num_envs, 1)
    stack_mask = masks if test else masks.reshape(num_envs, 1, 1, 1)
    if getattr(frame_stack_tensor, "_env", None) is not None:
        frame_stack_tensor.update(learner_obs, stack_mask, _from_env=envs)  # (a bound stack: the pointer swap)
    else:
        frame_stack_tensor.update(learner_obs, stack_mask)
    return obs, reward, ended, info, masks, total_episodes, total_steps, episode_rewards


def evaluate(trainer, eval_envs, frame_stack, num_episodes=10, seed=0):
    """Runs ``trainer``'s greedy policy on ``eval_envs`` until ``num_episodes`` episodes have ended and returns
    ``(reward_recorder, episode_length_recorder)`` -- the reference's evaluation loop (utils/utils.py:102-142), call for call:
    a fresh ``FrameStackTensor``, ``eval_envs.seed(seed)``, ``reset()``, the first observation pushed, then ``step_envs`` with the
    actions of ``trainer.compute_action(stack, deterministic=True)[1]``.

    With the HIP env nothing of the loop crosses PCIe but the done flags: the stack is bound to the env by the first ``step_envs``
    call (drawn by the step), the actions go from the policy to the env as a device tensor, the running returns live on the device.
    ``trainer`` needs ``.device`` and ``.compute_action(obs, deterministic=True) -> (..., actions, ...)`` as the reference's trainers have."""
    from .frame_stack import FrameStackTensor

    device = torch.device(trainer.device)
    fst = FrameStackTensor(eval_envs.num_envs, eval_envs.observation_space.shape, frame_stack, device)
    on_device = device.type == "cuda"

    def get_action():
        obs = fst.get()
        with torch.no_grad():
            act = trainer.compute_action(obs, deterministic=True)[1]
        act = torch.as_tensor(act).reshape(-1)
        return act.to(device=device, dtype=torch.int32) if on_device else act.cpu().numpy()

    reward_recorder, episode_length_recorder = [], []
    episode_rewards = torch.zeros((eval_envs.num_envs, 1), dtype=torch.float32, device=device) if on_device else np.zeros([eval_envs.num_envs, 1], dtype=np.float64)
    total_steps = total_episodes = 0
    eval_envs.seed(seed)
    obs = eval_envs.reset()
    fst.update(obs[0] if isinstance(obs, tuple) else obs)
    while True:
        out = step_envs(get_action(), eval_envs, episode_rewards, fst, reward_recorder, episode_length_recorder, total_steps, total_episodes,
                        device, frame_stack == 1)
        total_episodes, total_steps, episode_rewards = out[5], out[6], out[7]
        if total_episodes >= num_episodes:
            break
    return reward_recorder, episode_length_recorder
