"""``evaluate`` (reference utils/utils.py:102-142): the evaluation loop around ``step_envs`` + ``FrameStackTensor``.  CPU: over a small
scripted single-agent env, against the loop written out by hand; ``-m gpu``: over cPongTournament-v0 on the HIP env, where the stack is
bound to the env by the first ``step_envs`` call and drawn by the step."""
import numpy as np
import pytest
import torch


class ScriptedEnv:
    """N single-agent envs, episodes of scripted lengths, reward = env index + 1 per step, observation = a counter plane"""

    def __init__(self, n, lengths):
        from competitive_rl_amd import spaces

        self.num_envs, self.lengths = n, lengths
        self.observation_space = spaces.Box(0, 255, (1, 3, 3))
        self.action_space = spaces.Discrete(3)
        self.seeded = None

    def seed(self, s):
        self.seeded = s

    def _obs(self):
        return np.stack([np.full((1, 3, 3), (self.t[i] * 7 + i) % 256, np.uint8) for i in range(self.num_envs)]).astype(np.float32)

    def reset(self):
        self.t, self.ep = np.zeros(self.num_envs, int), np.zeros(self.num_envs, int)
        self.actions_seen = []
        return self._obs()

    def step(self, actions):
        self.actions_seen.append(np.asarray(actions).copy())
        self.t += 1
        done = np.array([self.t[i] >= self.lengths[i][self.ep[i] % len(self.lengths[i])] for i in range(self.num_envs)])
        rew = np.arange(1, self.num_envs + 1, dtype=np.float32).reshape(-1, 1)
        infos = [{"num_steps": int(self.t[i])} for i in range(self.num_envs)]
        for i in np.flatnonzero(done):
            self.t[i], self.ep[i] = 0, self.ep[i] + 1
        return self._obs(), rew, done, infos


class GreedyTrainer:
    def __init__(self, device):
        self.device, self.calls = device, 0

    def compute_action(self, obs, deterministic=True):
        assert deterministic and obs.dtype == torch.float32
        self.calls += 1
        return None, (obs.reshape(obs.shape[0], -1).sum(1).long() % 3), None   # (values, actions, log-probs) as the reference's trainers return


def test_evaluate_runs_the_references_loop():
    import competitive_rl_amd as crl

    lengths = [[3, 5], [4], [7, 2, 2]]
    env = ScriptedEnv(3, lengths)
    tr = GreedyTrainer("cpu")
    rewards, steps = crl.evaluate(tr, env, frame_stack=2, num_episodes=6, seed=123)
    assert env.seeded == 123
    # by hand: env 0 ends at t = 3, 8, 11, 16 ...; env 1 at 4, 8, 12 ...; env 2 at 7, 9, 11 ...  The loop stops after the step in which the
    # sixth episode ended (t = 9: episodes ended at 3, 4, 7, 8, 8, 9), recorded in env order within a step
    assert [int(s) for s in steps] == [3, 4, 7, 5, 4, 2]
    assert [float(r[0]) for r in rewards] == [3 * 1.0, 4 * 2.0, 7 * 3.0, 5 * 1.0, 4 * 2.0, 2 * 3.0]
    assert tr.calls == 9 and len(env.actions_seen) == 9
    # the policy saw the stack: two planes, the older one erased where an episode had just ended
    assert all(a.shape == (3,) for a in env.actions_seen)


@pytest.mark.gpu
def test_evaluate_on_the_hip_tournament_env_binds_the_stack():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    import competitive_rl_amd as crl

    from competitive_rl_amd import frame_stack as fs_mod

    made = []

    class Recording(fs_mod.FrameStackTensor):   # (evaluate's stack is a local of the call: keep a handle on it)
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            made.append(self)

    n = 96
    envs = crl.make_envs("cPongTournament-v0", num_envs=n, log_dir=None, seed=4)
    tr = GreedyTrainer(envs.env.device)
    original = fs_mod.FrameStackTensor
    fs_mod.FrameStackTensor = Recording
    try:
        rewards, steps = crl.evaluate(tr, envs, frame_stack=4, num_episodes=20, seed=9)
    finally:
        fs_mod.FrameStackTensor = original
    assert len(rewards) >= 20 and len(steps) == len(rewards)
    assert all(np.asarray(r).shape == (1,) and abs(float(r[0])) <= 21 for r in rewards) and all(int(s) > 0 for s in steps)
    (fst,) = made
    assert fst._env is not None and fst._env() is envs.env and fst.frame_stack == 4 and fst.fused_updates >= tr.calls - 1 > 50   # every update but possibly the first was drawn by the step
    assert tuple(fst.get().shape) == (n, 4, 42, 42) and fst.get().dtype == torch.float32
    envs.close()
