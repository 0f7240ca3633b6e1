"""The two-policy match loops (reference pong/evaluate.py:6-88) against ``tests/golden/pong_evaluate.npz``, recorded from the
reference's own functions over its DummyVecEnv / one wrapped env (tests/golden/gen_pong_evaluate_golden.py).

One of the two policies is a pure function of the observation (a weighted pixel sum mod 3), the other the rule-based bat (999):
a replay reproduces the recorded game results only if every observation it showed the policy was the reference's, pixel for pixel,
through whole 21-point matches.  CPU: this package's loops over the oracle as the env; ``-m gpu``: over the HIP env, the
observation-driven policy evaluated on the device (actions never leave HBM)."""
import os

import numpy as np
import pytest
import torch

from test_step_envs_golden import OracleVecEnv, blank_atlas

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CHEAT = 999


def obs_policy(obs):
    """The generator's policy (gen_pong_evaluate_golden.py::obs_policy), for numpy observations and for device tensors."""
    if isinstance(obs, torch.Tensor):
        batch = obs.dim() == 4
        o = obs.to(torch.float64).reshape(obs.shape[0] if batch else 1, -1)
        w = (torch.arange(o.shape[1], device=o.device) % 7 + 1).to(torch.float64)
        a = (o * w).sum(1).to(torch.int64) % 3
        return a if batch else int(a[0])
    o = np.asarray(obs, np.float64)
    batch = o.ndim == 4
    o = o.reshape(o.shape[0] if batch else 1, -1)
    a = ((o * (np.arange(o.shape[1]) % 7 + 1)).sum(1).astype(np.int64) % 3)
    return a if batch else int(a[0])


def padded(g, tag):
    """The recorded serve draws plus spare columns (a vector env restarts by itself after the last scored episode too)."""
    return tuple(np.concatenate([g[f"{tag}_draw_{k}"], np.zeros((g[f"{tag}_draw_{k}"].shape[0], 4), g[f"{tag}_draw_{k}"].dtype)], 1) for k in ("u", "bx", "by"))


class Traced:
    """Records what the loop sends to and gets from ``step`` (compared with the reference's trace step for step)."""

    def __init__(self, env, n):
        self.env, self.num_envs, self.acts, self.rew, self.done = env, n, [], [], []
        if hasattr(env, "done_host"):
            self.done_host = env.done_host

    def reset(self):
        return self.env.reset()

    def step(self, actions):
        o, r, d, info = self.env.step(actions)
        h = lambda x: x.cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)  # noqa: E731
        self.acts.append(h(actions).copy()), self.rew.append(h(r).copy()), self.done.append(h(d).copy())
        return o, r, d, info


def check_batch(g, tag, make_env):
    import competitive_rl_amd as crl
    from competitive_rl_amd.pong_evaluate import evaluate_two_policies_in_batch

    assert crl.evaluate_two_policies_in_batch is evaluate_two_policies_in_batch
    N = g[f"{tag}_acts"].shape[1]
    env = Traced(make_env(N, padded(g, tag)), N)
    rule = lambda obs: [CHEAT] * N  # noqa: E731
    c0, c1 = (rule, obs_policy) if int(g[f"{tag}_side0_is_rule"]) else (obs_policy, rule)
    r0, r1 = evaluate_two_policies_in_batch(c0, c1, env, int(g[f"{tag}_num_episodes"]))
    T = len(g[f"{tag}_acts"])
    for t in range(min(T, len(env.acts))):  # (first divergence, if any, before the totals)
        assert np.array_equal(env.acts[t], g[f"{tag}_acts"][t]), (tag, t)
        assert np.array_equal(env.rew[t], g[f"{tag}_rew"][t]), (tag, t)
        assert np.array_equal(np.asarray(env.done[t]).astype(bool).reshape(N, -1).all(1), g[f"{tag}_done"][t].all(1)), (tag, t)
    assert len(env.acts) == T
    assert [float(x) for x in r0] == g[f"{tag}_result0"].tolist() and [float(x) for x in r1] == g[f"{tag}_result1"].tolist()
    assert all(isinstance(x, int) for x in r0[:3] + r1[:3]) and sum(r0[:3]) >= int(g[f"{tag}_num_episodes"])


def check_single(g, make_env):
    from competitive_rl_amd.pong_evaluate import evaluate_two_policies

    env = Traced(make_env(1, padded(g, "single")), 1)
    lines = []

    class Console:
        def printMatchInfo(self, name, episode, r):
            lines.append((name, episode, float(r)))

    r0, r1 = evaluate_two_policies(obs_policy, lambda obs: CHEAT, env, int(g["single_num_episodes"]), print_console=Console(), env_name="pong")
    T = len(g["single_acts"])
    for t in range(min(T, len(env.acts))):
        assert np.array_equal(env.acts[t][0], g["single_acts"][t]) and np.array_equal(env.rew[t][0], g["single_rew"][t]), t
    assert len(env.acts) == T
    assert [float(x) for x in r0] == g["single_result0"].tolist() and [float(x) for x in r1] == g["single_result1"].tolist()
    assert [ln[1] for ln in lines] == list(range(int(g["single_num_episodes"]))) and sum(ln[2] for ln in lines) == r0[3]
    with pytest.raises(ValueError):
        evaluate_two_policies(obs_policy, obs_policy, Traced(None, 2), 1)


def _oracle_env(n, draws):
    g = dict(draw_u=draws[0], draw_bx=draws[1], draw_by=draws[2])
    return OracleVecEnv(g, n, 42)


def test_match_loops_reproduce_the_reference_over_the_oracle():
    g = np.load(os.path.join(G, "pong_evaluate.npz"))
    assert int(g["resized_dim"]) == 42
    check_batch(g, "batch_a", _oracle_env)
    check_batch(g, "batch_b", _oracle_env)
    check_single(g, _oracle_env)


def test_gym_style_single_env_protocol():
    """A gym-style env (tuple observation, scalar done, explicit reset per episode): the reference's own calling convention."""
    from competitive_rl_amd.pong_evaluate import evaluate_two_policies

    class Toy:
        def __init__(self):
            self.resets, self.t = 0, 0

        def reset(self):
            self.resets += 1
            self.t = 0
            return (np.zeros(1), np.ones(1))

        def step(self, action):
            self.t += 1
            assert action == [int(self.resets), 7]
            return (np.zeros(1), np.ones(1)), [1.0 if self.resets == 1 else (-1.0 if self.resets == 2 else 0.0), 0.5], self.t == 3, {}

    class P0:
        def __init__(self, env):
            self.env, self.n = env, 0

        def reset(self):
            self.n += 1

        def __call__(self, obs):
            return self.env.resets

    env = Toy()
    p0 = P0(env)
    r0, r1 = evaluate_two_policies(p0, lambda obs: 7, env, 3)
    assert r0 == [1, 1, 1, 0.0] and r1 == [1, 1, 1, 4.5] and env.resets == 3 and p0.n == 3


# --------------------------------------------------------------------------------------------- GPU
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")


def _hip_env(n, draws):
    import competitive_rl_amd as crl

    env = crl.HipPongVecEnv(n, mode="wrapped", resized_dim=42, frame_stack=1, score_atlas=blank_atlas())
    env.set_replay(*draws)
    return env


@pytest.mark.gpu
def test_match_loops_reproduce_the_reference_through_hip():
    """Whole recorded matches replayed on the GPU: the observation-driven policy reads the HIP env's device observations, its
    actions go back as a device tensor; game results, and every action / reward / done on the way, equal the reference's."""
    _need_gpu()
    g = np.load(os.path.join(G, "pong_evaluate.npz"))
    check_batch(g, "batch_a", _hip_env)
    check_batch(g, "batch_b", _hip_env)
    check_single(g, _hip_env)


@pytest.mark.gpu
def test_batch_match_of_builtin_opponents_on_the_device():
    """RULE_BASED against the MEDIUM CNN opponent (device ``Policy``) over 2 048 envs until 2 048 episodes have ended: the books
    balance (side 1's wins are side 0's losses, the cumulative rewards are opposite integers of at most 21 points per match)."""
    _need_gpu()
    import competitive_rl_amd as crl
    from competitive_rl_amd.tournament import get_compute_action_function

    n = 2048
    envs = crl.make_envs("cPongDouble-v0", num_envs=n, frame_stack=None, log_dir=None, resized_dim=42, seed=3)
    medium = get_compute_action_function("MEDIUM", n, device=envs.device)
    rule = get_compute_action_function("RULE_BASED", n, device=envs.device)
    r0, r1 = crl.evaluate_two_policies_in_batch(rule, medium.act_device, envs, n)   # (act_device: the opponent's actions stay on the device)
    assert sum(r0[:3]) >= n and r0[0] == r1[2] and r0[2] == r1[0] and r0[1] == r1[1] and r0[3] == -r1[3]
    assert abs(r0[3]) <= 21 * sum(r0[:3]) and r0[3] == int(r0[3])
    # (measured: the MEDIUM network wins 2 034 of 2 048 matches against the rule-based bat)
    assert r1[0] > r0[0]
    envs.close()


@pytest.mark.gpu
def test_vis_script_flow_on_the_single_env_handle():
    """vis.py:28-40 call for call: ``env = make_envs("cPongDouble-v0", num_envs=1, asynchronous=False, frame_stack=None, log_dir=...).envs[0]``,
    ``evaluate_two_policies(left, right, env=env, num_episode=N)`` with the gym calls of ONE env (tuple observation, scalar done, ``reset()``
    per episode) -- against the recording of the reference's function over the reference's single wrapped env: same actions, rewards
    and results, i.e. the handle's ``reset()`` after a finished episode does not reset the restarted env a second time."""
    _need_gpu()
    import competitive_rl_amd as crl

    g = np.load(os.path.join(G, "pong_evaluate.npz"))
    envs = crl.make_envs("cPongDouble-v0", num_envs=1, asynchronous=False, frame_stack=None, log_dir=None, resized_dim=42, score_atlas=blank_atlas())
    envs.set_replay(*padded(g, "single"))
    env = envs.envs[0]
    seen = []
    step0 = env.step

    def step(action):
        out = step0(action)
        seen.append((list(int(np.asarray(a).reshape(-1)[0]) for a in action), out[1].cpu().numpy().copy(), out[2]))
        return out

    env.step = step
    r0, r1 = crl.evaluate_two_policies(obs_policy, crl.get_compute_action_function("RULE_BASED"), env=env, num_episode=int(g["single_num_episodes"]))
    T = len(g["single_acts"])
    for t in range(min(T, len(seen))):
        assert seen[t][0] == g["single_acts"][t].tolist() and np.array_equal(seen[t][1], g["single_rew"][t]) and seen[t][2] == bool(g["single_done"][t]), t
    assert len(seen) == T and [float(x) for x in r0] == g["single_result0"].tolist() and [float(x) for x in r1] == g["single_result1"].tolist()
    # the first observation after a reset() is a tuple of the two agents' (1, R, R) frames; a batch of several envs has no per-env step
    o = env.reset()
    assert isinstance(o, tuple) and tuple(o[0].shape) == (1, 42, 42)
    env.close()
    envs.close()
    two = crl.make_envs("cPongDouble-v0", num_envs=2, frame_stack=None, log_dir=None)
    with pytest.raises(NotImplementedError):
        two.envs[0].reset()
    two.close()
    # CarRacing's handle: the flattened observation, car 0's reward, a scalar done
    car = crl.make_envs("cCarRacingDouble-v0", num_envs=1, frame_stack=4, log_dir=None).envs[0]
    oc = car.reset()
    oc2, rc, dc, ic = car.step(car.action_space.sample())
    assert tuple(oc.shape) == tuple(oc2.shape) == (8, 96, 96) and isinstance(dc, bool) and tuple(rc.shape) == (1,)
