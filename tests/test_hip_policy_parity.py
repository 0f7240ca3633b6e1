"""The HIP opponent-policy kernel (crl_policy_* of include/crl.h) against the numpy oracle and the
vectors recorded from the reference's Policy with the reference's checkpoints.  float32: logits
within 1e-4, actions identical (the recorded traces have a top-2 logit gap >= 2e-3)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "policy_light.npz")
TOL = 1e-4


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")


def make_policy(name, n):
    import competitive_rl_amd.tournament as T

    return T.get_compute_action_function(name.upper(), n)


@pytest.mark.parametrize("name", ["weak", "medium"])
def test_policy_kernel_replays_reference_trace(name):
    _need_gpu()
    g = np.load(GOLD)
    frames, actions, logits = g[name + "_frames"], g[name + "_actions"], g[name + "_logits"]
    n = frames.shape[1]
    pol = make_policy(name, n)
    for t in range(frames.shape[0]):
        obs = torch.from_numpy(frames[t][:, None]).cuda()
        a = pol.act_device(obs, want_logits=True)
        assert np.abs(pol.logits().cpu().numpy() - logits[t]).max() < TOL, t
        assert np.array_equal(a.cpu().numpy(), actions[t]), t
    # the stack the model saw is the last four frames, oldest first -- across episode ends
    st = pol.get_stack().cpu().numpy()
    assert np.array_equal(st, frames[-4:].transpose(1, 0, 2, 3))
    # reference call protocol: numpy (N, 1) int64
    out = pol(frames[0][:, None])
    assert out.shape == (n, 1) and out.dtype == np.int64
    pol.reset()
    assert int(pol.get_stack().max()) == 0
    pol.close()


@pytest.mark.parametrize("n", [1, 2, 4, 5, 6, 7, 9, 333])
def test_policy_kernel_vs_oracle_ragged_sizes(n):
    """Random stacks (workgroups hold five envs: the sizes cover every ragged tail, 1 to 4 envs in the last group), strided
    frame and action views, ring rotation over several calls."""
    _need_gpu()
    from oracle import policy_oracle as P

    w = P.load_weights(os.path.join(ROOT, "competitive_rl_amd", "assets", "pong_policy_medium.npz"))
    ora = P.PolicyOracle(w, n)
    pol = make_policy("medium", n)
    rs = np.random.RandomState(n)
    st0 = rs.randint(0, 256, (n, 4, 42, 42)).astype(np.uint8)
    pol.set_stack(st0)
    ora.stack = st0.copy()
    both = torch.zeros((n, 2, 1, 42, 42), dtype=torch.uint8, device="cuda")  # the env's (N, 2, K, R, R) layout
    act = torch.full((n, 2), -7, dtype=torch.int32, device="cuda")
    for t in range(6):
        f = rs.randint(0, 256, (n, 42, 42)).astype(np.uint8)
        if t % 2:
            f = (f > 200).astype(np.uint8) * 255  # sparse frames like Pong's
        both[:, 1, 0] = torch.from_numpy(f).cuda()
        pol.act_device(both[:, 1], out=act[:, 1], want_logits=True)
        ao = ora(f[:, None])
        lg = pol.logits().cpu().numpy()
        assert np.abs(lg - ora.logits).max() < TOL, t
        srt = np.sort(ora.logits, 1)
        clear = (srt[:, 2] - srt[:, 1]) > 10 * TOL
        got = act.cpu().numpy()
        assert np.array_equal(got[clear, 1], ao.reshape(-1)[clear]), t
        assert np.all(got[:, 0] == -7)  # the other column of the action array is untouched
        assert np.array_equal(pol.get_stack().cpu().numpy(), ora.stack), t
    pol.close()


def test_tournament_with_cnn_opponent_matches_oracle_game(atlas):
    """cPongTournament-v0 with the MEDIUM opponent end to end on the device: the same game as the CPU
    oracle env.  The opponent's action must be the oracle policy's wherever its two best logits are
    clearly apart; in a near-tie (float32 summation order) the oracle env follows the device's choice,
    so that one flipped action does not turn into a different game."""
    _need_gpu()
    import competitive_rl_amd as crl
    from oracle import policy_oracle as P
    from oracle import pong_oracle as po

    n, T = 7, 300
    tour = crl.make_envs("cPongTournament-v0", num_envs=n, log_dir=None, seed=21)
    assert tour.get_agent_names() == ["RANDOM", "WEAK", "MEDIUM", "RULE_BASED"]
    tour.reset_opponent("MEDIUM")
    env = po.PongOracle(n, atlas, obs_mode=po.GRAY, resized_dim=42, frame_stack=1, seed=21)
    ora = P.PolicyOracle(P.load_weights(os.path.join(ROOT, "competitive_rl_amd", "assets", "pong_policy_medium.npz")), n)
    o_h = tour.reset()
    o_c = env.reset().copy()
    assert np.array_equal(o_h.cpu().numpy(), o_c[:, 0])
    rs = np.random.RandomState(5)
    ndone = nclear = 0
    for t in range(T):
        mine = rs.randint(0, 3, n)
        opp = ora(o_c[:, 1]).reshape(-1)
        o_h, r_h, d_h, _ = tour.step(mine)
        played = tour._act[:, 1].cpu().numpy()
        srt = np.sort(ora.logits, 1)
        clear = (srt[:, 2] - srt[:, 1]) > 10 * TOL
        assert np.array_equal(played[clear], opp[clear]), t
        nclear += int(clear.sum())
        o_c, r_c, d_c = env.step(np.stack([mine, played], 1))
        o_c = o_c.copy()
        assert np.array_equal(o_h.cpu().numpy(), o_c[:, 0]), t
        assert np.array_equal(r_h.cpu().numpy().reshape(-1), r_c[:, 0]), t
        assert np.array_equal(d_h.cpu().numpy().reshape(-1), d_c.astype(bool)), t
        ndone += int(d_c.sum())
    assert nclear > 0.9 * n * T
    # switching opponents keeps each policy's own stack (competitive_pong_env.py:28-34)
    tour.reset_opponent("WEAK")
    tour.step(rs.randint(0, 3, n))
    tour.reset_opponent("RANDOM")
    tour.step(rs.randint(0, 3, n))
    tour.close()
    env.close()


def test_tournament_step_device_equals_step():
    """The hot-loop entry (device buffers, no clones) plays the same game as the VecEnv-protocol step."""
    _need_gpu()
    import competitive_rl_amd as crl

    n = 11
    a = crl.make_envs("cPongTournament-v0", num_envs=n, log_dir=None, seed=3)
    b = crl.make_envs("cPongTournament-v0", num_envs=n, log_dir=None, seed=3)
    for t_ in (a, b):
        t_.reset_opponent("WEAK")
    oa, ob = a.reset(), b.reset()
    assert torch.equal(oa, ob)
    rs = np.random.RandomState(1)
    for t in range(120):
        act = rs.randint(0, 3, n)
        oa, ra, da, _ = a.step(act)
        buf, rew, done = b.step_device(torch.from_numpy(act.astype(np.int32)).cuda())
        assert torch.equal(oa, buf[:, 0]) and torch.equal(ra[:, 0], rew[:, 0]) and torch.equal(da[:, 0], done.bool()), t
    a.close()
    b.close()


def test_policy_abi_rejects_bad_arguments():
    """Error behaviour across the C ABI: return codes + crl_last_error, no crash."""
    _need_gpu()
    import ctypes as C

    import competitive_rl_amd._native as N

    pol = make_policy("weak", 3)
    L = N.load()
    frames = torch.zeros((3, 1, 42, 42), dtype=torch.uint8, device="cuda")
    acts = torch.zeros((3,), dtype=torch.int32, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.crl_policy_act(pol._h, C.c_void_p(frames.data_ptr()), 1763, C.c_void_p(acts.data_ptr()), 1, None, st) == -1
    assert b"frame_stride" in L.crl_last_error()
    assert L.crl_policy_act(pol._h, None, 1764, C.c_void_p(acts.data_ptr()), 1, None, st) == -1
    assert L.crl_policy_act(pol._h, C.c_void_p(frames.data_ptr()), 1764, C.c_void_p(acts.data_ptr()), 1, None, st) == 0
    h = C.c_void_p()
    assert L.crl_policy_create(0, 0, None, None, None, None, None, None, C.byref(h)) == -1
    assert L.crl_policy_create_full(0, 3, None, None, None, None, None, None, None, None, C.byref(h)) == -1
    with pytest.raises(ValueError):  # LightActorCritic tensors offered to the full-size network
        import competitive_rl_amd as crl
        import competitive_rl_amd.tournament as T
        crl.Policy(T.single_obs_space, T.single_act_space, 3, use_light_model=False, weights={**pol.weights, "conv3_w": pol.weights["conv2_w"],
                                                                                               "conv3_b": pol.weights["conv2_b"]})
    pol.close()
    pol.close()  # idempotent


@pytest.mark.parametrize("mode", ["0", "1"])
def test_mfma_and_hybrid_policy_kernels_match_the_oracle(mode):
    """The non-default opponent kernels -- packed-FMA (CRL_POLICY_MFMA=0) and fp32-MFMA conv1 (=1); the default, conv1 as three
    exact bf16 products per tap on the matrix pipe, is what every other test in this file runs -- give the same logits within
    1e-4 and the same actions as the numpy oracle, over ring wrap-around and ragged group sizes; a child process, the switch is
    read once."""
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    import subprocess
    import sys

    code = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from competitive_rl_amd.tournament import get_compute_action_function
from oracle import policy_oracle as P
root = %r
for n in (5000, 8, 13):
    pol = get_compute_action_function("MEDIUM", n)
    ora = P.PolicyOracle(P.load_weights(os.path.join(root, "competitive_rl_amd", "assets", "pong_policy_medium.npz")), n)
    rs = np.random.RandomState(n)
    for t in range(7):
        f = (rs.random_sample((n, 1, 42, 42)) > 0.8).astype(np.uint8) * rs.randint(1, 256, (n, 1, 1, 1)).astype(np.uint8)
        a = pol.act_device(torch.from_numpy(f).cuda(), want_logits=True).cpu().numpy()
        ao = ora(f).reshape(-1)
        assert np.abs(pol.logits().cpu().numpy() - ora.logits).max() < 1e-4, (n, t)
        srt = np.sort(ora.logits, 1)
        clear = (srt[:, 2] - srt[:, 1]) > 1e-3
        assert np.array_equal(a[clear], ao[clear]), (n, t)
    pol.close()
print("policy ok")
""" % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CRL_POLICY_MFMA=mode), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "policy ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.gpu
def test_compute_action_on_a_stacked_observation():
    """``Policy.compute_action(stack, deterministic)`` (reference utils/policy_serving.py:48-56): the network on a whole (N, 4, 42, 42) stack --
    greedy = what feeding the same four frames one by one ends with, logits = the oracle network's on that stack, sampled actions follow
    the softmax of those logits -- and the policy's own frame stack is what it was."""
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    from competitive_rl_amd.tournament import get_compute_action_function
    from oracle import policy_oracle as P

    for name, n in (("MEDIUM", 257), ("WEAK", 1)):
        pol = get_compute_action_function(name, n)
        ora = P.PolicyOracle(P.load_weights(os.path.join(ROOT, "competitive_rl_amd", "assets", "pong_policy_%s.npz" % name.lower())), n)
        rs = np.random.RandomState(3)
        frames = [(rs.random_sample((n, 1, 42, 42)) > 0.8).astype(np.uint8) * rs.randint(1, 256, (n, 1, 1, 1)).astype(np.uint8) for _ in range(6)]
        for f in frames[:2]:
            pol.act_device(torch.from_numpy(f).cuda())               # the policy's own history: must survive compute_action
            ora(f)
        own = pol.get_stack().clone()
        stack = np.concatenate(frames[2:], 1)                         # (n, 4, 42, 42): another situation altogether
        a = pol.compute_action(stack, deterministic=True)
        assert tuple(a.shape) == (n, 1) and a.dtype == torch.int64 and a.is_cuda
        assert torch.equal(pol.get_stack(), own)
        logits = pol.logits().cpu().numpy().copy()
        ref = get_compute_action_function(name, n)                    # the same four frames one by one through a fresh policy
        for f in frames[2:]:
            last = ref.act_device(torch.from_numpy(f).cuda(), want_logits=True).cpu().numpy().copy()
        assert np.abs(ref.logits().cpu().numpy() - logits).max() == 0.0 and np.array_equal(a.cpu().numpy().reshape(-1), last)
        # sampling: actions in {0, 1, 2}; over many draws the frequencies of env 0 follow softmax(logits[0])
        draws = torch.cat([pol.compute_action(torch.from_numpy(stack).float(), deterministic=False) for _ in range(200)], 1).cpu().numpy()
        assert draws.shape == (n, 200) and set(np.unique(draws)) <= {0, 1, 2}
        p = np.exp(logits[0] - logits[0].max())
        p /= p.sum()
        freq = np.bincount(draws[0], minlength=3) / 200.0
        assert np.abs(freq - p).max() < 0.15, (freq, p)
        assert torch.equal(pol.get_stack(), own)
        pol.close(), ref.close()
    with pytest.raises(ValueError):
        get_compute_action_function("WEAK", 2).compute_action(np.zeros((2, 1, 42, 42), np.uint8))
