"""The numpy restatement of the built-in CNN opponents (oracle/policy_oracle.py) against the vectors
recorded from the reference's own Policy / LightActorCritic and checkpoints
(tests/golden/gen_policy_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import policy_oracle as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "policy_light.npz")
TOL = 1e-4  # float32 logits; the reference's convolution does not define a summation order


def weights(name):
    return P.load_weights(os.path.join(ROOT, "competitive_rl_amd", "assets", "pong_policy_%s.npz" % name))


@pytest.mark.parametrize("name", ["weak", "medium"])
def test_network_on_noise_stacks(name):
    g = np.load(GOLD)
    lg, v = P.forward(weights(name), g["noise"])
    assert np.abs(lg - g[name + "_noise_logits"]).max() < TOL
    crit = np.load(os.path.join(ROOT, "competitive_rl_amd", "assets", "pong_policy_%s.npz" % name))
    assert crit["critic_w"].shape == (1, 1600)
    assert np.abs(v - g[name + "_noise_values"]).max() < TOL


@pytest.mark.parametrize("name", ["weak", "medium"])
def test_policy_closed_loop_trace(name):
    """Policy.__call__ semantics: own 4-frame stack, never cleared at episode ends, greedy action."""
    g = np.load(GOLD)
    frames, actions, logits = g[name + "_frames"], g[name + "_actions"], g[name + "_logits"]
    assert g[name + "_dones"].sum() > 0  # the trace crosses episode boundaries
    pol = P.PolicyOracle(weights(name), frames.shape[1])
    for t in range(frames.shape[0]):
        a = pol(frames[t][:, None])
        assert np.abs(pol.logits - logits[t]).max() < TOL, t
        assert np.array_equal(a.reshape(-1), actions[t]), t
    # the first call sees three zero planes and the reset frame
    assert np.array_equal(pol.stack[:, -1], frames[-1])


def test_shipped_names_match_reference_list():
    """builtin_policies.py:26-33 minus the two agents whose checkpoints the reference does not ship."""
    import competitive_rl_amd.tournament as T

    assert T.get_builtin_agent_names() == ["RANDOM", "WEAK", "MEDIUM", "RULE_BASED"]
    assert T.single_obs_space.shape == (1, 42, 42) and T.single_act_space.n == 3


def test_policy_has_no_cpu_fallback():
    """The product Policy is the HIP kernel or nothing: on a box without a GPU it refuses to construct."""
    import torch

    import competitive_rl_amd as crl
    import competitive_rl_amd.tournament as T

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        crl.Policy(T.single_obs_space, T.single_act_space, 4, crl.policy_serving.BUILTIN_CHECKPOINTS["WEAK"], use_light_model=True)


def test_shipped_weights_are_the_reference_checkpoints():
    """The npz assets hold the eight model tensors of the reference checkpoints unchanged (shape + a checksum that
    the golden logits depend on: forward() above would not match otherwise)."""
    for name in ("weak", "medium"):
        w = weights(name)
        assert w["conv1_w"].shape == (16, 4, 4, 4) and w["conv2_w"].shape == (16, 16, 2, 2) and w["actor_w"].shape == (3, 1600)
        assert all(np.isfinite(v).all() for v in w.values())
    assert not np.array_equal(weights("weak")["actor_w"], weights("medium")["actor_w"])


@pytest.mark.parametrize("name", ["weak", "medium"])
def test_reference_checkpoint_loader_matches_shipped_weights(name):
    """load_light_weights reads the reference's torch checkpoints (build container only: skipped where the reference
    tree is absent) and yields exactly the tensors shipped as .npz."""
    import sys

    from competitive_rl_amd.policy_serving import load_light_weights

    pkl = "/root/reference/resources/pong/checkpoint-%s.pkl" % name
    if not os.path.isfile(pkl):
        pytest.skip("reference tree not present")
    sys.dont_write_bytecode = True
    a, b = load_light_weights(pkl), weights(name)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_full_size_network_matches_the_reference_module():
    """ActorCritic (utils/network.py:14-70; STRONG / ALPHA_PONG's model, no checkpoint in the reference tree): the numpy restatement
    against logits and values the reference's own torch module produced on seeded weights (tests/golden/gen_policy_full_golden.py)."""
    from tests.policy_full_weights import make_stacks, make_weights

    g = np.load(os.path.join(ROOT, "tests", "golden", "policy_full.npz"))
    w, x = make_weights(int(g["weight_seed"])), make_stacks(int(g["stack_seed"]))
    assert int(g["feature_size"]) == 256
    lg, v = P.forward_full(w, x)
    assert lg.shape == (len(x), 3) and np.abs(lg - g["logits"]).max() < TOL, float(np.abs(lg - g["logits"]).max())
    assert np.abs(v - g["values"]).max() < TOL
    assert np.abs(g["logits"]).max() > 0.1  # (the vectors are not degenerate)
