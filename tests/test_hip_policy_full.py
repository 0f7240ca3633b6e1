"""The full-size ActorCritic opponent on the device (crl_policy_create_full, csrc/pong_policy_full.hip) against the vectors recorded
from the reference's own torch module (tests/golden/policy_full.npz) and against the numpy oracle on ragged sizes.  float32:
logits within 1e-4, actions identical wherever the two best logits are further apart than 10 x that."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")


def make_policy(n, weights=None):
    from competitive_rl_amd import spaces
    from competitive_rl_amd.policy_serving import Policy

    return Policy(spaces.Box(0, 255, (1, 42, 42)), spaces.Discrete(3), n, use_light_model=False, weights=weights)


def test_full_network_matches_the_reference_module_vectors():
    _need_gpu()
    from tests.policy_full_weights import make_stacks, make_weights

    g = np.load(os.path.join(ROOT, "tests", "golden", "policy_full.npz"))
    w, x = make_weights(int(g["weight_seed"])), make_stacks(int(g["stack_seed"]), g["logits"].shape[0])
    n = x.shape[0]
    pol = make_policy(n, w)
    # the model sees the three newest planes of the stack + the pushed frame
    st0 = np.concatenate([np.full((n, 1, 42, 42), 77, np.uint8), x[:, :3]], axis=1)
    pol.set_stack(st0)
    a = pol.act_device(torch.from_numpy(x[:, 3:4]).cuda(), want_logits=True)
    lg = pol.logits().cpu().numpy()
    assert np.abs(lg - g["logits"]).max() < TOL
    assert np.array_equal(a.cpu().numpy(), g["logits"].argmax(1))
    assert np.array_equal(pol.get_stack().cpu().numpy(), x)
    pol.close()


@pytest.mark.parametrize("n", [1, 3, 17, 130, 257])
def test_full_network_vs_oracle_ragged_sizes(n):
    """conv2's tiles of 16 positions straddle envs (121 positions each), conv3's tiles hold 128 envs: the sizes cover one env, a
    ragged last tile in both, and a second workgroup row with one env; strided frame / action views; ring rotation over calls."""
    _need_gpu()
    from oracle import policy_oracle as P
    from tests.policy_full_weights import make_weights

    w = make_weights(100 + n)
    ora = P.PolicyOracle(w, n, full=True)
    pol = make_policy(n, w)
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
        assert (got[:, 0] == -7).all()
    assert np.array_equal(pol.get_stack().cpu().numpy(), ora.stack)
    pol.reset()
    assert int(pol.get_stack().max()) == 0
    pol.close()


def test_full_network_default_initialisation_and_call_protocol():
    """No checkpoint: the reference builds ActorCritic with orthogonal weights and zero biases (network.py:18-38); the served
    policy does the same and answers in the reference's protocol (numpy (N, 1) int64)."""
    _need_gpu()
    from oracle import policy_oracle as P

    n = 9
    pol = make_policy(n)
    w = pol.weights
    assert abs(float(np.linalg.norm(w["conv1_w"].reshape(16, -1)[0])) - 2.0 ** 0.5) < 1e-4 and not w["conv3_b"].any()
    f = np.random.RandomState(5).randint(0, 256, (n, 1, 42, 42)).astype(np.uint8)
    out = pol(f)
    assert out.shape == (n, 1) and out.dtype == np.int64
    ora = P.PolicyOracle({**w, "critic_w": np.zeros((1, 256), np.float32), "critic_b": np.zeros(1, np.float32)}, n, full=True)
    ora(f)
    pol.act_device(torch.from_numpy(f).cuda(), want_logits=True)
    ora(f)
    assert np.abs(pol.logits().cpu().numpy() - ora.logits).max() < TOL
    pol.close()


def test_tournament_against_a_full_size_opponent_matches_oracle_game(atlas):
    """A full-size ActorCritic added to the tournament pool (TournamentEnvWrapper.add_agent) plays the same game as the CPU
    oracle env with the numpy network on the right-hand paddle (near-ties follow the device's choice, as in
    test_hip_policy_parity.py)."""
    _need_gpu()
    import competitive_rl_amd as crl
    from oracle import policy_oracle as P
    from oracle import pong_oracle as po
    from tests.policy_full_weights import make_weights

    n, T = 5, 150
    tour = crl.make_envs("cPongTournament-v0", num_envs=n, log_dir=None, seed=33)
    w = make_weights(5)
    tour.add_agent("MINE", make_policy(n, w))
    assert tour.get_agent_names()[-1] == "MINE"
    tour.reset_opponent("MINE")
    env = po.PongOracle(n, atlas, obs_mode=po.GRAY, resized_dim=42, frame_stack=1, seed=33)
    ora = P.PolicyOracle(w, n, full=True)
    o_h = tour.reset()
    o_c = env.reset().copy()
    assert np.array_equal(o_h.cpu().numpy(), o_c[:, 0])
    rs = np.random.RandomState(6)
    nclear = 0
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
    assert nclear > 0.9 * n * T
    tour.close()
    env.close()


def test_full_network_more_envs_than_one_pass():
    """The activation scratch holds 65 536 envs; a larger policy runs in passes.  Envs on both sides of the boundary against the oracle."""
    _need_gpu()
    from oracle import policy_oracle as P
    from tests.policy_full_weights import make_weights

    n, probe = 65536 + 192, [0, 65535, 65536, 65536 + 191]
    w = make_weights(3)
    pol = make_policy(n, w)
    g = torch.Generator(device="cuda").manual_seed(9)
    ora = P.PolicyOracle(w, len(probe), full=True)
    for t in range(5):
        f = torch.randint(0, 256, (n, 1, 42, 42), dtype=torch.uint8, device="cuda", generator=g)
        a = pol.act_device(f, want_logits=True)
        ao = ora(f[probe].cpu().numpy())
        assert np.abs(pol.logits()[probe].cpu().numpy() - ora.logits).max() < TOL, t
        srt = np.sort(ora.logits, 1)
        clear = (srt[:, 2] - srt[:, 1]) > 10 * TOL
        assert np.array_equal(a[probe].cpu().numpy()[clear], ao.reshape(-1)[clear]), t
    pol.close()
