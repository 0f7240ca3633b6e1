"""Long cPongTournament-v0 run against the CPU oracle game + numpy opponent (one-off soak, GPU box):
    PYTHONPATH=. python tools/tournament_soak.py [envs] [steps] [WEAK|MEDIUM]
The device opponent must pick the oracle policy's action wherever the two best logits are clearly apart; in a near-tie
the oracle game follows the device's choice (see tests/test_hip_policy_parity.py)."""
import os
import sys

import numpy as np

import competitive_rl_amd as crl
from competitive_rl_amd import _native
from oracle import policy_oracle as P
from oracle import pong_oracle as po

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
name = sys.argv[3] if len(sys.argv) > 3 else "MEDIUM"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tour = crl.make_envs("cPongTournament-v0", num_envs=n, log_dir=None, seed=77)
tour.reset_opponent(name)
env = po.PongOracle(n, _native.load_score_atlas(), obs_mode=po.GRAY, resized_dim=42, frame_stack=1, seed=77)
ora = P.PolicyOracle(P.load_weights(os.path.join(root, "competitive_rl_amd", "assets", "pong_policy_%s.npz" % name.lower())), n)
o_h, o_c = tour.reset(), env.reset().copy()
assert np.array_equal(o_h.cpu().numpy(), o_c[:, 0])
rs = np.random.RandomState(9)
near, episodes = 0, 0
for t in range(steps):
    mine = rs.randint(0, 3, n)
    opp = ora(o_c[:, 1]).reshape(-1)
    o_h, r_h, d_h, _ = tour.step(mine)
    played = tour._act[:, 1].cpu().numpy()
    srt = np.sort(ora.logits, 1)
    clear = (srt[:, 2] - srt[:, 1]) > 1e-3
    assert np.array_equal(played[clear], opp[clear]), t
    near += int((~clear).sum())
    o_c, r_c, d_c = env.step(np.stack([mine, played], 1))
    o_c = o_c.copy()
    assert np.array_equal(o_h.cpu().numpy(), o_c[:, 0]) and np.array_equal(r_h.cpu().numpy().reshape(-1), r_c[:, 0]), t
    assert np.array_equal(d_h.cpu().numpy().reshape(-1), d_c.astype(bool)), t
    episodes += int(d_c.sum())
tour.close()
env.close()
print(f"steps {steps} envs {n} opponent {name}: frames, rewards, dones identical; episodes ended {episodes}; "
      f"near-ties (top-2 logit gap < 1e-3) {near} of {n * steps}")
