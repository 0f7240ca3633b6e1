"""Long wrapped-Pong parity run against the oracle (one-off soak, GPU box): many episodes, all score
pairs, time-outs rare.  PYTHONPATH=. python tools/pong_soak.py [envs] [steps]"""
import sys

import numpy as np
import torch

import competitive_rl_amd as crl
from competitive_rl_amd import _native
from oracle import pong_oracle as po

n, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 6000
atlas = _native.load_score_atlas()
rs = np.random.RandomState(123)
env = crl.HipPongVecEnv(n, seed=31, mode="wrapped", resized_dim=84, frame_stack=4)
ora = po.PongOracle(n, atlas, obs_mode=po.GRAY, seed=31, resized_dim=84, frame_stack=4)
a = torch.stack(env.reset(), 1).cpu().numpy()
assert np.array_equal(a, ora.reset())
dones = 0
for t in range(steps):
    acts = rs.randint(0, 3, (n, 2)).astype(np.int32)
    acts[rs.random_sample((n, 2)) < 0.3] = 999  # rule-based play on a third of the moves: longer rallies, more scores
    obs, rew, done, _ = env.step(acts)
    oo, orew, odone = ora.step(acts)  # (the oracle keeps the K-stack as pixels: it has to render every step)
    assert np.array_equal(rew.cpu().numpy(), orew), t
    assert np.array_equal(done[:, 0].cpu().numpy().astype(np.uint8), odone), t
    if t % 7 == 0:
        assert np.array_equal(torch.stack(obs, 1).cpu().numpy(), oo), t
    dones += int(odone.sum())
env.close()
ora.close()
print("steps", steps, "envs", n, "episodes ended", dones, "bit-exact")
