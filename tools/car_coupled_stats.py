"""How many envs are coupled / touching, and how the manifold counts are distributed (sizes the touching-env solver)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import competitive_rl_amd as crl

n = int(os.environ.get("N", 16384))
env = crl.HipCarVecEnv(n, seed=0)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
hist = np.zeros(9, np.int64)
for t in range(1, 1501):
    a = torch.rand((n, 2, 2), generator=g, device="cuda") * 2 - 1
    env.step_device(a, render=False)
    if t % 50 == 0:
        st = env.get_state()
        nc = st["n_contact"]
        h = np.bincount(nc, minlength=9)
        if t >= 500:
            hist += h
        if t % 250 == 0:
            print(t, "coupled %.4f" % st["coupled"].mean(), "touching %.4f" % (nc > 0).mean(), "max nc", nc.max(), "hist", h[1:].tolist())
print("steady-state histogram of manifolds per touching env (nc = 1..8):", hist[1:].tolist())
