import torch, numpy as np
import competitive_rl_amd as crl
n = 4096
env = crl.HipCarVecEnv(n, seed=0)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
for t in range(1, 1501):
    a = torch.rand((n, 2, 2), generator=g, device="cuda") * 2 - 1
    env.step_device(a, render=False)
    if t in (1, 10, 50, 100, 200, 400, 800, 1200, 1500):
        st = env.get_state()
        print(t, "coupled %.3f" % st["coupled"].mean(), "touching %.3f" % (st["n_contact"] > 0).mean(), "contacts/env %.2f" % st["n_contact"].mean())
