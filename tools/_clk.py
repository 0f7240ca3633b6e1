import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import competitive_rl_amd as crl
n = 16384
env = crl.HipCarVecEnv(n, seed=0)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = [torch.rand((n, 2, 2), generator=g, device="cuda") * 2 - 1 for _ in range(16)]
st = env.get_state()
st["elapsed"] = (np.arange(n) * 1000 // n).astype(st["elapsed"].dtype)
env.set_state(st)
t0 = time.time()
k = 0
while time.time() - t0 < 14:
    for _ in range(50):
        env.step_device(acts[k % 16]); k += 1
    torch.cuda.synchronize()
print("steps", k, "ms/step", 14e3 / k)
