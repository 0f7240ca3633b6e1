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
for i in range(1000):
    env.step_device(acts[i % 16])
torch.cuda.synchronize()
_ = env.get_state()
torch.cuda.synchronize()
for rep in range(3):
  K = 120
  ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
  ev[0].record()
  t0 = time.perf_counter()
  host = []
  for k in range(K):
      env.step_device(acts[k % 16])
      ev[k + 1].record()
      host.append(time.perf_counter() - t0)
  torch.cuda.synchronize()
  d = [ev[k].elapsed_time(ev[k + 1]) for k in range(K)]
  print("spikes:", [(i, round(x, 1)) for i, x in enumerate(d) if x > 3.0])
  print("host enqueue cumulative ms at step 10/50/119:", [round(1e3 * host[i], 1) for i in (10, 50, 119)])
  print("mean first 25:", sum(d[:25]) / 25, "mean last 50:", sum(d[-50:]) / 50)
