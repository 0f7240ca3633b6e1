"""Per-step durations of the steady-state CarRacing bench loop (torch events around every step): percentiles and the outliers.
PYTHONPATH=. python tools/car_step_times.py [envs] [steps]"""
import sys

import numpy as np
import torch

import competitive_rl_amd as crl

n, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, int(sys.argv[2]) if len(sys.argv) > 2 else 3000
env = crl.HipCarVecEnv(n, seed=0)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1234)
pool = [torch.rand((n, 2, 2), generator=g, device="cuda") * 2 - 1 for _ in range(16)]
st = env.get_state()
st["elapsed"] = (np.arange(n, dtype=np.int64) * 1000 // n).astype(st["elapsed"].dtype)
env.set_state(st)
for i in range(1000):
    env.step_device(pool[i % 16])
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
ev[0].record()
for i in range(steps):
    env.step_device(pool[i % 16])
    ev[i + 1].record()
torch.cuda.synchronize()
d = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]) * 1e3
print("steps", steps, "mean %.1f us  median %.1f  p10 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (d.mean(), np.median(d), np.percentile(d, 10), np.percentile(d, 90), np.percentile(d, 99), d.max()))
big = np.nonzero(d > 2 * np.median(d))[0]
print("steps longer than twice the median:", len(big), "their share of the total time: %.1f %%" % (100 * d[big].sum() / d.sum()), "at", big[:20].tolist(), np.round(d[big][:20]).tolist())
print("mean without them: %.1f us" % d[d <= 2 * np.median(d)].mean())
env.close()
