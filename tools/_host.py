import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import competitive_rl_amd as crl
n = 16384
env = crl.HipCarVecEnv(n, seed=0)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.rand((64, n, 2, 2), generator=g, device="cuda") * 2 - 1
for w in range(100):
    env.step_device(acts[w % 64])
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter()
    for k in range(300):
        env.step_device(acts[k % 64])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"trial {trial}: enqueue {1e3*(t1-t0)/300:.3f} ms/step, total {1e3*(t2-t0)/300:.3f} ms/step -> {n*300/(t2-t0)/1e6:.2f} M env-steps/s")
