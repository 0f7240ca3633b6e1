"""CarRacing with the batch split over K contexts that step CONCURRENTLY on K caller streams (same global env ids: the union is the one
batch, as with ranks): does a step whose kernels are latency chains overlap with itself?
    PYTHONPATH=. python tools/car_multi_ctx.py <total envs> <K> [steps] [solver]
Steady state as in bench.py (TimeLimit counters staggered, 1 000 un-timed steps)."""
import sys
import time

import torch

import competitive_rl_amd as crl

total, K = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
solver = sys.argv[4] if len(sys.argv) > 4 else "box2d"
n = total // K
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1)
pool = [torch.rand((total, 2, 2), generator=gen, device=dev) * 2 - 1 for _ in range(16)]
envs = [crl.HipCarVecEnv(n, seed=0, device=dev, env_id_base=k * n, solver=solver) for k in range(K)]
streams = [torch.cuda.Stream(device=dev) for _ in range(K)] if K > 1 else [torch.cuda.current_stream(dev)]
for k, e in enumerate(envs):
    e.reset()
    st = e.get_state()
    st["elapsed"] = ((torch.arange(n, dtype=torch.int64) + k * n) * 1000 // total).numpy().astype(st["elapsed"].dtype)
    e.set_state(st)
torch.cuda.synchronize()
acts = [[pool[i][k * n:(k + 1) * n].contiguous() for k in range(K)] for i in range(16)]


def step(i):
    for k, e in enumerate(envs):
        with torch.cuda.stream(streams[k]):
            e.step_device(acts[i % 16][k])


for i in range(1000):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{total} envs as {K} x {n} ({solver}): {dt * 1e6:.1f} us per step, {total / dt / 1e6:.2f} M env-steps/s", flush=True)
for e in envs:
    e.close()
