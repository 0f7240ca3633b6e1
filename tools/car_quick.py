"""Quick CarRacing step timing at full size, for A/B runs under a short `timeout` (no pre-roll, prints as it goes).
PYTHONPATH=. python tools/car_quick.py [envs] [steps]"""
import sys
import time

import torch

import os

import competitive_rl_amd as crl

n, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, int(sys.argv[2]) if len(sys.argv) > 2 else 300
K = int(sys.argv[3]) if len(sys.argv) > 3 else 50  # steps between host synchronisations
early_stream = torch.cuda.Stream() if os.environ.get("QUICK_STREAM") == "early" else None  # created BEFORE the env's own streams
env = crl.HipCarVecEnv(n, seed=int(os.environ.get("QUICK_SEED", "0")), solver=os.environ.get("QUICK_SOLVER", "box2d"))
st = None
env.reset()
g = torch.Generator(device="cuda").manual_seed(3)
NA = int(os.environ.get("QUICK_ACTIONS", "16"))  # distinct action tensors, cycled (bench.py: 16 -- a car then repeats a 0.32 s pattern,
# drives in circles and meets its partner far more often than under fresh random actions: 7 % touching envs instead of ~3 %)
acts = torch.rand((NA, n, 2, 2), generator=g, device="cuda") * 2 - 1
# steady-state mix: stagger the TimeLimit
s0 = env.get_state()
PERIOD = int(os.environ.get("QUICK_EPISODE", "1000"))  # < 1000: every env starts this close to the TimeLimit again after each reset
# is not possible (the TimeLimit is 1000); instead the stagger is compressed so that resets come in bursts: n / PERIOD per step for PERIOD steps
s0["elapsed"] = 1000 - PERIOD + (torch.arange(n, dtype=torch.int64) * PERIOD // n).numpy()
env.set_state(s0)
torch.cuda.synchronize()
user_stream = early_stream if early_stream is not None else (torch.cuda.Stream() if os.environ.get("QUICK_STREAM") else None)  # the caller works on a created stream instead of the legacy default stream
if user_stream is not None:
    torch.cuda.set_stream(user_stream)
for blk in range(steps // K):
    t0 = time.time()
    for t in range(K):
        env.step_device(acts[(blk * K + t) % NA])
    th = (time.time() - t0) / K * 1e3  # the host's share: enqueueing a step (the GPU runs behind)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / K * 1e3
    print(f"steps {blk * K:5d}-{blk * K + K - 1:5d}: {dt:8.3f} ms/step  (host enqueue {th:.3f})  cap_hits {env.cap_hits()}", flush=True)
env.close()
