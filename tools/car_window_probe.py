"""Where does the time of a SHORT timed window of CarRacing steps go?  K steps after a device synchronise: time until the
caller's stream is done vs until the whole device is (the context's low-priority walk-ahead stream included).
PYTHONPATH=. python tools/car_window_probe.py [K] [repeats]"""
import sys
import time

import torch

import competitive_rl_amd as crl

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 16384
env = crl.HipCarVecEnv(n, seed=0)
env.reset()
g = torch.Generator(device="cuda").manual_seed(3)
acts = torch.rand((16, n, 2, 2), generator=g, device="cuda") * 2 - 1
st = env.get_state()
st["elapsed"] = (torch.arange(n, dtype=torch.int64) * 1000 // n).numpy().astype(st["elapsed"].dtype)
env.set_state(st)
for i in range(1000):
    env.step_device(acts[i % 16])
torch.cuda.synchronize()
for r in range(R):
    for i in range(5):
        env.step_device(acts[i % 16])
    torch.cuda.synchronize()
    if "state" in sys.argv:
        env.get_state()  # (what bench.py does in front of its timed region, to count the resets inside)
    if "timing" in sys.argv:
        env.kernel_timing(True)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(K):
        env.step_device(acts[i % 16])
        evs[i + 1].record()
    t1 = time.perf_counter()
    torch.cuda.current_stream().synchronize()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("   per-step ms:", " ".join(f"{evs[i].elapsed_time(evs[i + 1]):.2f}" for i in range(K)))
    print(f"K={K}: host enqueue {1e3 * (t1 - t0):7.2f} ms | caller's stream done {1e3 * (t2 - t0):7.2f} ms ({1e3 * (t2 - t0) / K:.3f} per step) | "
          f"device done {1e3 * (t3 - t0):7.2f} ms ({1e3 * (t3 - t0) / K:.3f} per step)", flush=True)
env.close()
