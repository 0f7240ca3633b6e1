"""Step latency of the drop-in calls at small env counts (BASELINE config #1 is 4 envs): where the time of one ``envs.step`` goes when
the kernels are microseconds.  PYTHONPATH=. python tools/small_n_latency.py"""
import time

import numpy as np
import torch

import competitive_rl_amd as crl


def timed(fn, n=300, warm=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for N in (1, 4, 64, 1024):
    rows = []
    for kw in (dict(), dict(output="numpy", obs_dtype="float32")):
        envs = crl.make_envs("cPongDouble-v0", num_envs=N, frame_stack=None, log_dir=None, **kw)
        envs.reset()
        a_np = np.random.RandomState(0).randint(0, 3, (N, 2))
        a_dev = torch.as_tensor(a_np, dtype=torch.int32).cuda()
        rows.append(("step(numpy actions)" + (" -> numpy out" if kw else ""), timed(lambda: envs.step(a_np))))
        if not kw:
            rows.append(("step(device actions)", timed(lambda: envs.step(a_dev))))
            rows.append(("step_device", timed(lambda: envs.step_device(a_dev))))

            def sync_step():
                envs.step_device(a_dev)
                torch.cuda.synchronize()

            rows.append(("step_device + sync", timed(sync_step)))
        envs.close()
    car = crl.make_envs("cCarRacingDouble-v0", num_envs=N, frame_stack=4, log_dir=None)
    car.reset()
    ca = np.random.RandomState(1).uniform(-1, 1, (N, 2, 2)).astype(np.float32)
    cad = torch.as_tensor(ca).cuda()
    rows.append(("car step(numpy actions)", timed(lambda: car.step(ca), 200)))
    rows.append(("car step_device", timed(lambda: car.step_device(cad), 200)))
    car.close()
    print(f"N = {N}: " + "; ".join(f"{k} {v:.1f} us" for k, v in rows), flush=True)
