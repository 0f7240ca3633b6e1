"""Race hunt for the multi-stream CarRacing step: two identical runs must produce identical rewards, dones and frame checksums
at every step, and so must a third run with CRL_CAR_NO_OVERLAP=1 (one stream, everything in place: the sequential reference of
the pipeline -- staged resets, walk-ahead timing and all).  Run on the GPU box: PYTHONPATH=. python tools/car_soak.py [envs] [steps]"""
import os
import sys

import torch

import competitive_rl_amd as crl

n, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 3000


def run():
    env = crl.HipCarVecEnv(n, seed=21)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(4)
    sig = []
    w = torch.arange(1, 96 * 96 + 1, device="cuda", dtype=torch.int64)
    for t in range(steps):
        a = torch.rand((n, 2, 2), generator=g, device="cuda") * 2 - 1
        if t % 3 == 0:
            a[:, :, 1] = a[:, :, 1].abs()
        obs, rew, done = env.step_device(a)
        cs = (obs.view(n, 2, -1).to(torch.int64) * w).sum()
        sig.append((int(cs.item()), float(rew.double().sum().item()), int(done.sum().item())))
    st = env.get_state()
    env.close()
    return sig, st


a, sa = run()
b, sb = run()
os.environ["CRL_CAR_NO_OVERLAP"] = "1"  # (read when the context is created)
c, sc = run()
del os.environ["CRL_CAR_NO_OVERLAP"]
bad = [t for t in range(steps) if a[t] != b[t] or a[t] != c[t]]
print("steps", steps, "envs", n, "episodes ended", sum(x[2] for x in a), "coupled at end", int(sa["coupled"].sum()),
      "first mismatching step", bad[0] if bad else None)
assert not bad
assert sa.tobytes() == sb.tobytes() and sa.tobytes() == sc.tobytes()
print("identical")
