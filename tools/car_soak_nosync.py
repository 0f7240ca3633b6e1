"""Race hunt for the pipelined CarRacing step WITHOUT host synchronisation inside the loop (tools/car_soak.py reads a checksum back every
step, which hides cross-stream races): the bench's steady-state workload (staggered TimeLimit, 16 cycled action tensors: 7 % of the envs
touch, 16 resets per step), signatures of every step accumulated on the device, three runs -- pipelined twice (determinism) and
CRL_CAR_NO_OVERLAP=1 (one stream, everything in place) -- must agree step for step, state at the end included.
    PYTHONPATH=. python tools/car_soak_nosync.py [envs] [steps] [box2d|fma]"""
import os
import sys

import torch

import competitive_rl_amd as crl

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
solver = sys.argv[3] if len(sys.argv) > 3 else "box2d"


def run():
    env = crl.HipCarVecEnv(n, seed=21, solver=solver)
    env.reset()
    st = env.get_state()
    st["elapsed"] = (torch.arange(n, dtype=torch.int64) * 1000 // n).numpy().astype(st["elapsed"].dtype)
    env.set_state(st)
    g = torch.Generator(device="cuda").manual_seed(4)
    acts = torch.rand((16, n, 2, 2), generator=g, device="cuda") * 2 - 1
    w = torch.arange(1, 96 * 96 + 1, device="cuda", dtype=torch.int64)
    sig = torch.zeros((steps, 3), dtype=torch.float64, device="cuda")
    for t in range(steps):
        obs, rew, done = env.step_device(acts[t % 16])
        sig[t, 0] = (obs.view(n, 2, -1).to(torch.int64) * w).sum().double()
        sig[t, 1] = rew.double().sum()
        sig[t, 2] = done.sum().double()
    torch.cuda.synchronize()
    st = env.get_state()
    caps = env.cap_hits()
    env.close()
    return sig.cpu(), st, caps


a, sa, ca = run()
b, sb, cb = run()
os.environ["CRL_CAR_NO_OVERLAP"] = "1"  # (read when the context is created)
c, sc, cc = run()
del os.environ["CRL_CAR_NO_OVERLAP"]
bad_ab = torch.nonzero((a != b).any(1)).reshape(-1)
bad_ac = torch.nonzero((a != c).any(1)).reshape(-1)
print(f"solver {solver}: {steps} steps x {n} envs, episodes ended {int(a[:, 2].sum())}, touching at the end {int((sa['n_contact'] > 0).sum())}, cap hits {ca}; "
      f"first step where two pipelined runs differ: {int(bad_ab[0]) if len(bad_ab) else None}; pipelined vs one-stream: {int(bad_ac[0]) if len(bad_ac) else None}")
assert not len(bad_ab) and not len(bad_ac)
assert sa.tobytes() == sb.tobytes() == sc.tobytes()
