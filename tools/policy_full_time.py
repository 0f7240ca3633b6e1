"""Time the full-size ActorCritic opponent (crl_policy_create_full): ms per call at N envs.  python tools/policy_full_time.py [N] [calls]"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from competitive_rl_amd import spaces  # noqa: E402
from competitive_rl_amd.policy_serving import Policy  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pol = Policy(spaces.Box(0, 255, (1, 42, 42)), spaces.Discrete(3), n, use_light_model=False)
frames = [torch.randint(0, 256, (n, 1, 42, 42), dtype=torch.uint8, device="cuda") for _ in range(4)]
for i in range(4):
    pol.act_device(frames[i])
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(calls):
    pol.act_device(frames[i & 3])
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / calls
print(f"policy_full n={n}: {ms:.3f} ms / call, {n / ms / 1e3:.2f} M decisions/s, {4.79e6 * n / ms / 1e9:.1f} TFLOP/s fp32 (peak 157)")
