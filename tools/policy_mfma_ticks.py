#!/usr/bin/env python3
# (CRL_POLICY_* switches: profiling variant only, CRL_LIB_VARIANT=abl)
"""CRL_POLICY_MFMA_DEBUG=4: per-phase cycle counters of the fp32-MFMA opponent kernel (first 64 workgroups, all 8 wavefronts).
Columns per group: 0 ring write-back, 1 tile loop, 2 wait + barrier after it, 3 ticket request, 4 previous group's sums + outputs,
5 next group's staging requests."""
import os, sys
os.environ["CRL_POLICY_MFMA_DEBUG"] = "4"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from competitive_rl_amd.tournament import get_compute_action_function
n = 65536
W = int(os.environ.get("W", "16"))
pol = get_compute_action_function("MEDIUM", n)
f = (torch.rand((n, 1, 42, 42), device="cuda") > 0.85).to(torch.uint8) * 255
for _ in range(5):
    pol.act_device(f, want_logits=True)
torch.cuda.synchronize()
t = pol.logits().cpu().numpy().reshape(-1)[:64 * W * 8].reshape(-1, W, 8)[:64]
groups = 65536 / 8 / 256
print("mean cycles per group and phase (rows: wavefront 0..7):")
print(np.round(t.mean(0)[:, :7] / groups).astype(int))
print("total cycles per wavefront", np.round(t.mean(0)[:, 7]).astype(int), "groups per workgroup ~", groups)
