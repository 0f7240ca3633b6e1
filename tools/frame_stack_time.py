#!/usr/bin/env python3
"""FrameStackTensor.update alone at the protocol leg's size: 65 536 envs, (1, 84, 84) u8 observation into a float32 (N, 4, 84, 84) stack."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import competitive_rl_amd as crl  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
fst = crl.FrameStackTensor(n, (1, 84, 84), 4, "cuda", **({"out_of_place": sys.argv[2] == "to"} if len(sys.argv) > 2 else {}))
obs = torch.randint(0, 256, (n, 1, 84, 84), dtype=torch.uint8, device="cuda")
mask = torch.ones((n, 1), device="cuda")
for _ in range(5):
    fst.update(obs, mask)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    fst.update(obs, mask)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
gb = n * 84 * 84 * (3 * 4 + 4 * 4 + 1) / 1e9
print(f"FrameStackTensor.update ({'out of place' if getattr(fst, 'out_of_place', False) else 'in place'}): {ms:.3f} ms, {gb:.2f} GB -> {gb / ms:.2f} TB/s = {gb / ms / 8.0:.2f} of HBM")
