"""(The CRL_POLICY_* switches exist only in the profiling variant: run with CRL_LIB_VARIANT=abl after
`python -m competitive_rl_amd.build --variant abl -DCRL_ABLATION`.)
Time the opponent-policy kernel alone and the cPongTournament-v0 step around it.

    python tools/policy_bench.py [num_envs] [calls]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import competitive_rl_amd as crl
from competitive_rl_amd.tournament import get_compute_action_function

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 50
pol = get_compute_action_function("MEDIUM", n)
frames = (torch.rand((n, 1, 42, 42), device="cuda") > 0.9).to(torch.uint8) * 255
for _ in range(5):
    pol.act_device(frames)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(calls):
    pol.act_device(frames)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / calls
flop = n * 516800 * 2
print(f"policy kernel: {ms * 1e3:.1f} us per call at {n} envs = {flop / ms / 1e9:.2f} TFLOP/s fp32 "
      f"({flop / ms / 1e9 / 78.6 * 100:.1f} % of 78.6 TFLOP/s plain-FMA issue, {flop / ms / 1e9 / 157.3 * 100:.1f} % of packed peak)")

if os.environ.get("CRL_POLICY_DEBUG", "0") == "4":
    pol.act_device(frames, want_logits=True)
    torch.cuda.synchronize()
    t = pol.logits().reshape(-1)[:512 * 6].reshape(512, 6).double().mean(0) / (n / 5 / 512)
    names = ["staging", "write-back", "gather", "conv", "actor", "epilogue"]
    print("cycles per group (first wavefront, mean over workgroups):", {k: int(v) for k, v in zip(names, t.tolist())}, "sum", int(t.sum()))
    hw = pol.logits().reshape(-1).view(torch.int32)[4096:4096 + 512 * 8].reshape(512, 4, 2).cpu().numpy()
    import collections
    def dec(h, x):
        return (int(x) & 15, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15, (h >> 4) & 3)  # xcc, se, sh, cu, simd
    place = collections.defaultdict(list)
    for b in range(512):
        for w in range(4):
            xcc, se, sh, cu, simd = dec(int(hw[b, w, 0]), hw[b, w, 1])
            place[(xcc, se, sh, cu)].append((b, w, simd))
    print("CUs used:", len(place), "; first CUs:")
    for k in sorted(place)[:4]:
        print("  ", k, sorted(place[k]))
    tl = pol.logits().reshape(-1).view(torch.int64)[8192:8192 + 128].reshape(2, 16, 4).cpu().numpy()
    t0 = tl[tl > 0].min()
    for g in range(10):
        print("group", g, " WG48 conv pass0 [%d,%d] pass1 [%d,%d]   WG304 pass0 [%d,%d] pass1 [%d,%d]" % tuple(int(v - t0) for v in list(tl[0, g]) + list(tl[1, g])))
    sys.exit(0)
tour = crl.make_envs("cPongTournament-v0", num_envs=n, log_dir=None, seed=1)
for name in ("RULE_BASED", "MEDIUM"):
    tour.reset_opponent(name)
    tour.reset()
    act = torch.randint(0, 3, (calls + 10, n), device="cuda", dtype=torch.int32)
    for t in range(10):
        tour.step(act[t])
    torch.cuda.synchronize()
    a.record()
    for t in range(calls):
        tour.step(act[10 + t])
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / calls
    print(f"cPongTournament-v0 vs {name}: {ms * 1e3:.1f} us per step, {n / ms / 1e3:.2f} M env-steps/s")
tour.close()
