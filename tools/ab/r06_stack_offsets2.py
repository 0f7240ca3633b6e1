"""Round 6 experiment, part 2: the float32 frame-stack draw alternates between ~1 390 and ~1 500 us with the PAIRING of observation buffer and stack
buffer written by one launch (tools/ab/r06_stack_offsets.py: the env flips its two observation buffers and the stack its two buffers together, so
a run keeps one pairing; 71 flips later the next run has the other).  Here the observation buffers are placed at chosen byte offsets from the
stack buffers (one arena) and each placement is timed in both pairings."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import competitive_rl_amd as crl

n, R, k = 65536, 84, 4
dev = torch.device("cuda", 0)
env = crl.make_envs("cPongDouble-v0", num_envs=n, log_dir=None, seed=0, resized_dim=R, frame_stack=None, device=dev)
elems = n * k * R * R                      # floats per stack buffer
obs_bytes = n * 2 * R * R
arena = torch.zeros((2 * elems * 4 + 4 * obs_bytes + (256 << 20)), dtype=torch.uint8, device=dev)
base = (-arena.data_ptr()) % (2 << 20)     # 2 MB-aligned origin inside the arena
gen = torch.Generator(device=dev).manual_seed(1)
pool = [torch.randint(0, 3, (n, 2), generator=gen, device=dev, dtype=torch.int32) for _ in range(16)]


def fview(off):
    return arena[base + off: base + off + elems * 4].view(torch.float32).view(n, k, R, R)


def oview(off):
    return arena[base + off: base + off + obs_bytes].view(n, 2, 1, R, R)


def measure(stack_offs, obs_offs, steps=40):
    f = crl.FrameStackTensor(n, (1, R, R), k, dev)
    f.current_obs, f._spare = fview(stack_offs[0]), fview(stack_offs[1])
    f.current_obs.zero_()
    env._obs = [oview(obs_offs[0]), oview(obs_offs[1])]
    env._flip = 0
    assert f.bind(env)
    env.reset()
    f.update_from_env(env)
    for i in range(9):    # (an odd number of flips in all: reset + 9 = 10 -> even; keep the pairing defined: after this, obs[0] pairs with the buffer that was `current_obs`)
        env.step(pool[i % 16]); f.update_from_env(env)
    torch.cuda.synchronize(); env.kernel_time_ms(1); env.kernel_timing(True)
    for i in range(steps):
        env.step(pool[i % 16]); f.update_from_env(env)
    torch.cuda.synchronize(); env.kernel_timing(False)
    ms, cnt = env.kernel_time_ms(1)
    f.unbind()
    return ms / cnt * 1e3


S = elems * 4
SA, SB = 0, (S + (2 << 20) - 1) // (2 << 20) * (2 << 20)           # stack buffers: 2 MB aligned
O0 = SB + (S + (2 << 20) - 1) // (2 << 20) * (2 << 20)               # first observation buffer: 2 MB aligned behind them
print(f"stack buffers at +0 and +{SB >> 20} MB, observation buffers from +{O0 >> 20} MB; obs buffer {obs_bytes >> 20} MB")
for d in (0, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, (1 << 20) + 4096):
    oa, ob = O0 + d, O0 + d + (obs_bytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    t1 = measure((SA, SB), (oa, ob))
    t2 = measure((SA, SB), (ob, oa))     # the other pairing
    print(f"obs offset +{d:>8d} B: pairing 1 {t1:7.1f} us | pairing 2 {t2:7.1f} us")
# the same with the stack buffers' distance changed instead
for d in (4096, 65536, 1 << 20):
    t1 = measure((SA, SB + d), (O0 + (8 << 20), O0 + (8 << 20) + (obs_bytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)))
    print(f"second stack buffer +{d:>8d} B further: {t1:7.1f} us")
