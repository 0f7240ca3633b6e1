"""Round 6 experiment: does the float32 frame-stack draw depend on WHERE its buffers lie?  (Back-to-back processes on one box read 1.44 /
1.63 / 1.53 ms for the same leg.)  One process, one env; the stack's two buffers are views into one arena at varying byte offsets, the
launch timed by hipEvents over 60 steps per placement; each placement measured twice, interleaved."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import competitive_rl_amd as crl

n, R, k = 65536, 84, 4
dev = torch.device("cuda", 0)
env = crl.make_envs("cPongDouble-v0", num_envs=n, log_dir=None, seed=0, resized_dim=R, frame_stack=None, device=dev)
elems = n * k * R * R
arena = torch.zeros(2 * elems + (64 << 20) // 4, dtype=torch.float32, device=dev)   # two buffers + 64 MB of slack
gen = torch.Generator(device=dev).manual_seed(1)
pool = [torch.randint(0, 3, (n, 2), generator=gen, device=dev, dtype=torch.int32) for _ in range(16)]


def measure(off_a, off_b):
    f = crl.FrameStackTensor(n, (1, R, R), k, dev)
    f.current_obs = arena[off_a // 4: off_a // 4 + elems].view(n, k, R, R)
    f._spare = arena[elems + off_b // 4: elems + off_b // 4 + elems].view(n, k, R, R)
    f.current_obs.zero_()
    assert f.bind(env)
    env.reset()
    f.update_from_env(env)
    for i in range(10):
        env.step(pool[i % 16]); f.update_from_env(env)
    torch.cuda.synchronize(); env.kernel_time_ms(1); env.kernel_timing(True)
    for i in range(60):
        env.step(pool[i % 16]); f.update_from_env(env)
    torch.cuda.synchronize(); env.kernel_timing(False)
    ms, cnt = env.kernel_time_ms(1)
    f.unbind()
    return ms / cnt * 1e3


placements = [(0, 0), (4096, 0), (65536, 0), (1 << 20, 0), (2 << 20, 0), (0, 1 << 20), (3 << 20, 5 << 20), (112896 // 2 // 16 * 16, 0), (16 << 20, 48 << 20)]
print("base addresses:", hex(arena.data_ptr()), "arena", arena.numel() * 4 >> 20, "MB")
for rep in range(2):
    for oa, ob in placements:
        print(f"rep {rep} offsets ({oa:>9d}, {ob:>9d}) B: {measure(oa, ob):8.1f} us")
