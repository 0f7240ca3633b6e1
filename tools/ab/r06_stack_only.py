"""Round 6: the float32 frame stack drawn ALONE (crl_draw_stack without an observation tensor) against the fused launch (stack + uint8 observation)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import competitive_rl_amd as crl

n, R, k = 65536, 84, 4
dev = torch.device("cuda", 0)
env = crl.make_envs("cPongDouble-v0", num_envs=n, log_dir=None, seed=0, resized_dim=R, frame_stack=None, device=dev)
f = crl.FrameStackTensor(n, (1, R, R), k, dev)
assert f.bind(env)
env.reset(); f.update_from_env(env)
gen = torch.Generator(device=dev).manual_seed(1)
pool = [torch.randint(0, 3, (n, 2), generator=gen, device=dev, dtype=torch.int32) for _ in range(16)]
for i in range(20):
    env.step(pool[i % 16]); f.update_from_env(env)
bufs = [f.current_obs, f._other_buffer()]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    torch.cuda.synchronize(); e0.record()
    for i in range(50):
        env._draw_stack_into(f._stack_desc(bufs[i & 1], 4, False))
    e1.record(); torch.cuda.synchronize()
    alone = e0.elapsed_time(e1) / 50 * 1e3
    torch.cuda.synchronize(); env.kernel_time_ms(1); env.kernel_timing(True)
    for i in range(50):
        env.step(pool[i % 16]); f.update_from_env(env)
    torch.cuda.synchronize(); env.kernel_timing(False)
    ms, cnt = env.kernel_time_ms(1)
    print(f"stack alone {alone:7.1f} us ({n * k * R * R * 4 / alone / 1e6:.2f} TB/s) | stack + observation in one launch {ms / cnt * 1e3:7.1f} us")
