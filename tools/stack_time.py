"""Kernel time of the fused frame-stack draw (crl_step_stack) by hipEvents on the launch stream: 65 536 envs, 84 x 84, k = 4.
Usage: python tools/stack_time.py [f32|u8] [steps]   (CRL_LIB_VARIANT=abl CRL_GRAY_JPW=<jobs per wavefront> for the A/B)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import competitive_rl_amd as crl

kind = sys.argv[1] if len(sys.argv) > 1 else "f32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n = 65536
dev = torch.device("cuda")
env = crl.make_envs("cPongDouble-v0", num_envs=n, log_dir=None, seed=0, resized_dim=84, frame_stack=None, device=dev)
f = crl.FrameStackTensor(n, (1, 84, 84), 4, dev, dtype=torch.float32 if kind == "f32" else torch.uint8)
assert f.bind(env)
env.reset()
f.update_from_env(env)
gen = torch.Generator(device=dev).manual_seed(1)
pool = [torch.randint(0, 3, (n, 2), generator=gen, device=dev, dtype=torch.int32) for _ in range(16)]
for i in range(20):
    env.step(pool[i % 16])
    f.update_from_env(env)
torch.cuda.synchronize()
env.kernel_time_ms(1)
env.kernel_timing(True)
for i in range(steps):
    env.step(pool[i % 16])
    f.update_from_env(env)
torch.cuda.synchronize()
env.kernel_timing(False)
ms, cnt = env.kernel_time_ms(1)
assert f.fused_updates == steps + 21, f.fused_updates
stack_b = n * 4 * 7056 * (4 if kind == "f32" else 1)
obs_b = n * 7056 * (2 if kind == "f32" else 1)
print(f"{kind} stack, JPW={os.environ.get('CRL_GRAY_JPW', 'default')}: {ms / cnt * 1e3:.1f} us per launch over {cnt} launches; "
      f"{(stack_b + obs_b) / (ms / cnt * 1e-3) / 1e12:.2f} TB/s on {(stack_b + obs_b) / 1e9:.2f} GB")
