"""Round 6 experiment, part 3: which variable makes the float32 frame-stack draw alternate between two times?  The env's own (torch-allocated)
observation buffers, the stack's buffers in an arena; the flip parity of the env and the starting buffer of the stack set explicitly."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import competitive_rl_amd as crl

n, R, k = 65536, 84, 4
dev = torch.device("cuda", 0)
env = crl.make_envs("cPongDouble-v0", num_envs=n, log_dir=None, seed=0, resized_dim=R, frame_stack=None, device=dev)
elems = n * k * R * R
arena = torch.zeros(2 * elems + (64 << 20) // 4, dtype=torch.float32, device=dev)
gen = torch.Generator(device=dev).manual_seed(1)
pool = [torch.randint(0, 3, (n, 2), generator=gen, device=dev, dtype=torch.int32) for _ in range(16)]
A, B = arena[:elems].view(n, k, R, R), arena[elems:2 * elems].view(n, k, R, R)
print("obs buffers", [hex(t.data_ptr()) for t in env._obs], "stack buffers", hex(A.data_ptr()), hex(B.data_ptr()))


def measure(flip, first, steps=40):
    f = crl.FrameStackTensor(n, (1, R, R), k, dev)
    f.current_obs, f._spare = (A, B) if first == 0 else (B, A)
    f.current_obs.zero_()
    env._flip = flip
    assert f.bind(env)
    env.reset()
    f.update_from_env(env)
    for i in range(9):
        env.step(pool[i % 16]); f.update_from_env(env)
    torch.cuda.synchronize(); env.kernel_time_ms(1); env.kernel_timing(True)
    for i in range(steps):
        env.step(pool[i % 16]); f.update_from_env(env)
    torch.cuda.synchronize(); env.kernel_timing(False)
    ms, cnt = env.kernel_time_ms(1)
    f.unbind()
    return ms / cnt * 1e3


for rep in range(3):
    print("rep", rep, " ".join(f"flip {fl} first {fi}: {measure(fl, fi):7.1f} us |" for fl in (0, 1) for fi in (0, 1)))
# and with nothing set explicitly: the flip parity carries over (41 flips per call: it alternates)
print("carry-over:", " ".join(f"{measure(env._flip, 0):7.1f}" for _ in range(6)))
