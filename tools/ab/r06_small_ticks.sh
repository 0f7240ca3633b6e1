#!/bin/bash
# Round 6: phase cycle stamps of the gray env kernel at 42 x 42 and 84 x 84, K = 1 (profiling build)
cd ${GRAFT_REPO_ROOT:-.}
for R in 42 84; do
CRL_LIB_VARIANT=abl CRL_GRAY_DEBUG=128 python - <<PY 2>&1 | grep -v amdgpu
import torch, sys
sys.path.insert(0, '.')
import competitive_rl_amd as crl
n = 65536
env = crl.make_envs("cPongDouble-v0", num_envs=n, log_dir=None, seed=0, resized_dim=$R, frame_stack=None)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
pool = [torch.randint(0, 3, (n, 2), generator=g, device="cuda", dtype=torch.int32) for _ in range(16)]
for i in range(40): env.step_device(pool[i % 16])
torch.cuda.synchronize(); env.kernel_time_ms(1); env.kernel_timing(True)
for i in range(50): env.step_device(pool[i % 16])
torch.cuda.synchronize(); env.kernel_timing(False)
ms, cnt = env.kernel_time_ms(1)
print(f"R=$R K=1 draw (instrumented instance): {ms / cnt * 1e3:.1f} us")
env.close()
PY
done
