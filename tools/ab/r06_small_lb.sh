#!/bin/bash
# Round 6 A/B: register budget of the R <= 45 gray instance (CRL_GRAY_SMALL_LB = workgroups per CU; variants by tools/gray_variant.py lb<N> ...)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do for v in "" lb3 lb4 lb5 lb8; do
  echo "variant='$v' $(CRL_LIB_VARIANT=$v python - <<'PY'
import torch, sys, os, time
sys.path.insert(0, '.')
import competitive_rl_amd as crl
n = 65536
env = crl.make_envs("cPongDouble-v0", num_envs=n, log_dir=None, seed=0, resized_dim=42, frame_stack=None)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
pool = [torch.randint(0, 3, (n, 2), generator=g, device="cuda", dtype=torch.int32) for _ in range(16)]
for i in range(20): env.step_device(pool[i % 16])
torch.cuda.synchronize(); env.kernel_time_ms(1); env.kernel_timing(True)
for i in range(200): env.step_device(pool[i % 16])
torch.cuda.synchronize(); env.kernel_timing(False)
ms, cnt = env.kernel_time_ms(1)
print(f"42x42 K=1 draw: {ms / cnt * 1e3:.1f} us")
PY
)"
done; done 2>&1 | grep -v amdgpu
