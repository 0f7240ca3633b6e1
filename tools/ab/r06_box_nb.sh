#!/bin/bash
# Round 6 A/B: the six boxes of a tile without control flow (python tools/gray_variant.py nb pong_raster_gray.hip -DCRL_GRAY_BOX_BRANCHY = the old form)
cd ${GRAFT_REPO_ROOT:-.}
krn() { CRL_LIB_VARIANT=$1 python - <<PY 2>&1 | grep -v amdgpu
import torch, sys
sys.path.insert(0, '.')
import competitive_rl_amd as crl
n = 65536
env = crl.HipPongVecEnv(n, seed=0, mode="wrapped", resized_dim=$2, frame_stack=$3)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
pool = [torch.randint(0, 3, (n, 2), generator=g, device="cuda", dtype=torch.int32) for _ in range(16)]
for i in range(20): env.step_device(pool[i % 16])
torch.cuda.synchronize(); env.kernel_time_ms(1); env.kernel_timing(True)
for i in range(200): env.step_device(pool[i % 16])
torch.cuda.synchronize(); env.kernel_timing(False)
ms, cnt = env.kernel_time_ms(1)
print(f"{ms / cnt * 1e3:.1f}")
PY
}
for rep in 1 2 3; do
  echo "R=42 K=1: shipped $(krn '' 42 1) us | branch-free boxes $(krn nb 42 1) us    R=84 K=1: $(krn '' 84 1) | $(krn nb 84 1)    R=84 K=4: $(krn '' 84 4) | $(krn nb 84 4)"
done
CRL_LIB_VARIANT=nb python -m pytest tests/test_hip_pong_parity.py tests/test_hip_round2.py tests/test_hip_stack_fused.py -m gpu -q -x 2>&1 | tail -2
