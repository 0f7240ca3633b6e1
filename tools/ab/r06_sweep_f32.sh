#!/bin/bash
# Round 6: the address-linear gray writer with float32 output (tools/gray_variant.py sw pong_raster_gray.hip -DCRL_ABLATION).
# CRL_GRAY_SWEEP: 1 = the full writer (bit-exact), 2 / 3 / 4 = skeletons (64-byte headers | dense 8-byte records | dense 2-byte records).
cd ${GRAFT_REPO_ROOT:-.}
export CRL_LIB_VARIANT=sw
python - <<'PY'
import numpy as np, torch, os, sys
sys.path.insert(0, ".")
os.environ["CRL_GRAY_SWEEP"] = "1"
import competitive_rl_amd as crl
from competitive_rl_amd import _native
from oracle import pong_oracle as po
atlas = _native.load_score_atlas()
for K, n in ((4, 130), (1, 67)):
    env = crl.HipPongVecEnv(n, seed=5, mode="wrapped", resized_dim=84, frame_stack=K, obs_dtype="float32")
    ora = po.PongOracle(n, atlas, obs_mode=po.GRAY, resized_dim=84, frame_stack=K, seed=5)
    assert np.array_equal(torch.stack(env.reset(), 1).cpu().numpy(), ora.reset().astype(np.float32))
    rs = np.random.RandomState(K)
    for t in range(150):
        a = rs.randint(0, 3, (n, 2))
        obs, rew, done, _ = env.step(a)
        oo, orew, odone = ora.step(a)
        assert np.array_equal(torch.stack(obs, 1).cpu().numpy(), oo.astype(np.float32)), (K, t)
    env.close()
print("float32 sweep writer: bit-exact against the oracle")
PY
run() { python bench.py --workload $1 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_kernel_us'],1))"; }
echo "env kernel: fused84 $(CRL_GRAY_SWEEP=0 run fused84) us, fused84_f32 $(CRL_GRAY_SWEEP=0 run fused84_f32) us"
for mode in 2 3 4 1; do for nb in 1 2 4; do
  echo "CRL_GRAY_SWEEP=$mode NB=$nb: fused84 $(CRL_GRAY_SWEEP=$mode CRL_GRAY_SWEEP_NB=$nb run fused84) us, fused84_f32 $(CRL_GRAY_SWEEP=$mode CRL_GRAY_SWEEP_NB=$nb run fused84_f32) us"
done; done
