#!/bin/bash
# Round 6 A/B: float32 store epilogue batches / launch bounds (tools/gray_variant.py builds), fused84_f32 kernel us and the protocol legs.
cd ${GRAFT_REPO_ROOT:-.}
for v in "" e28l1 e7l1 e7l3 e7l4 e14l3; do
  a=$(CRL_LIB_VARIANT=$v python bench.py --workload fused84_f32 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_kernel_us'],1))")
  b=$(CRL_LIB_VARIANT=$v python bench.py --workload protocol --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); l=d['legs_ms_per_step']; print({k: round(v,3) for k,v in l.items() if k.startswith('step_envs')})")
  echo "variant='$v' fused84_f32 kernel $a us; protocol $b"
done
