python bench.py --steps 40 --warmup 5 --workload fused84_f32_ref --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32_ref', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_kernel_us'])"
for d in 16 32 48; do
CRL_LIB_VARIANT=abl CRL_GRAY_DEBUG=$d python bench.py --steps 40 --warmup 5 --workload fused84_f32_ref --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32_ref debug=$d', d['ms_per_step'], d['roofline']['avg_kernel_us'])"
done
timeout 900 python -m pytest tests/test_hip_round2.py -x -q -m gpu -k "f32 or float32" 2>&1 | tail -2
