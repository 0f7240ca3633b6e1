timeout 200 python bench.py --workload protocol --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['legs_ms_per_step'])"
timeout 300 python -m pytest tests/test_step_envs_golden.py tests/test_hip_round2.py -x -q -m gpu -k "stack or step_envs or frame" 2>&1 | tail -2
