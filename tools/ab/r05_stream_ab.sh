for rep in 1 2 3; do
for wl in car car_fma; do
python bench.py --workload $wl --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl default stream', d['ms_per_step'])"
CRL_BENCH_STREAM=1 python bench.py --workload $wl --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl created stream', d['ms_per_step'])"
done
done
timeout 900 python -m pytest tests/test_hip_car_episodes.py tests/test_hip_car_parity.py -x -q -m gpu 2>&1 | tail -2
