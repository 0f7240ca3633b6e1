python -m pytest tests/test_hip_car_parity.py tests/test_hip_round2.py tests/test_hip_car_step_golden.py tests/test_car_wrappers_golden.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --workload car --steps 400 --warmup 5 --no-cpu-baseline"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"].get("resets_in_timed_region"), d["roofline"]["avg_kernel_us"])'
for k in 1 2 3; do echo "== default"; $B 2>/dev/null | tail -1 | python -c "$P"; done
echo "== from reset"; CRL_BENCH_CAR_PREROLL=0 $B 2>/dev/null | tail -1 | python -c "$P"
CAR_STEPS=100 bash tools/car_timeline.sh > gpurun_out/tl_bb1.txt 2>&1; python tools/car_timeline_summary.py > gpurun_out/tl_bb1_summ.txt
