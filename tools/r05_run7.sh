(time python bench.py --steps 20 --warmup 5) > gpurun_out/r05_bench_all.json 2> gpurun_out/r05_bench_all.err
wc -c gpurun_out/r05_bench_all.json; tail -5 gpurun_out/r05_bench_all.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_bench_all.json").read().strip().splitlines()[-1])
print(d["configs_brief"]); print(d["configs"]["protocol"])
PY
timeout 600 python -m pytest "tests/test_hip_car_parity.py::test_car_step_device_draws_into_the_callers_tensor" tests/test_hip_round2.py -x -q -m gpu 2>&1 | tail -3
