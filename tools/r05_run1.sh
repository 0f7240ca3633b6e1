set -x
timeout 60 ./tools/solve_chain_probe > gpurun_out/probe.txt 2>&1
timeout 900 python -m pytest tests/test_hip_car_parity.py tests/test_hip_car_episodes.py -x -q -m gpu > gpurun_out/car_tests.log 2>&1; echo "rc=$?" >> gpurun_out/car_tests.log
for sv in box2d fma box2d fma; do QUICK_SOLVER=$sv PYTHONPATH=. timeout 100 python tools/car_quick.py 16384 2000 500 > gpurun_out/quick_$sv.$RANDOM.txt 2>&1; done
tail -3 gpurun_out/car_tests.log; cat gpurun_out/probe.txt; tail -n 4 gpurun_out/quick_*
