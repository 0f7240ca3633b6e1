for i in 1 2 3; do
timeout 300 python -m pytest tests/test_hip_car_episodes.py tests/test_hip_car_parity.py tests/test_car_wrappers_golden.py -x -q -m gpu --timeout 100 > gpurun_out/dbg_$i.log 2>&1; echo "run $i rc=$?"; tail -3 gpurun_out/dbg_$i.log
done
