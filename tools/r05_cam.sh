timeout 600 python -m pytest tests/test_f64_math.py -x -q -m gpu -s -k camera 2>&1 | tail -4
timeout 900 python -m pytest tests/test_hip_car_episodes.py tests/test_hip_car_parity.py tests/test_hip_full_size_sampled.py -x -q -m gpu 2>&1 | tail -2
run() { lbl=$1; shift; env "$@" CRL_LIB_VARIANT=abl PYTHONPATH=. timeout 100 python tools/car_quick.py 16384 2000 500 2>&1 | grep "steps  1" | sed "s/^/$lbl: /"; }
for rep in 1 2 3; do
run "abl camera: plain double first" X=1
run "abl camera: double-double     " CRL_CAR_CAMERA_EXACT=1
run "abl fma camera: plain first   " QUICK_SOLVER=fma
run "abl fma camera: double-double " QUICK_SOLVER=fma CRL_CAR_CAMERA_EXACT=1
done
