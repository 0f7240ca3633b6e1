timeout 900 python -m pytest tests/test_hip_car_episodes.py tests/test_hip_car_parity.py tests/test_hip_full_size_sampled.py tests/test_car_wrappers_golden.py -x -q -m gpu 2>&1 | tail -2
run() { lbl=$1; shift; env "$@" CRL_LIB_VARIANT=abl PYTHONPATH=. timeout 100 python tools/car_quick.py 16384 2000 500 2>&1 | grep "steps  1" | sed "s/^/$lbl: /"; }
for rep in 1 2 3; do
run "abl frames: thirds     " X=1
run "abl frames: whole tiles" CRL_CAR_OBS_WHOLE_TILES=1
run "abl fma frames: thirds " QUICK_SOLVER=fma
run "abl fma frames: whole  " QUICK_SOLVER=fma CRL_CAR_OBS_WHOLE_TILES=1
done
