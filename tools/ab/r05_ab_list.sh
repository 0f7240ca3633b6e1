timeout 900 python -m pytest tests/test_hip_car_episodes.py tests/test_hip_car_parity.py -x -q -m gpu 2>&1 | tail -2
run() { lbl=$1; shift; env "$@" CRL_LIB_VARIANT=abl PYTHONPATH=. timeout 100 python tools/car_quick.py 16384 2000 500 2>&1 | grep "steps  1" | sed "s/^/$lbl: /"; }
for rep in 1 2 3; do
run "abl 4 list launches     " X=1
run "abl 5 list launches     " CRL_CAR_LIST_SPLIT=1
run "abl fma 4 list launches " QUICK_SOLVER=fma
run "abl fma 5 list launches " QUICK_SOLVER=fma CRL_CAR_LIST_SPLIT=1
done
