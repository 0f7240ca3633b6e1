run() { lbl=$1; shift; env "$@" PYTHONPATH=. timeout 100 python tools/car_quick.py 16384 2000 500 2>&1 | grep "steps  1" | sed "s/^/$lbl: /"; }
for rep in 1 2 3; do
run "near frames: list kernel  " X=1
run "near frames: view + thirds" CRL_CAR_NEAR_LONG=1
run "fma near: list kernel     " QUICK_SOLVER=fma
run "fma near: view + thirds   " QUICK_SOLVER=fma CRL_CAR_NEAR_LONG=1
done
CRL_CAR_NEAR_LONG=1 timeout 600 python -m pytest tests/test_hip_car_episodes.py tests/test_hip_car_parity.py -x -q -m gpu 2>&1 | tail -2
