timeout 900 python -m pytest tests/test_hip_car_episodes.py tests/test_hip_car_parity.py -x -q -m gpu 2>&1 | tail -2
run() { lbl=$1; shift; env "$@" PYTHONPATH=. timeout 100 python tools/car_quick.py 16384 2000 500 2>&1 | grep "steps  1" | sed "s/^/$lbl: /"; }
for rep in 1 2 3; do
run "touch frames: view + thirds    " X=1
run "touch frames: one list launch  " CRL_CAR_TOUCH_FRAMES_LIST=1
run "fma touch frames: view + thirds" QUICK_SOLVER=fma
run "fma touch frames: one launch   " QUICK_SOLVER=fma CRL_CAR_TOUCH_FRAMES_LIST=1
done
