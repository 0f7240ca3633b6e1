run() { lbl=$1; shift; env "$@" CRL_LIB_VARIANT=abl PYTHONPATH=. timeout 100 python tools/car_quick.py 16384 2000 500 2>&1 | grep "steps  1" | sed "s/^/$lbl: /"; }
for rep in 1 2 3; do
run "abl view: one-manifold islands " X=1
run "abl view: in the frame launch  " CRL_CAR_TOUCH_VIEW=0
run "abl fma view: one-manifold     " QUICK_SOLVER=fma
run "abl fma view: in the frames    " QUICK_SOLVER=fma CRL_CAR_TOUCH_VIEW=0
done
timeout 900 python -m pytest tests/test_hip_car_episodes.py tests/test_hip_car_parity.py -x -q -m gpu 2>&1 | tail -3
