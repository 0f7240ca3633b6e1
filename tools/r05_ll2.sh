timeout 900 python -m pytest tests/test_hip_car_episodes.py tests/test_hip_car_parity.py tests/test_hip_full_size_sampled.py -x -q -m gpu -k "car" 2>&1 | tail -2
run() { lbl=$1; shift; env "$@" CRL_LIB_VARIANT=abl PYTHONPATH=. timeout 100 python tools/car_quick.py 16384 2000 500 2>&1 | grep "steps  1" | sed "s/^/$lbl: /"; }
for rep in 1 2 3; do
run "abl views: one-manifold in the solve" X=1
run "abl views: all behind the solve     " CRL_CAR_TOUCH_VIEW=0
run "abl one list launch (rounds 3-4)    " CRL_CAR_TOUCH_VIEW=0 CRL_CAR_TOUCH_FRAMES_LIST=1
run "abl fma views: one-manifold in solve" QUICK_SOLVER=fma
run "abl fma views: all behind the solve " QUICK_SOLVER=fma CRL_CAR_TOUCH_VIEW=0
run "abl fma one list launch             " QUICK_SOLVER=fma CRL_CAR_TOUCH_VIEW=0 CRL_CAR_TOUCH_FRAMES_LIST=1
done
