timeout 900 python -m pytest tests/test_hip_car_episodes.py tests/test_hip_car_parity.py tests/test_hip_full_size_sampled.py -x -q -m gpu -k "car" 2>&1 | tail -2
run() { lbl=$1; shift; env "$@" CRL_LIB_VARIANT=abl PYTHONPATH=. timeout 100 python tools/car_quick.py 16384 2000 500 2>&1 | grep "steps  1" | sed "s/^/$lbl: /"; }
for rep in 1 2 3; do
run "abl narrow: split     " X=1
run "abl narrow: all behind" CRL_CAR_NARROW_LATE=1
run "abl fma narrow: split " QUICK_SOLVER=fma
run "abl fma narrow: behind" QUICK_SOLVER=fma CRL_CAR_NARROW_LATE=1
done
for sv in box2d fma; do QUICK_SOLVER=$sv PYTHONPATH=. timeout 100 python tools/car_quick.py 16384 2000 500 2>&1 | grep "steps  1" | sed "s/^/ship $sv: /"; done
