run() { lbl=$1; shift; env "$@" PYTHONPATH=. timeout 100 python tools/car_quick.py 16384 2000 500 2>&1 | grep "steps  1" | sed "s/^/$lbl: /"; }
for rep in 1 2 3; do
run "frames on crit    " X=1
run "frames on one     " CRL_CAR_TOUCH_FRAMES=one
run "fma frames on crit" QUICK_SOLVER=fma
run "fma frames on one " QUICK_SOLVER=fma CRL_CAR_TOUCH_FRAMES=one
done
CRL_CAR_TOUCH_FRAMES=one timeout 600 python -m pytest tests/test_hip_car_episodes.py tests/test_hip_car_parity.py -x -q -m gpu -k "pipelined or staged or collide_ahead or sharding" 2>&1 | tail -2
