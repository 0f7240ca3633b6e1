# what the walk-ahead pieces cost a step (profiling build; CRL_CAR_ABL_NO_WALK=1: resets reuse the stored walk, no walk kernel runs -- wrong tracks)
run() { lbl=$1; shift; env "$@" CRL_LIB_VARIANT=abl PYTHONPATH=. timeout 100 python tools/car_quick.py 16384 2000 500 2>&1 | grep "steps  1" | sed "s/^/$lbl: /"; }
for rep in 1 2 3; do
run "walk-ahead on " X=1
run "walk-ahead off" CRL_CAR_ABL_NO_WALK=1
done
run "fma walk-ahead on " QUICK_SOLVER=fma
run "fma walk-ahead off" QUICK_SOLVER=fma CRL_CAR_ABL_NO_WALK=1
