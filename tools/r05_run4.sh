# A/B: merged view kernel vs split (abl build has both); bulk stream at low priority
run() { # label, env...
  lbl=$1; shift
  env "$@" PYTHONPATH=. timeout 100 python tools/car_quick.py 16384 2000 500 2>&1 | grep "steps  1" | sed "s/^/$lbl: /"
}
for rep in 1 2; do
run "abl merged box2d" CRL_LIB_VARIANT=abl
run "abl split  box2d" CRL_LIB_VARIANT=abl CRL_CAR_VIEW_SPLIT=1
run "abl merged fma  " CRL_LIB_VARIANT=abl QUICK_SOLVER=fma
run "abl split  fma  " CRL_LIB_VARIANT=abl QUICK_SOLVER=fma CRL_CAR_VIEW_SPLIT=1
run "ship merged fma " QUICK_SOLVER=fma
run "ship lowbulk fma" QUICK_SOLVER=fma CRL_CAR_STREAM_ORDER=S2oDg
run "ship lowbulk box" CRL_CAR_STREAM_ORDER=S2oDg
run "ship merged box " X=1
done
timeout 300 python -m pytest tests/test_hip_car_parity.py -x -q -m gpu 2>&1 | tail -2
