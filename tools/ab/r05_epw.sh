run() { lbl=$1; shift; env "$@" PYTHONPATH=. timeout 100 python tools/car_quick.py 16384 2000 500 2>&1 | grep "steps  1" | sed "s/^/$lbl: /"; }
for rep in 1 2 3; do
run "EPW1 8     " X=1
run "EPW1 4     " CRL_LIB_VARIANT=e4
run "EPW1 16    " CRL_LIB_VARIANT=e16
run "fma EPW1 8 " QUICK_SOLVER=fma
run "fma EPW1 4 " QUICK_SOLVER=fma CRL_LIB_VARIANT=e4
run "fma EPW1 16" QUICK_SOLVER=fma CRL_LIB_VARIANT=e16
done
