run() { lbl=$1; shift; env "$@" PYTHONPATH=. timeout 100 python tools/car_quick.py 16384 2000 500 2>&1 | grep "steps  1" | sed "s/^/$lbl: /"; }
for rep in 1 2 3 4; do
run "views: one-manifold in the solve" X=1
run "views: all behind the solve     " CRL_LIB_VARIANT=tv0
run "fma views: one-manifold in solve" QUICK_SOLVER=fma
run "fma views: all behind the solve " QUICK_SOLVER=fma CRL_LIB_VARIANT=tv0
done
