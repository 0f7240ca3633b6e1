run() { lbl=$1; shift; env "$@" CRL_LIB_VARIANT=abl PYTHONPATH=. timeout 30 python tools/car_quick.py 16384 2000 500 2>&1 | grep "steps  1" | sed "s/^/$lbl: /"; }
for rep in 1 2 3; do
run "abl camera: in the solve " X=1
run "abl camera: own kernel   " CRL_CAR_CAMERA_KERNEL=1
run "abl fma camera: in solve " QUICK_SOLVER=fma
run "abl fma camera: kernel   " QUICK_SOLVER=fma CRL_CAR_CAMERA_KERNEL=1
done
