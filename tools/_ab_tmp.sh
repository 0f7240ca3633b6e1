export PYTHONPATH=.
CRL_LIB_VARIANT=abl CRL_CAR_STAMPS=1 timeout 100 python tools/car_quick.py 16384 500 500 2>&1 | grep "touch class"
run() { echo "variant=$1"; shift; env "$@" timeout 100 python tools/car_quick.py 16384 2000 500 2>&1 | tail -2; }
for rep in 1 2 3; do
run new X=1
run old CRL_LIB_VARIANT=oldtouch
done
timeout 900 python -m pytest tests/test_hip_car_parity.py tests/test_hip_car_episodes.py -x -q 2>&1 | tail -3
