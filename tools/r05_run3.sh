for sv in box2d fma; do
CRL_LIB_VARIANT=abl CRL_CAR_STAMPS=1 QUICK_SOLVER=$sv PYTHONPATH=. timeout 100 python tools/car_quick.py 16384 1500 500 > gpurun_out/stamps_$sv.txt 2>&1
done
tail -n 12 gpurun_out/stamps_*.txt
