python -m pytest tests/test_hip_car_parity.py -x -q -m gpu 2>&1 | tail -2
CAR_STEPS=60 CAR_WARMUP=80 bash tools/car_timeline.sh 2>&1 | tail -3
python tools/_summ.py
for i in 1 2; do python bench.py --workload car --steps 300 --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-135; done
python bench.py --workload car --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-135
