python -m pytest tests/ -x -q -m gpu -k car 2>&1 | tail -2
python tools/_series.py 2>&1 | tail -3
CAR_STEPS=30 CAR_WARMUP=5 bash tools/car_timeline.sh 2>&1 | grep -E "step_k|narrow|touch_k|post" | tail -4
