python -m pytest tests/test_hip_round2.py -x -q -m gpu -k "descriptor or shards or action_containment" 2>&1 | tail -5
python -m pytest tests/test_hip_car_parity.py -x -q -m gpu -k "api_surface" 2>&1 | tail -3
