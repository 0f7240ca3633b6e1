python -m pytest tests/ -x -q -m gpu -k "car" 2>&1 | tail -3
