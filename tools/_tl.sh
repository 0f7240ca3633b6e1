python -m pytest tests/ -x -q -m gpu -k car 2>&1 | tail -2
python tools/_series.py 2>&1 | tail -9
