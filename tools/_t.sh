python tools/latency.py 2>/dev/null | tail -1 | cut -c1-600
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
