UVO_LAT_TOPUP_PROFILE=1 python tools/latency.py 2>&1 | grep -v amdgpu | tail -2 | cut -c1-400
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
