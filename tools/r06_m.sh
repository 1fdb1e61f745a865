O=gpurun_out/r06_m; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; tail -3 $O/gputests.log
timeout 900 python tools/soak_parity.py 60 11 > $O/soak_parity.log 2>&1; tail -1 $O/soak_parity.log
