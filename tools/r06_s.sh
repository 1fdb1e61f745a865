O=gpurun_out/r06_s; mkdir -p $O
python -m pytest tests/test_gpu_pyramid.py tests/test_gpu_parity.py -x -q -m gpu > $O/gputests.log 2>&1; tail -3 $O/gputests.log
bash tools/var_ab.sh r06_s
