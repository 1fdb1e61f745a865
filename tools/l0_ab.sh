#!/bin/bash
# A/B of level 0 read in place (UVO_TUNE_LEVEL0_INPLACE) on the bench workload, alternating, three rounds: value, step, unoverlapped kernel times.
for i in 1 2 3; do
  for v in 1 0; do
    echo -n "L0=$v: "
    UVO_BENCH_L0=$v python3 bench.py --no-cpu-baseline --no-subrecords --no-verify --steps 40 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step_unoverlapped'])"
  done
done
