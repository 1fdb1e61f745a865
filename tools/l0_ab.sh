#!/bin/bash
# A/B of level 0 read in place (UVO_TUNE_LEVEL0_INPLACE) on the bench workload, alternating, three rounds.
mkdir -p gpurun_out
for i in 1 2 3; do
  for v in 1 0; do
    echo "L0=$v" >> gpurun_out/l0_ab.txt
    UVO_BENCH_L0=$v python3 bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/l0_ab.txt
  done
done
