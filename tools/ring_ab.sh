#!/bin/bash
# A/B of UVO_TUNE_PYR_RING (4: ROI + 4 pixels; 0: the whole border) on the bench workload, alternating, three rounds.
for i in 1 2 3; do
  for v in 4 0; do
    echo -n "ring=$v: "
    UVO_BENCH_RING=$v python3 bench.py --no-cpu-baseline --no-subrecords --no-verify --steps 40 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step_unoverlapped']['k_resize_level'])"
  done
done
