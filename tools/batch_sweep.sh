#!/bin/bash
# On the box: unoverlapped per-kernel times (event pair per launch, pipeline depth 1) against the batch size -- intercept = ramp + tail + event
# overhead of a launch, slope = steady-state cost per frame.
for b in ${@:-32 64 128 256 512}; do
  echo -n "batch $b: "
  python bench.py --no-cpu-baseline --no-subrecords --no-verify --steps 10 --batch $b 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step_unoverlapped'])"
done
