#!/bin/bash
# A/B of the pyramid launch forms (per-level launches against k_pyr_tiles level groups) on the latency path and on the bench workload.
#   bash tools/pyr_tiles_ab.sh <tag> "<latency specs>" "<bench specs>"      spec = FORM=<n> | GROUPS=<first:txXty[w],...>
TAG=${1:-r05_pyr}
LAT=${2:-"FORM=1 FORM=0"}
BEN=${3:-"FORM=1 FORM=2"}   # UVO_TUNE_PYR_FORM accepts 0 (auto), 1 (per-level launches), 2 (k_pyr_tiles)
O=gpurun_out/$TAG
mkdir -p $O
for spec in $LAT; do
  env UVO_LAT_PYR_${spec} python tools/latency.py 2>>$O/latency_ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$spec', 'host', d['host_ms_median'], 'device', d['device_ms_median'], d['kernel_us'])" 2>&1 | tail -1 | tee -a $O/latency_ab.txt
done
for spec in $BEN; do
  env UVO_BENCH_PYR_${spec} python bench.py --steps 60 --no-cpu-baseline --no-subrecords --no-verify 2>>$O/bench_ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); pk=d['roofline']['per_kernel']; print('$spec', d['value'], d['ms_per_step'], {k:(v['live_ms'],v['alone_ms']) for k,v in pk.items() if 'pyr' in k or 'resize' in k})" 2>&1 | tail -1 | tee -a $O/bench_ab.txt
done
