#!/bin/bash
# kernel-trace timeline of the batch-1 path.  usage: bash tools/latency_trace.sh <tag> [ENV=VALUE ...]   (one run per extra argument; "" = defaults)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
[ $# -eq 0 ] && set -- "X=0"
for spec in "$@"; do
  rm -rf /tmp/lt
  env UVO_LAT_TRACE=1 $spec rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -- python3 tools/latency.py > $O/lat_$spec.json 2> $O/lat_err.txt
  echo "== $spec $(tail -1 $O/lat_$spec.json)"
  python3 tools/latency_trace.py /tmp/lt | tee $O/timeline_$spec.json | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: v for k, v in d.items() if k != 'timeline'})
for o in d['timeline']: print('   %-18s start %7.1f dur %6.1f gap %5.1f  %s' % (o['kernel'], o['start_us'], o['dur_us'], o['gap_before_us'], o['stream']))"
done
