for d in 1 2 3; do
  echo -n "pipeline depth $d: "; UVO_PIPELINE_DEPTH=$d python bench.py --no-cpu-baseline --no-subrecords --no-verify --steps 60 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"
done
