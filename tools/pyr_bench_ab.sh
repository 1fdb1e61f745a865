for v in legacy 5,8,3 4,8,3 3,8,3 legacy 5,8,3; do
  echo -n "$v: "; UVO_BENCH_PYR=$v python bench.py --no-cpu-baseline --no-subrecords --no-verify --steps 80 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"
done
