for us in 0 50 100 200 400; do
  echo -n "delay $us us: "; UVO_BENCH_DELAY_US=$us python bench.py --no-cpu-baseline --no-subrecords --no-verify --steps 60 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"
done
