CFG=${1:-2}
for v in 1 0 1 0; do
  echo -n "config $CFG fuse_blur_tree $v: "; UVO_BENCH_FUSE=$v python bench.py --config $CFG --no-cpu-baseline --no-subrecords --no-verify 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], {k: v for k, v in j['roofline']['kernel_ms_per_step_unoverlapped'].items() if 'oct' in k or 'gauss' in k})"
done
