O=gpurun_out/r06_a; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; tail -3 $O/gputests.log
for i in 1 2; do
for br in 1 0; do
UVO_BENCH_BLUR_ROUNDING=$br python bench.py --no-cpu-baseline --no-subrecords --no-verify > $O/bench_br${br}_$i.json 2>$O/err.txt
python - <<PY
import json
d=json.loads(open("$O/bench_br${br}_$i.json").read().strip().splitlines()[-1])
print("BLUR_ROUNDING=$br run $i", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step_unoverlapped"], {k:v["live_ms"] for k,v in d["roofline"]["per_kernel"].items()})
PY
done
done
python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-600
