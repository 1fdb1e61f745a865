O=gpurun_out/r06_c; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; tail -3 $O/gputests.log
show() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', d['value'], d['ms_per_step'], 'alone', r['kernel_ms_per_step_unoverlapped'], 'live', {k:v['live_ms'] for k,v in r['per_kernel'].items()}, d.get('step_spread'))
"; }
for i in 1 2; do
for st in 0 1 2 3 4 5 7; do
  UVO_BENCH_STAGGER=$st python bench.py --no-cpu-baseline --no-subrecords --no-verify > $O/b_st${st}_$i.json 2>>$O/err.txt; show $O/b_st${st}_$i.json "STAGGER=$st run $i" | tee -a $O/stagger_ab.txt
done
done
# the round-4 tree on the same box (its own bench.py and library): k_octree_gauss before the packed-fp32 column pass
for i in 1 2; do
  (cd variants/r04_tree && python bench.py --no-cpu-baseline --no-subrecords --no-verify > ../../$O/r04_$i.json 2>>../../$O/err.txt)
  python -c "
import json
d=json.loads(open('$O/r04_$i.json').read().strip().splitlines()[-1]); r=d['roofline']
print('r04 tree run $i', d['value'], d['ms_per_step'], 'alone', r['kernel_ms_per_step_unoverlapped'], 'live', {k:v['live_ms'] for k,v in r['per_kernel'].items()})" | tee -a $O/r04_vs_r06.txt
  python bench.py --no-cpu-baseline --no-subrecords --no-verify > $O/r06_$i.json 2>>$O/err.txt; show $O/r06_$i.json "r06 tree run $i" | tee -a $O/r04_vs_r06.txt
done
for ch in 2 4 8; do
  UVO_BENCH_CHUNKS=$ch python bench.py --no-cpu-baseline --no-subrecords --no-verify --h2h > $O/h2h_$ch.json 2>>$O/err.txt
  python -c "
import json
d=json.loads(open('$O/h2h_$ch.json').read().strip().splitlines()[-1]); h=d['host_to_host']
print('CHUNKS=$ch', h['value'], h['ms_per_job'], h.get('link_GBps'), h.get('h2h_frac'))" | tee -a $O/h2h_ab.txt
done
tail -5 $O/err.txt
