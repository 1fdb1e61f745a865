O=gpurun_out/r06_e; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; tail -3 $O/gputests.log
show() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', d['value'], d['ms_per_step'], 'alone', r['kernel_ms_per_step_unoverlapped'], 'live', {k:v['live_ms'] for k,v in r['per_kernel'].items()})
"; }
cp u-vip-slam_amd/libuvo.so /tmp/libuvo_plain.so
for i in 1 2 3; do
  for v in lb5scalar lb4scalar lb5pk grows128 flcap376; do
    cp variants/libuvo_$v.so u-vip-slam_amd/libuvo.so
    python bench.py --no-cpu-baseline --no-subrecords --no-verify > $O/${v}_$i.json 2>>$O/err.txt; show $O/${v}_$i.json "$v run $i" | tee -a $O/variants_ab.txt
  done
  (cd variants/r04_tree && python bench.py --no-cpu-baseline --no-subrecords --no-verify > ../../$O/r04_$i.json 2>>../../$O/err.txt); show $O/r04_$i.json "r04 tree run $i" | tee -a $O/variants_ab.txt
done
cp /tmp/libuvo_plain.so u-vip-slam_amd/libuvo.so
for i in 1 2; do
  for g in "+5:1x1" "+5:2x2" "+4:2x2" "+6:1x1" "+4:2x2r" "+5:1x1w"; do
    UVO_BENCH_PYR_FORM=2 UVO_BENCH_PYR_GROUPS="$g" python bench.py --no-cpu-baseline --no-subrecords --no-verify > $O/pyr_$i.json 2>>$O/err.txt; show $O/pyr_$i.json "PYR_GROUPS=$g run $i" | tee -a $O/pyr_hybrid_ab.txt
  done
  python bench.py --no-cpu-baseline --no-subrecords --no-verify > $O/pyr0_$i.json 2>>$O/err.txt; show $O/pyr0_$i.json "per-level launches run $i" | tee -a $O/pyr_hybrid_ab.txt
done
grep -v amdgpu.ids $O/err.txt | tail -5
