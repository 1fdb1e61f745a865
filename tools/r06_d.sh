O=gpurun_out/r06_d; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; tail -3 $O/gputests.log
show() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; h=d.get('host_to_host') or {}
print('$2', d['value'], d['ms_per_step'], 'alone', r['kernel_ms_per_step_unoverlapped'], 'live', {k:v['live_ms'] for k,v in r['per_kernel'].items()}, 'h2h', h.get('value'), h.get('ms_per_job'), h.get('link_GBps'), h.get('h2h_frac'))
"; }
cp u-vip-slam_amd/libuvo.so /tmp/libuvo_plain.so
for i in 1 2 3; do
  cp /tmp/libuvo_plain.so u-vip-slam_amd/libuvo.so
  python bench.py --no-cpu-baseline --no-subrecords --no-verify > $O/scalar_$i.json 2>>$O/err.txt; show $O/scalar_$i.json "colpass=scalar run $i" | tee -a $O/blur_colpass_ab.txt
  cp variants/libuvo_pk.so u-vip-slam_amd/libuvo.so
  python bench.py --no-cpu-baseline --no-subrecords --no-verify > $O/pk_$i.json 2>>$O/err.txt; show $O/pk_$i.json "colpass=pk run $i" | tee -a $O/blur_colpass_ab.txt
  (cd variants/r04_tree && python bench.py --no-cpu-baseline --no-subrecords --no-verify > ../../$O/r04_$i.json 2>>../../$O/err.txt); show $O/r04_$i.json "r04 tree run $i" | tee -a $O/blur_colpass_ab.txt
done
cp /tmp/libuvo_plain.so u-vip-slam_amd/libuvo.so
for i in 1 2; do
  (cd variants/r04_tree && python bench.py --no-cpu-baseline --no-subrecords --no-verify --h2h > ../../$O/r04_h2h_$i.json 2>>../../$O/err.txt); show $O/r04_h2h_$i.json "r04 tree h2h run $i" | tee -a $O/h2h_ab.txt
  for ch in 2 4; do
    UVO_BENCH_CHUNKS=$ch python bench.py --no-cpu-baseline --no-subrecords --no-verify --h2h > $O/h2h_${ch}_$i.json 2>>$O/err.txt; show $O/h2h_${ch}_$i.json "r06 CHUNKS=$ch run $i" | tee -a $O/h2h_ab.txt
  done
done
grep -v amdgpu.ids $O/err.txt | tail -5
