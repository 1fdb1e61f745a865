O=gpurun_out/r06_f; mkdir -p $O
show() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', d['value'], d['ms_per_step'], 'alone', r['kernel_ms_per_step_unoverlapped'], 'live', {k:v['live_ms'] for k,v in r['per_kernel'].items()})
"; }
cp u-vip-slam_amd/libuvo.so /tmp/libuvo_plain.so
for i in 1 2 3; do
  for v in plain desc5 desc6 oct6 oct4; do
    cp variants/libuvo_$v.so u-vip-slam_amd/libuvo.so
    python bench.py --no-cpu-baseline --no-subrecords --no-verify > $O/${v}_$i.json 2>>$O/err.txt; show $O/${v}_$i.json "$v run $i" | tee -a $O/occupancy_ab.txt
  done
done
for v in plain desc5 desc6; do
  cp variants/libuvo_$v.so u-vip-slam_amd/libuvo.so
  python bench.py --config 3 --steps 20 --no-cpu-baseline --no-subrecords --no-verify > $O/${v}_hd.json 2>>$O/err.txt; show $O/${v}_hd.json "$v config 3" | tee -a $O/occupancy_ab.txt
done
cp /tmp/libuvo_plain.so u-vip-slam_amd/libuvo.so
grep -v amdgpu.ids $O/err.txt | tail -5
