# latency A/B of two builds of the quad-tree + its per-phase trace (variants/libuvo_octold.so, _octnew.so, _trace.so):  bash tools/oct_ab.sh <tag>
O=gpurun_out/${1:-oct_ab}; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; tail -3 $O/gputests.log
cp u-vip-slam_amd/libuvo.so /tmp/libuvo_plain.so
for i in 1 2 3; do
for v in octold octnew; do
  cp variants/libuvo_$v.so u-vip-slam_amd/libuvo.so
  python tools/latency.py 2>>$O/lat.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'host', d['host_ms_median'], 'topup', d['topup_host_ms_median'], 'device', d['device_ms_median'], d['kernel_us'])" | tee -a $O/latency_oct_ab.txt
done
done
for v in octold octnew; do
  cp variants/libuvo_$v.so u-vip-slam_amd/libuvo.so
  for i in 1 2; do python bench.py --no-cpu-baseline --no-subrecords --no-verify | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', d['value'], r['kernel_ms_per_step_unoverlapped'])" | tee -a $O/latency_oct_ab.txt; done
done
cp variants/libuvo_trace.so u-vip-slam_amd/libuvo.so
SEQ=1 BATCH=1 python tools/oct_trace.py > $O/oct_trace_batch1.txt 2>$O/err.txt; tail -28 $O/oct_trace_batch1.txt
cp /tmp/libuvo_plain.so u-vip-slam_amd/libuvo.so
