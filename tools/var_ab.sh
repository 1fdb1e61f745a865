# bench A/B of the build variants under variants/libuvo_*.so, three rounds:  bash tools/var_ab.sh <tag> [bench args]
TAG=${1:-var_ab}; shift
O=gpurun_out/$TAG; mkdir -p $O
cp u-vip-slam_amd/libuvo.so /tmp/libuvo_plain.so
for i in 1 2 3; do
  for f in variants/libuvo_*.so; do
    v=$(basename $f .so | sed s/libuvo_//)
    cp $f u-vip-slam_amd/libuvo.so
    python bench.py --no-cpu-baseline --no-subrecords --no-verify "$@" 2>>$O/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v run $i', d['value'], d['ms_per_step'], 'alone', r['kernel_ms_per_step_unoverlapped'])" | tee -a $O/variants_ab.txt
  done
done
cp /tmp/libuvo_plain.so u-vip-slam_amd/libuvo.so
