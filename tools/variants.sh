# A/B of build variants on the GPU box.  Here:   bash tools/variants.sh build name1="-DX=1" name2="-DX=2" ...   (libuvo.so ends as the plain build)
# On the box:  bash tools/variants.sh run [bench args]   -> one line per variant
if [ "$1" = build ]; then
  shift; mkdir -p variants; rm -f variants/*.so
  for kv in "$@"; do
    name=${kv%%=*}; flags=${kv#*=}
    UVO_EXTRA_FLAGS="$flags" python u-vip-slam_amd/build.py > /dev/null 2>&1 || { echo "build failed: $name"; exit 1; }
    cp u-vip-slam_amd/libuvo.so variants/libuvo_$name.so
  done
  UVO_EXTRA_FLAGS="" python u-vip-slam_amd/build.py > /dev/null 2>&1
else
  shift
  cp u-vip-slam_amd/libuvo.so /tmp/libuvo_plain.so
  for f in variants/libuvo_*.so; do
    cp $f u-vip-slam_amd/libuvo.so
    echo -n "$(basename $f .so | sed s/libuvo_//): "
    python bench.py --no-cpu-baseline --no-subrecords --no-verify "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms_per_step_unoverlapped'])"
  done
  cp /tmp/libuvo_plain.so u-vip-slam_amd/libuvo.so
fi
