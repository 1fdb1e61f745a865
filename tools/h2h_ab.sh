# A/B of the host-to-host leg over the build variants in variants/ (tools/variants.sh build ...):  bash tools/h2h_ab.sh [bench args]
cp u-vip-slam_amd/libuvo.so /tmp/libuvo_plain.so
for i in 1 2; do for f in variants/libuvo_*.so; do cp $f u-vip-slam_amd/libuvo.so; echo -n "$(basename $f .so): "; python bench.py --no-cpu-baseline --no-subrecords --h2h --no-verify "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['host_to_host']['value'], d['host_to_host']['ms_per_job'])"; done; done
cp /tmp/libuvo_plain.so u-vip-slam_amd/libuvo.so
