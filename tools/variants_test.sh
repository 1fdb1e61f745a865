#!/bin/bash
# On the box: run a pytest selection against every build variant under variants/ (see tools/variants.sh build).
cp u-vip-slam_amd/libuvo.so /tmp/libuvo_plain.so
for f in variants/libuvo_*.so; do
  cp $f u-vip-slam_amd/libuvo.so
  echo "$(basename $f .so | sed s/libuvo_//): $(timeout 900 python -m pytest "$@" -x -q 2>&1 | tail -1)"
done
cp /tmp/libuvo_plain.so u-vip-slam_amd/libuvo.so
