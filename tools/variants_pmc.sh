#!/bin/bash
# On the box: one PMC counter set for one kernel, for every build variant under variants/:  bash tools/variants_pmc.sh "SQ_INSTS_VALU SQ_INSTS_SALU" k_fast_score
cp u-vip-slam_amd/libuvo.so /tmp/libuvo_plain.so
for f in variants/libuvo_*.so; do
  cp $f u-vip-slam_amd/libuvo.so
  echo -n "$(basename $f .so | sed s/libuvo_//): "
  bash tools/pmc_one.sh "$@" | tail -1
done
cp /tmp/libuvo_plain.so u-vip-slam_amd/libuvo.so
