# rocprofv3 trace of the streamed host-to-host leg for every build variant in variants/:  bash tools/h2h_trace.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp u-vip-slam_amd/libuvo.so /tmp/libuvo_plain.so
for f in variants/libuvo_*.so; do
  cp $f u-vip-slam_amd/libuvo.so
  n=$(basename $f .so)
  rm -rf /tmp/tr_$n
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr_$n -- python3 tools/h2h_trace.py 16 2>/dev/null | tail -1
  echo "== $n"; python3 tools/h2h_trace_summary.py /tmp/tr_$n
done
cp /tmp/libuvo_plain.so u-vip-slam_amd/libuvo.so
