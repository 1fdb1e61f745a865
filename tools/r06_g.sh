O=gpurun_out/r06_g; mkdir -p $O
cp u-vip-slam_amd/libuvo.so /tmp/libuvo_plain.so
cp variants/libuvo_trace.so u-vip-slam_amd/libuvo.so
SEQ=1 BATCH=1 python tools/oct_trace.py > $O/oct_trace_batch1.txt 2>$O/err.txt; cat $O/oct_trace_batch1.txt | head -80
cp /tmp/libuvo_plain.so u-vip-slam_amd/libuvo.so
tail -3 $O/err.txt
