# one gpurun call that refreshes everything a round commits under profiles/: GPU tests, both bench configs, latency, rocprof + PMC.
#   bash tools/round_checkpoint.sh <tag>
TAG=${1:-r02_b}
O=gpurun_out/$TAG
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; tail -3 $O/gputests.log
# counters first: the bench lines below quote the newest profiles/r*_pmc.json (traffic, instruction counts), which must be this build's
bash tools/profile_round.sh $TAG 2
bash tools/profile_round.sh ${TAG}_hd 3
cp $O/pmc.json profiles/${TAG}_pmc.json; cp gpurun_out/${TAG}_hd/pmc.json profiles/${TAG}_hd_pmc.json; cp $O/fetch_calibration.json profiles/${TAG}_fetch_calibration.json
python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-400
python bench.py --config 3 > $O/bench_config3.json 2> $O/bench3.err; tail -1 $O/bench_config3.json | cut -c1-300
python tools/latency.py > $O/latency_batch1.json 2>/dev/null; tail -1 $O/latency_batch1.json | cut -c1-300
# the per-call kernel timeline of the batch-1 path (rocprofv3 kernel trace: start / duration / gap of every launch)
bash tools/latency_trace.sh ${TAG}_lt "X=0" > $O/latency_timeline.txt 2>&1; cp gpurun_out/${TAG}_lt/timeline_X=0.json $O/latency_timeline.json; tail -8 $O/latency_timeline.txt
# two ranks on this one GPU (gloo rendezvous, both on device 0): the N>1 code path of bench.py end to end
UVO_BENCH_DRYRUN_ONE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 10 --warmup 2 > $O/bench_2ranks_one_gpu.json 2> $O/bench2.err; tail -1 $O/bench_2ranks_one_gpu.json | cut -c1-200
timeout 900 python tools/soak_parity.py 60 7 > $O/soak_parity.log 2>&1; tail -1 $O/soak_parity.log
timeout 600 python tools/soak_matcher.py > $O/soak_matcher.log 2>&1; tail -1 $O/soak_matcher.log
timeout 600 python tools/soak_misc.py > $O/soak_misc.log 2>&1; tail -1 $O/soak_misc.log
# a longer randomized parity soak of the final build (UVO_SOAK_LONG=n trials)
if [ -n "$UVO_SOAK_LONG" ]; then timeout 1500 python tools/soak_parity.py $UVO_SOAK_LONG 23 > $O/soak_parity_long.log 2>&1; tail -1 $O/soak_parity_long.log; fi
