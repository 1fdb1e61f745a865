# one gpurun call that refreshes everything a round commits under profiles/: GPU tests, both bench configs, latency, rocprof + PMC.
#   bash tools/round_checkpoint.sh <tag>
TAG=${1:-r02_b}
O=gpurun_out/$TAG
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; tail -3 $O/gputests.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-400
python bench.py --config 3 > $O/bench_config3.json 2> $O/bench3.err; tail -1 $O/bench_config3.json | cut -c1-300
python tools/latency.py > $O/latency_batch1.json 2>/dev/null; tail -1 $O/latency_batch1.json | cut -c1-300
bash tools/profile_round.sh $TAG 2
bash tools/profile_round.sh ${TAG}_hd 3
