# kernel-trace timelines of the bench step: gaps between dependent launches, one lane alone (depth 1) and two lanes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-gaps}; mkdir -p $O
export UVO_BENCH_PYR=${2:-3,2,5}
for depth in 1 2; do
  export UVO_PIPELINE_DEPTH=$depth
  rm -rf /tmp/st$depth
  rocprofv3 --kernel-trace --output-format csv -d /tmp/st$depth -- python3 bench.py --no-cpu-baseline --no-subrecords --no-verify --steps 40 > $O/bench_d$depth.json 2> $O/err_d$depth.txt
  echo "== depth $depth"; python3 tools/step_trace_summary.py /tmp/st$depth | tee $O/summary_d$depth.txt
done
