# copy what a round commits from gpurun_out/<tag>{,_hd} (tools/round_checkpoint.sh) into profiles/:  bash tools/commit_profiles.sh r03_r
T=$1
for f in bench.json bench_config3.json bench_2ranks_one_gpu.json bench_under_rocprof_depth1.json bench_under_rocprof_depth2.json fetch_calibration.json gputests.log kernel_stats_depth1.csv kernel_stats_depth2.csv latency_batch1.json latency_timeline.json latency_timeline.txt pmc.json soak_parity.log soak_parity_long.log soak_matcher.log soak_misc.log; do
  [ -f gpurun_out/$T/$f ] && cp gpurun_out/$T/$f profiles/${T}_$f
done
for f in bench_under_rocprof_depth1.json bench_under_rocprof_depth2.json kernel_stats_depth1.csv kernel_stats_depth2.csv pmc.json; do
  [ -f gpurun_out/${T}_hd/$f ] && cp gpurun_out/${T}_hd/$f profiles/${T}_hd_$f
done
ls profiles | grep "^${T}_"
