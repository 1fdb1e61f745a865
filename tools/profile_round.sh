# per-round profile collection (run through gpurun): kernel-trace stats of the default bench command at ONE pipeline depth per file, then
# the PMC counters in separate passes (gpurun refuses --pmc combined with other trace domains).
#   bash tools/profile_round.sh <tag, e.g. r06_final> [config: 2 | 3]
# Writes gpurun_out/<tag>/kernel_stats_depth1.csv (every launch alone on the chip: UVO_PIPELINE_DEPTH=1) and kernel_stats_depth2.csv (the
# shipped two-lane pipeline: the other lane's kernels run beside each launch); bench.py's roofline.rocprof_csv names their copies in profiles/.
TAG=${1:-r06_a}
CFG=${2:-2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/$TAG
mkdir -p $O
SHORT="--config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-subrecords --no-verify"
export UVO_BENCH_SKIP_SERIAL=1   # no depth-1 pass inside a depth-2 run: every launch of a file ran at the file's depth
for D in 1 2; do
  export UVO_PIPELINE_DEPTH=$D
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt$D -o kt -- python3 bench.py --config $CFG --no-cpu-baseline --no-subrecords --no-verify > $O/bench_under_rocprof_depth$D.json 2> $O/kt$D.err
  find $O/kt$D -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_depth$D.csv \;
done
unset UVO_BENCH_SKIP_SERIAL
export UVO_PIPELINE_DEPTH=2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_a -- python3 bench.py $SHORT > /dev/null 2> $O/pmc_a.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_b -- python3 bench.py $SHORT > /dev/null 2> $O/pmc_b.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/$O/pmc_c -- python3 bench.py $SHORT > /dev/null 2> $O/pmc_c.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 --output-format csv -d $R/$O/pmc_d -- python3 bench.py $SHORT > /dev/null 2> $O/pmc_d.err
unset UVO_PIPELINE_DEPTH
# FETCH_SIZE / WRITE_SIZE calibration at the load widths the kernels use (once per tag)
if [ ! -f $O/fetch_calibration.json ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/cal_a -- $R/tools/ubench/stream_read > /dev/null 2> $O/cal_a.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/cal_b -- $R/tools/ubench/stream_read > /dev/null 2> $O/cal_b.err
  python3 tools/fetch_calibration.py $O/cal_a $O/cal_b > $O/fetch_calibration.json
fi
python3 tools/pmc_summary.py --config $CFG --calibration $O/fetch_calibration.json $O/pmc_a $O/pmc_b $O/pmc_c $O/pmc_d > $O/pmc.json
# keep the merge small
find $O -name "*.csv" -size +2M -delete
find $O -name "*kernel_trace.csv" -delete
head -8 $O/kernel_stats_depth1.csv
head -8 $O/kernel_stats_depth2.csv
tail -1 $O/bench_under_rocprof_depth2.json | cut -c1-300
