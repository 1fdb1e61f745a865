# per-round profile collection (run through gpurun): kernel-trace stats, then the PMC counters in separate passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_f -o kt -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_f_bench.json 2> gpurun_out/prof_f.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_a -- python3 bench.py --steps 2 --warmup 1 > /dev/null 2> gpurun_out/pmc_a.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_b -- python3 bench.py --steps 2 --warmup 1 > /dev/null 2> gpurun_out/pmc_b.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_c -- python3 bench.py --steps 2 --warmup 1 > /dev/null 2> gpurun_out/pmc_c.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 --output-format csv -d $R/gpurun_out/pmc_d -- python3 bench.py --steps 2 --warmup 1 > /dev/null 2> gpurun_out/pmc_d.err
python3 tools/pmc_summary.py gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c gpurun_out/pmc_d > gpurun_out/pmc_summary.json
find gpurun_out/prof_f -name "*kernel_stats.csv" | head -2
ls gpurun_out/prof_f | head
# keep the merge small
find gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c gpurun_out/pmc_d -name "*.csv" -size +2M -delete
find gpurun_out/prof_f -name "*kernel_trace.csv" -size +8M -delete
tail -2 gpurun_out/prof_f_bench.json | cut -c1-200
