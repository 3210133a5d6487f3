# Round-5 profile run (one gpurun call): the default bench line, then the driver's bench command under rocprofv3 --kernel-trace --stats
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r05_prof_stats
python bench.py > gpurun_out/r05_bench_n1_builder.json 2> gpurun_out/r05_bench_n1_builder.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_n1_driver_args.json 2> gpurun_out/r05_bench_n1_driver_args.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_prof_stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r05_bench_profiled.json 2> gpurun_out/r05_bench_profiled.err
python tools/summarize_prof.py r05 gpurun_out/r05_prof_stats
find gpurun_out/r05_prof_stats -name "*kernel_trace.csv" -delete
echo done
