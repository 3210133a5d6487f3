# Round-5 profile run (one gpurun call): the default bench line, then the driver's bench command under rocprofv3 --kernel-trace --stats.
#   usage: bash tools/r05_profile.sh [tag]      (default tag r05; r05b = the final build of the round, after the shard limit went to 64)
set -x
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/${TAG}_prof_stats
python bench.py > gpurun_out/${TAG}_bench_n1_builder.json 2> gpurun_out/${TAG}_bench_n1_builder.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_n1_driver_args.json 2> gpurun_out/${TAG}_bench_n1_driver_args.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof_stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-also --no-traffic > gpurun_out/${TAG}_bench_profiled.json 2> gpurun_out/${TAG}_bench_profiled.err
python tools/summarize_prof.py ${TAG} gpurun_out/${TAG}_prof_stats
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_gemv_by_grid.csv gpurun_out/
find gpurun_out/${TAG}_prof_stats -name "*kernel_trace.csv" -delete
echo done
