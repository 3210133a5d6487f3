# Round-4 profile run: the bench command un-profiled and under rocprofv3 --kernel-trace --stats, general GEMV and option symmetric (one gpurun call).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r04d_prof_stats gpurun_out/r04d_prof_stats_sym
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r04d_bench_unprofiled.json 2> gpurun_out/r04d_bench_unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04d_prof_stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r04d_bench_profiled.json 2> gpurun_out/r04d_bench_profiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04d_prof_stats_sym -- python3 bench.py --symmetric --steps 100 --warmup 10 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r04d_bench_profiled_sym.json 2> gpurun_out/r04d_bench_profiled_sym.err
python bench.py --symmetric --steps 100 --warmup 10 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r04d_bench_unprofiled_sym.json 2> gpurun_out/r04d_bench_unprofiled_sym.err
