set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r04_bench_n1_builder.json 2> gpurun_out/r04_bench_n1_builder.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_prof_stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r04_bench_profiled.json 2> gpurun_out/r04_bench_profiled.err
GPU_MAX_HW_QUEUES=8 LAM_HIP_DIRECT_SAME_DEVICE=1 LAM_BENCH_DEVICE_IDS=0,0 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r04_bench_one_process_2shards_one_device.json 2> gpurun_out/r04_bench_one_process_2shards.err
mkdir -p /tmp/sw && python tools/sweep.py --grid file --files /tmp/sw --files-max-n 20000 --csv gpurun_out/r04_reference_file_grid.csv > gpurun_out/r04_reference_file_grid.txt 2>&1
python tools/gemv_probe.py 131072 --dtype bf16 --variants 0,3,6 --rounds 3 --reps 10 > gpurun_out/r04_bf16_shapes.txt 2>&1
