# Round-4 evidence run (one gpurun call; outputs under gpurun_out/, copied to profiles/ by hand afterwards)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r04_bench_n1_builder.json 2> gpurun_out/r04_bench_n1_builder.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_n1_driver_args.json 2> gpurun_out/r04_bench_n1_driver_args.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_prof_stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r04_bench_profiled.json 2> gpurun_out/r04_bench_profiled.err
GPU_MAX_HW_QUEUES=8 LAM_HIP_DIRECT_SAME_DEVICE=1 LAM_BENCH_DEVICE_IDS=0,0 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r04_bench_one_process_2shards_one_device.json 2> gpurun_out/r04_bench_one_process_2shards.err
LAM_HIP_FORCE_RCCL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29871 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_torchrun_1rank_real_rccl.json 2> gpurun_out/r04_torchrun_1rank_real_rccl.err
python tools/variant_vs_size.py --variants 10,13,17 --cg 100 131072 98304 81920 73728 65536 61440 57344 49152 40000 32768 30000 20000 16384 10000 8192 > gpurun_out/r04_variant_vs_size.txt 2>&1
python tools/variant_vs_size.py --dtype f32 --variants 10,13,17 131072 98304 65536 40000 20000 10000 >> gpurun_out/r04_variant_vs_size.txt 2>&1
python tools/host_cpu_time.py 65536 32768 10000 4096 > gpurun_out/r04_host_cpu_time.txt 2>&1
python tools/thread_cpu.py 65536 > gpurun_out/r04_thread_cpu.txt 2>&1
LAM_HIP_LIB=$PWD/2024-eumaster4hpc-student-challenge_amd/liblam_hip_tuning.so python tools/host_enqueue_cost.py 4096 300 2 4 8 > gpurun_out/r04_host_enqueue_cost.txt 2>&1
mkdir -p /tmp/sw && python tools/sweep.py --grid file --files /tmp/sw --files-max-n 20000 --csv gpurun_out/r04_reference_file_grid.csv > gpurun_out/r04_reference_file_grid.txt 2>&1
