# Round-4 (second session) evidence run: one gpurun call; outputs under gpurun_out/, copied to profiles/ by hand afterwards
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r04c_bench_n1_builder.json 2> gpurun_out/r04c_bench_n1_builder.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r04c_bench_n1_driver_args.json 2> gpurun_out/r04c_bench_n1_driver_args.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04c_prof_stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r04c_bench_profiled.json 2> gpurun_out/r04c_bench_profiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04c_prof_stats_sym -- python3 bench.py --symmetric --steps 100 --warmup 10 --no-cpu-baseline --no-also --no-traffic > gpurun_out/r04c_bench_profiled_sym.json 2> gpurun_out/r04c_bench_profiled_sym.err
find gpurun_out/r04c_prof_stats gpurun_out/r04c_prof_stats_sym -name "*kernel_trace.csv" -size +20M -delete
GPU_MAX_HW_QUEUES=8 LAM_HIP_DIRECT_SAME_DEVICE=1 LAM_BENCH_DEVICE_IDS=0,0 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r04c_bench_one_process_2shards_one_device.json 2> gpurun_out/r04c_bench_one_process_2shards.err
LAM_HIP_FORCE_RCCL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29871 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04c_torchrun_1rank_real_rccl.json 2> gpurun_out/r04c_torchrun_1rank_real_rccl.err
python tools/rank_chain.py > gpurun_out/r04c_rank_mode_chain.txt 2>&1
python tools/symmetric_probe.py f64 > gpurun_out/r04c_symmetric_probe.txt 2>&1
python tools/symmetric_probe.py f32 131072 65536 32768 >> gpurun_out/r04c_symmetric_probe.txt 2>&1
python tools/gemv_counters.py --cases f64:65536,f64:65536:sym,f64:32768:sym,f32:131072:sym --out gpurun_out/r04c_symv_counters.json > gpurun_out/r04c_symv_counters.txt 2>&1
