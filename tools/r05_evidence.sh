#!/bin/bash
# round-5 evidence, one gpurun call: margins tables, probes, the scaling grids and both bench topologies on ONE device
set -x
export TMPDIR=/tmp
O=gpurun_out
MOCK=$PWD/tests/mock_rccl/libmock_rccl_async.so
python tools/parity_margins.py --out $O/r05_parity_margins.txt > $O/r05_margins.log 2>&1
python tools/parity_margins.py --precision f32 --out $O/r05_parity_margins_f32.txt >> $O/r05_margins.log 2>&1
python tests/margins_bf16.py --out $O/r05_parity_margins_f32.txt --append >> $O/r05_margins.log 2>&1
python tools/small_n_probe.py 12000 10000 8192 > $O/r05_small_n_probe.txt 2>&1
python tools/upload_rate_padded.py > $O/r05_upload_rate_padded.txt 2>&1
python tools/sweep.py --grid scaling --scale 0.1 --launcher both --preload $MOCK --max-ranks 4 --csv $O/r05_scaling_grid_one_device.csv > $O/r05_scaling_grid_one_device.txt 2>&1
LD_PRELOAD=$MOCK LAM_BENCH_DEVICE_IDS=0,0 GPU_MAX_HW_QUEUES=8 LAM_HIP_DIRECT_SAME_DEVICE=1 MOCK_RCCL_TIMEOUT_MS=20000 python bench.py --gpus 2 --order 8192 --steps 50 --warmup 5 > $O/r05_bench_2gpus_one_device.json 2> $O/r05_bench_2gpus_one_device.err
LD_PRELOAD=$MOCK LAM_BENCH_DEVICE_IDS=0,0,0,0 GPU_MAX_HW_QUEUES=12 LAM_HIP_DIRECT_SAME_DEVICE=1 MOCK_RCCL_TIMEOUT_MS=20000 python bench.py --gpus 4 --order 32768 --steps 50 --warmup 5 > $O/r05_bench_4gpus_one_device.json 2> $O/r05_bench_4gpus_one_device.err
echo done
