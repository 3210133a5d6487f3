# bench.py under the driver's launcher with ONE rank on the REAL RCCL (LAM_HIP_FORCE_RCCL=1: a 1-rank communicator): every collective call
# of every exchange, the experimental-exchange leg in a child of its own, rccl_version / rccl_ranks in the line
set -x
export TMPDIR=/tmp
LAM_HIP_FORCE_RCCL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29871 bench.py --gpus 1 --steps 50 --warmup 5 > gpurun_out/r05_torchrun_1rank_real_rccl.json 2> gpurun_out/r05_torchrun_1rank_real_rccl.err
python tools/rank_chain.py 23168 60 > gpurun_out/r05_rank_mode_chain.txt 2>&1
echo done
