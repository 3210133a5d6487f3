# Round-5 soak of the rank mode on the stream-ordered RCCL double (one gpurun call): repeated solves TO CONVERGENCE on live rank
# contexts, the three exchanges in rotation, every solution's bits compared across ranks and with the first pass
export LD_PRELOAD=$PWD/tests/mock_rccl/libmock_rccl_async.so GPU_MAX_HW_QUEUES=20 MOCK_RCCL_TIMEOUT_MS=20000
for cfg in "4 4096 2400" "8 8192 600" "3 1001 1500" "6 5000 600" "4 4100 600 f32" "3 3001 600 bf16" "5 1001 600 f32"; do
  timeout -k 10 500 python tests/mock_rccl/soak_ranks.py $cfg 2>&1 | grep -v "^#  *[0-9]* s:" | cut -c1-300 | tail -3 || exit 1
done
