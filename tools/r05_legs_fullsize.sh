export TMPDIR=/tmp
MOCK=$PWD/tests/mock_rccl/libmock_rccl_async.so
for g in 2 4; do
ids=$(python -c "print(','.join(['0']*$g))")
T0=$(date +%s.%N); env LD_PRELOAD=$MOCK LAM_BENCH_DEVICE_IDS=$ids GPU_MAX_HW_QUEUES=$((2*g+4)) LAM_HIP_DIRECT_SAME_DEVICE=1 MOCK_RCCL_TIMEOUT_MS=30000 python bench.py --gpus $g --steps 20 --warmup 5 > gpurun_out/r05_legs_full_$g.json 2> gpurun_out/r05_legs_full_$g.err
echo "wall: $(python -c "import time; print(round(time.time()-$T0,1))") s"
python - <<PY
import json
o=json.loads(open("gpurun_out/r05_legs_full_$g.json").read().strip().splitlines()[-1])
rm=o["rank_mode_rccl"]
print("gpus $g: value", o["value"], "exchange_us", o["exchange_us"], "self_check", o["self_check"]["passed"], "| rank leg:", rm.get("value"), rm.get("leg_wall_s"), rm.get("error"), rm.get("self_check",{}).get("passed"))
for k,v in o["exchange_modes"].items():
    if isinstance(v,dict): print("   ", k[:60], v.get("value"), v.get("error"))
for k,v in rm.get("exchange_modes",{}).items():
    if isinstance(v,dict): print("   rank:", k[:60], v.get("value"), v.get("error"))
PY
done
