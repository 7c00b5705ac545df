#!/bin/bash
# rehearsal of bench.py's N > 1 code path on a ONE-GPU box: two ranks, both on cuda:0, gloo collectives (dev probe)
export MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29500 + RANDOM % 400)) WORLD_SIZE=2 LOCAL_RANK=0 HSA_ENABLE_IPC_MODE_LEGACY=0
ARGS="--gpus 2 --backend gloo --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-full-depth --no-second-cell"
(RANK=1 timeout -k 5 280 python -u bench.py $ARGS > gpurun_out/b2_r1.log 2>&1 &)
RANK=0 timeout -k 5 280 python -u bench.py $ARGS > gpurun_out/b2_r0.json 2> gpurun_out/b2_r0.log
echo "rank 0 exit $?"
tail -3 gpurun_out/b2_r0.log
python - <<PY
import json
d = json.loads(open("gpurun_out/b2_r0.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("n_gpus", "value", "ms_per_step", "scaling")})
v = d.get("vae_step", {})
print("vae_step:", {k: v.get(k) for k in ("ms_per_step", "value", "n_gpus", "error")})
PY
