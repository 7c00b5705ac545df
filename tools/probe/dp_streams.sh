#!/bin/bash
export MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29500 + RANDOM % 400)) WORLD_SIZE=2 LOCAL_RANK=0 HSA_ENABLE_IPC_MODE_LEGACY=0
for kv in "$@"; do export "$kv"; done
(RANK=0 timeout -k 5 60 python -u tools/probe/dp_streams.py > gpurun_out/s0.log 2>&1 &)
RANK=1 timeout -k 5 60 python -u tools/probe/dp_streams.py > gpurun_out/s1.log 2>&1
sleep 2
grep "^rank" gpurun_out/s0.log; grep "^rank" gpurun_out/s1.log
