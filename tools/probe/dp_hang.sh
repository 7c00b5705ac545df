#!/bin/bash
# runs the two data-parallel rank workers under the given environment assignments with a traceback dump after 45 s
# (dev probe for a hang):  bash tools/probe/dp_hang.sh TAG VAR=1 VAR2=0 ...
TAG=$1; shift
export MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29500 + RANDOM % 400)) WORLD_SIZE=2 LOCAL_RANK=0 HSA_ENABLE_IPC_MODE_LEGACY=0
for kv in "$@"; do export "$kv"; done
W='import faulthandler, runpy; faulthandler.dump_traceback_later(45, exit=True); runpy.run_path("tests/dp_rank_worker.py", run_name="__main__")'
(RANK=0 timeout -k 5 100 python -u -c "$W" > gpurun_out/r0_$TAG.log 2>&1 &)
RANK=1 timeout -k 5 100 python -u -c "$W" > gpurun_out/r1_$TAG.log 2>&1
sleep 2
echo "== $TAG ($*): $(grep -c 'ranks bit-identical' gpurun_out/r0_$TAG.log) VAE sections done, timeout: $(grep -c 'Timeout' gpurun_out/r0_$TAG.log)"
