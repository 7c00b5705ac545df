#!/bin/bash
# rehearsal of bench.py's N > 1 code path on a ONE-GPU box through bench.py's own launcher: two ranks, both on cuda:0,
# gloo collectives (dev probe; the command VERDICT r3 names)
ARGS="--gpus 2 --backend gloo --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-full-depth --no-second-cell"
GGPM_BENCH_ONE_DEVICE=1 GGPM_LAUNCH_TIMEOUT=280 python -u bench.py $ARGS > gpurun_out/b2_r0.json 2> gpurun_out/b2_r0.log
echo "launcher exit $?"
tail -3 gpurun_out/b2_r0.log
python - <<PY
import json
d = json.loads(open("gpurun_out/b2_r0.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("n_gpus", "value", "ms_per_step", "scaling")})
print(d["config"]["workload"])
v = d.get("vae_step", {})
print("vae_step:", {k: v.get(k) for k in ("ms_per_step", "value", "n_gpus", "error")})
PY
