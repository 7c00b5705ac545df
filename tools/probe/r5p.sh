set -e
O=$PWD/gpurun_out/r5p; mkdir -p $O
Q="--no-cpu-baseline --no-configs4 --no-full-depth --no-second-cell --steps 30"
for V in shipped noload shipped2 noload2; do
  L=$PWD/ggpm_amd/libggpm_hip.${V%2}.so; [ "${V%2}" = shipped ] && L=$PWD/ggpm_amd/libggpm_hip.so
  GGPM_LIB_PATH=$L timeout -k 10 200 python bench.py $Q > $O/bench_$V.json 2> $O/bench_$V.log || { echo "$V failed"; tail -3 $O/bench_$V.log; exit 0; }
done
python - <<'PY'
import json
for n in ("shipped","noload","shipped2","noload2"):
    j=json.load(open("gpurun_out/r5p/bench_%s.json"%n))
    r=j["roofline"]["all_depth_kernels"]
    print(n, "enc", j["ms_per_step"], "vae", j["vae_step"]["ms_per_step"], {lv:{k:v["avg_launch_us"] for k,v in r[lv].items()} for lv in r})
PY
