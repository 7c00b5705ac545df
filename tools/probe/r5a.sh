set -e
O=gpurun_out/r5a; mkdir -p $O
tools/probe/sigmoid_ulp > $O/sigmoid.txt 2>&1
Q="--no-vae --no-cpu-baseline --no-configs4 --no-full-depth --steps 30"
python bench.py $Q > $O/bench_acc1.json 2> $O/bench_acc1.log
GGPM_LIB_PATH=$PWD/ggpm_amd/libggpm_hip.fastsig.so python bench.py $Q > $O/bench_fast1.json 2> $O/bench_fast1.log
python bench.py $Q > $O/bench_acc2.json 2> $O/bench_acc2.log
GGPM_LIB_PATH=$PWD/ggpm_amd/libggpm_hip.fastsig.so python bench.py $Q > $O/bench_fast2.json 2> $O/bench_fast2.log
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "configs4_polymer_shard_matches_oracle" -s > $O/cal.log 2>&1
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
tail -3 $O/pytest.log
