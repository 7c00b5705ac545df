O=$PWD/gpurun_out/r5u; mkdir -p $O
python bench.py --no-cpu-baseline > $O/bench2.json 2> $O/bench2.log; grep -E "full VAE|timed region done" $O/bench2.log | cut -c1-200
python bench.py --config 4 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.log; grep -E "timed region done" $O/bench_c4.log | cut -c1-200
