O=$PWD/gpurun_out/r5q; mkdir -p $O
for i in 1 2; do GGPM_BENCH_TRACE_STEPS=1 python bench.py --only-vae --rnn GRU > $O/v$i.json 2> $O/v$i.log; grep -E "as vae_train|in-loop per step" $O/v$i.log | cut -c1-900; done
