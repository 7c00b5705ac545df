set -e
O=$PWD/gpurun_out/r5j; mkdir -p $O
python -m pytest tests -q -m gpu -x -k "vae or heads or training_loop or data_parallel or bench_with_two" > $O/pytest_vae.log 2>&1 || true
tail -6 $O/pytest_vae.log
python bench.py --only-vae --rnn GRU > $O/vae_gru.json 2> $O/vae_gru.log
grep -E "ms/step" $O/vae_gru.log | tail -4
