O=$PWD/gpurun_out/r5s; mkdir -p $O
python -m pytest tests/test_gpu_training_loop.py tests/test_gpu_parity.py -q -m gpu -k "loop or vae or schedule or tree_level or heads or index_structures or transpose" > $O/pytest_sub.log 2>&1; tail -5 $O/pytest_sub.log
for i in 1 2 3; do python bench.py --only-vae --rnn GRU > $O/v$i.json 2> $O/v$i.log; grep -E "full VAE step|rebuilt|as vae_train" $O/v$i.log | cut -c1-300; done
