O=$PWD/gpurun_out/r5v; mkdir -p $O
python -m pytest tests/test_gpu_training_loop.py -q -m gpu > $O/pytest_loop.log 2>&1; tail -2 $O/pytest_loop.log
for i in 1 2 3; do python bench.py --only-vae --rnn GRU > $O/v$i.json 2> $O/v$i.log; grep -E "full VAE step|as vae_train" $O/v$i.log | cut -c1-300; done
