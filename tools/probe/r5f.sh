set -e
O=$PWD/gpurun_out/r5f; mkdir -p $O
python bench.py --only-vae --rnn GRU > $O/vae_gru.json 2> $O/vae_gru.log
tail -4 $O/vae_gru.log
python bench.py --only-vae --rnn LSTM > $O/vae_lstm.json 2> $O/vae_lstm.log
tail -3 $O/vae_lstm.log
python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1 || true
tail -5 $O/pytest.log
