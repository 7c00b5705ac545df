set -e
O=$PWD/gpurun_out/r5m; mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1 || true
tail -5 $O/pytest.log
python bench.py --only-vae --rnn GRU > $O/vae_gru.json 2> $O/vae_gru.log
grep -E "ms/step" $O/vae_gru.log | tail -5
PIPE=1 STEPS=30 python3 tools/vae_phase_times.py > $O/phase.txt 2>&1
sed -n 2,27p $O/phase.txt
