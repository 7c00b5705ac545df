set -e
O=gpurun_out/r5c; mkdir -p $O
python -m pytest tests -q -m gpu -k "composition or vae_step_at_config or pyloop or bf16" -s > $O/pytest_sel.log 2>&1 || true
tail -5 $O/pytest_sel.log
python bench.py --config 4 --no-cpu-baseline --steps 6 > $O/c4_shipped.json 2> $O/c4_shipped.log
GGPM_LIB_PATH=$PWD/ggpm_amd/libggpm_hip.accbf16.so python bench.py --config 4 --no-cpu-baseline --steps 6 > $O/c4_accbf16.json 2> $O/c4_accbf16.log
GGPM_LIB_PATH=$PWD/ggpm_amd/libggpm_hip.accbf16.so python -m pytest tests -q -m gpu -k "bf16" -s > $O/pytest_bf16_acc.log 2>&1 || true
tail -3 $O/pytest_bf16_acc.log
python -m pytest tests -q -m gpu > $O/pytest.log 2>&1 || true
tail -5 $O/pytest.log
