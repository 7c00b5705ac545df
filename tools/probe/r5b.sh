set -e
O=gpurun_out/r5b; mkdir -p $O
python tools/probe/vae_cfg_err.py > $O/err_acc.txt 2>&1
GGPM_LIB_PATH=$PWD/ggpm_amd/libggpm_hip.fastsig.so python tools/probe/vae_cfg_err.py > $O/err_fast.txt 2>&1
python tools/probe/vae_cfg_err.py GRU 300 32 20 32 4243 > $O/err_acc_s4243.txt 2>&1
GGPM_LIB_PATH=$PWD/ggpm_amd/libggpm_hip.fastsig.so python tools/probe/vae_cfg_err.py GRU 300 32 20 32 4243 > $O/err_fast_s4243.txt 2>&1
python -m pytest tests -q -m gpu > $O/pytest.log 2>&1 || true
tail -5 $O/pytest.log
