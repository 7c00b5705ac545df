O=$PWD/gpurun_out/r5r; mkdir -p $O
python -m pytest tests/test_gpu_training_loop.py -q -m gpu > $O/pytest_loop.log 2>&1; tail -30 $O/pytest_loop.log | cut -c1-400
