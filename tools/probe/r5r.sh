O=$PWD/gpurun_out/r5r; mkdir -p $O
python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py > $O/bench_default.json 2> $O/bench_default.log; tail -c 600 $O/bench_default.log
