set -e
O=gpurun_out/r5d; mkdir -p $O
timeout -k 10 300 python tools/probe/dataflow_ab.py 6 30 GRU > $O/dataflow_ab.txt 2>&1 || echo "dataflow_ab exit $?" >> $O/dataflow_ab.txt
tail -8 $O/dataflow_ab.txt
python -m pytest tests -q -m gpu -k "composition or bf16_level or bf16_gate" -s > $O/pytest_sel.log 2>&1 || true
tail -4 $O/pytest_sel.log
