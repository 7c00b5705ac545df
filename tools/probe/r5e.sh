set -e
O=$PWD/gpurun_out/r5e; mkdir -p $O
ROOT=$PWD
timeout -k 10 300 python tools/probe/dataflow_ab.py 3 30 GRU > $O/dataflow_ab.txt 2>&1 || echo "dataflow_ab exit $?" >> $O/dataflow_ab.txt
tail -30 $O/dataflow_ab.txt
cd /tmp && export TMPDIR=/tmp
Q="--no-vae --no-cpu-baseline --no-second-cell --no-full-depth --no-roofline --no-configs4 --steps 20"
export GGPM_LIB_PATH=$ROOT/ggpm_amd/libggpm_hip.dev.so
export GGPM_DATAFLOW=1
rm -rf /tmp/prof_df
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_df -- python3 $ROOT/bench.py $Q > $O/prof_df.log 2>&1
python3 $ROOT/tools/step_listing.py /tmp/prof_df 30 > $O/df_step_listing.txt
python3 $ROOT/tools/prof_summary.py /tmp/prof_df --steps 16 --label "encoder step GRU, dataflow form of the atom level's forward" > $O/df_kernel_stats.txt
cd $ROOT
unset GGPM_DATAFLOW GGPM_LIB_PATH
python -m pytest tests -q -m gpu -k "composition" -s > $O/pytest_sel.log 2>&1 || true
tail -4 $O/pytest_sel.log
