set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
for C in GRU LSTM; do
  rm -rf /tmp/prof_vae_$C
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vae_$C -- python3 $B --only-vae --vae-profile resident --rnn $C > $OUT/prof_vae_$C.log 2>&1
  python3 $ROOT/tools/prof_summary.py /tmp/prof_vae_$C --steps 20 --label "full VAE step $C, schedules resident (bench.py --only-vae --vae-profile resident)" > $OUT/vae_${C}_kernel_stats.txt
done
python3 $ROOT/tools/vae_launches.py GRU=/tmp/prof_vae_GRU LSTM=/tmp/prof_vae_LSTM > $OUT/vae_launches.json
python3 $ROOT/tools/vae_timeline.py /tmp/prof_vae_GRU 3 > $OUT/vae_GRU_queue_timeline.txt
python3 $ROOT/tools/step_listing.py /tmp/prof_vae_GRU 25 > $OUT/vae_GRU_step_listing.txt
cd $ROOT
PIPE=1 STEPS=30 python3 tools/vae_phase_times.py > $OUT/vae_GRU_phase_times.txt 2>&1
RNN=LSTM PIPE=1 STEPS=30 python3 tools/vae_phase_times.py > $OUT/vae_LSTM_phase_times.txt 2>&1
