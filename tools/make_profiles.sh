#!/bin/bash
# Regenerates the artefacts under profiles/ on a GPU box (run from the repo root through gpurun; writes gpurun_out/<tag>/).
#   bash tools/make_profiles.sh r05p
# rocprofv3 runs the program itself (python3 bench.py ...), never through env / bash -c; PMC passes are separate runs.
set -u
TAG=${1:-r05p}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
QUICK="--no-vae --no-cpu-baseline --no-second-cell --no-full-depth --no-roofline --no-configs4 --steps 20"

python3 $B > $OUT/bench_default.json 2> $OUT/bench_default.log
python3 $B --config 4 --no-cpu-baseline > $OUT/bench_config4.json 2> $OUT/bench_config4.log

for C in GRU LSTM; do
  rm -rf /tmp/prof_enc_$C
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_enc_$C -- python3 $B $QUICK --rnn $C > $OUT/prof_enc_$C.log 2>&1
  python3 $ROOT/tools/prof_summary.py /tmp/prof_enc_$C --steps 16 --label "encoder step $C (bench.py $QUICK --rnn $C)" > $OUT/${C}_kernel_stats.txt
  python3 $ROOT/tools/vae_timeline.py /tmp/prof_enc_$C 3 > $OUT/${C}_queue_timeline.txt
  python3 $ROOT/tools/step_listing.py /tmp/prof_enc_$C 30 > $OUT/${C}_step_listing.txt
  rm -rf /tmp/prof_vae_$C
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vae_$C -- python3 $B --only-vae --vae-profile resident --rnn $C > $OUT/prof_vae_$C.log 2>&1
  python3 $ROOT/tools/prof_summary.py /tmp/prof_vae_$C --steps 20 --label "full VAE step $C, schedules resident (bench.py --only-vae --vae-profile resident)" > $OUT/vae_${C}_kernel_stats.txt
done
python3 $ROOT/tools/vae_launches.py GRU=/tmp/prof_vae_GRU LSTM=/tmp/prof_vae_LSTM > $OUT/vae_launches.json
python3 $ROOT/tools/vae_timeline.py /tmp/prof_vae_GRU 3 > $OUT/vae_GRU_queue_timeline.txt
python3 $ROOT/tools/step_listing.py /tmp/prof_vae_GRU 25 > $OUT/vae_GRU_step_listing.txt
rm -rf /tmp/prof_vae_loop
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vae_loop -- python3 $B --only-vae --vae-profile in_loop > $OUT/prof_vae_loop.log 2>&1
python3 $ROOT/tools/prof_summary.py /tmp/prof_vae_loop --steps 20 --label "full VAE step GRU as vae_train.py:78 calls it: host batch in, schedule + uploads inside the step, no memoised index structures" > $OUT/vae_GRU_in_loop_kernel_stats.txt

# the unprofiled step: host / GPU time of the package's phase marks (no profiler attached)
cd $ROOT
PIPE=1 STEPS=30 python3 tools/vae_phase_times.py > $OUT/vae_GRU_phase_times.txt 2>&1
RNN=LSTM PIPE=1 STEPS=30 python3 tools/vae_phase_times.py > $OUT/vae_LSTM_phase_times.txt 2>&1
cd /tmp

# HBM traffic of the depth kernels: two separate PMC passes
for C in GRU LSTM; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${C}_$CTR
    rocprofv3 --pmc $CTR --output-format csv -d /tmp/pmc_${C}_$CTR -- python3 $B --no-vae --no-cpu-baseline --no-second-cell --no-full-depth --no-roofline --no-configs4 --steps 3 --warmup 1 --pool 3 --rnn $C > $OUT/pmc_${C}_$CTR.log 2>&1
  done
  python3 $ROOT/tools/pmc_summary.py /tmp/pmc_${C}_FETCH_SIZE /tmp/pmc_${C}_WRITE_SIZE 20 > $OUT/${C}_pmc_hbm_traffic.txt 2>&1
done

# SQ counters of the depth kernels (one PMC pass, no tracing)
for C in GRU LSTM; do
  rm -rf /tmp/pmc_sq_$C
  rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d /tmp/pmc_sq_$C -- python3 $B --no-vae --no-cpu-baseline --no-second-cell --no-full-depth --no-roofline --no-configs4 --steps 3 --warmup 1 --pool 3 --rnn $C > $OUT/pmc_sq_$C.log 2>&1
  python3 $ROOT/tools/pmc_sq.py /tmp/pmc_sq_$C $(echo $C | tr A-Z a-z)_ > $OUT/${C}_sq_counters.txt 2>&1
done

cd $ROOT
python3 tools/parity_report.py --config 1 --rnn GRU > $OUT/parity_report_configs1_gru.txt 2>&1
ls -la $OUT | head -40
