#!/bin/bash
# Round-5 additions to tools/make_profiles.sh: rocprofv3 summaries that make the SECONDARY legs of the bench
# line reproducible from profiles/ -- configs[4] (fp32 and bf16) kernel stats + HBM counter traffic, HBM counter traffic of the
# full VAE step.  Run from the repo root through gpurun; writes gpurun_out/<tag>/.
#   bash tools/make_profiles_r05.sh r05p [part]      part: all (default) | c4 | c4pmc (counter passes only) | vae
# rocprofv3 runs the program itself (python3 bench.py ...), never through env / bash -c; PMC passes are separate runs.
set -u
TAG=${1:-r05p}
PART=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"

if [ "$PART" = all ] || [ "$PART" = c4 ] || [ "$PART" = c4pmc ]; then
  for DT in f32 bf16; do
    C4="--config 4 --dtype $DT --no-cpu-baseline --no-second-cell --no-full-depth --no-roofline --steps 6"
    if [ "$PART" != c4pmc ]; then
      rm -rf /tmp/prof_c4_$DT
      rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4_$DT -- python3 $B $C4 > $OUT/prof_c4_$DT.log 2>&1
      python3 $ROOT/tools/prof_summary.py /tmp/prof_c4_$DT --steps 6 --label "configs[4] encoder step, $DT (bench.py $C4)" > $OUT/config4_${DT}_kernel_stats.txt
      echo "c4 $DT trace done" >&2
    fi
    for CTR in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pmc_c4_${DT}_$CTR
      rocprofv3 --pmc $CTR --output-format csv -d /tmp/pmc_c4_${DT}_$CTR -- python3 $B --config 4 --dtype $DT --no-cpu-baseline --no-second-cell --no-full-depth --no-roofline --steps 2 --warmup 1 --pool 2 > $OUT/pmc_c4_${DT}_$CTR.log 2>&1
      echo "c4 $DT $CTR done" >&2
    done
    python3 $ROOT/tools/pmc_summary.py /tmp/pmc_c4_${DT}_FETCH_SIZE /tmp/pmc_c4_${DT}_WRITE_SIZE 16 > $OUT/config4_${DT}_pmc_hbm_traffic.txt 2>&1
    python3 $ROOT/tools/pmc_step_total.py /tmp/pmc_c4_${DT}_FETCH_SIZE /tmp/pmc_c4_${DT}_WRITE_SIZE adam_flat_k 12 >> $OUT/config4_${DT}_pmc_hbm_traffic.txt 2>&1
  done
fi

if [ "$PART" = all ] || [ "$PART" = vae ]; then
  for C in GRU LSTM; do
    for CTR in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pmc_vae_${C}_$CTR
      rocprofv3 --pmc $CTR --output-format csv -d /tmp/pmc_vae_${C}_$CTR -- python3 $B --only-vae --vae-profile resident --rnn $C --steps 4 --pool 4 > $OUT/pmc_vae_${C}_$CTR.log 2>&1
      echo "vae $C $CTR done" >&2
    done
    python3 $ROOT/tools/pmc_step_total.py /tmp/pmc_vae_${C}_FETCH_SIZE /tmp/pmc_vae_${C}_WRITE_SIZE adam_flat_k 16 > $OUT/vae_${C}_pmc_hbm_traffic.txt 2>&1
  done
fi
ls -la $OUT
