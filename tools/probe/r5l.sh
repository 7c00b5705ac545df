set -e
ROOT=$PWD
O=$ROOT/gpurun_out/r5l; mkdir -p $O
Q="--no-vae --no-cpu-baseline --no-configs4 --no-full-depth --no-second-cell --steps 30"
for V in shipped ldsbc noload shipped2 ldsbc2; do
  L=$ROOT/ggpm_amd/libggpm_hip.${V%2}.so; [ "${V%2}" = shipped ] && L=$ROOT/ggpm_amd/libggpm_hip.so
  GGPM_LIB_PATH=$L python bench.py $Q > $O/bench_$V.json 2> $O/bench_$V.log
done
cd /tmp && export TMPDIR=/tmp
P="--no-vae --no-cpu-baseline --no-second-cell --no-full-depth --no-roofline --no-configs4 --steps 3 --warmup 1 --pool 3"
for V in shipped noload; do
  L=$ROOT/ggpm_amd/libggpm_hip.$V.so; [ $V = shipped ] && L=$ROOT/ggpm_amd/libggpm_hip.so
  export GGPM_LIB_PATH=$L
  for CTR in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${V}_$CTR
    rocprofv3 --pmc $CTR --output-format csv -d /tmp/pmc_${V}_$CTR -- python3 $ROOT/bench.py $P > $O/pmc_${V}_$CTR.log 2>&1
  done
  python3 $ROOT/tools/pmc_summary.py /tmp/pmc_${V}_FETCH_SIZE /tmp/pmc_${V}_WRITE_SIZE 12 > $O/pmc_$V.txt 2>&1
done
for V in shipped ldsbc; do
  L=$ROOT/ggpm_amd/libggpm_hip.$V.so; [ $V = shipped ] && L=$ROOT/ggpm_amd/libggpm_hip.so
  export GGPM_LIB_PATH=$L
  rm -rf /tmp/pmc_sq_$V
  rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d /tmp/pmc_sq_$V -- python3 $ROOT/bench.py $P > $O/pmc_sq_$V.log 2>&1
  python3 $ROOT/tools/pmc_sq.py /tmp/pmc_sq_$V gru_ > $O/sq_$V.txt 2>&1
done
ls $O
