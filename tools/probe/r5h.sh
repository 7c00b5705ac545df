set -e
ROOT=$PWD
O=$ROOT/gpurun_out/r5h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_vae_GRU
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vae_GRU -- python3 $ROOT/bench.py --only-vae --vae-profile resident --rnn GRU > $O/prof_vae_GRU.log 2>&1
python3 $ROOT/tools/prof_summary.py /tmp/prof_vae_GRU --steps 20 --label "full VAE step GRU, schedules resident" > $O/vae_GRU_kernel_stats.txt
python3 $ROOT/tools/step_listing.py /tmp/prof_vae_GRU 25 > $O/vae_GRU_step_listing.txt
cd $ROOT
PIPE=1 STEPS=30 python3 tools/vae_phase_times.py > $O/vae_GRU_phase_times.txt 2>&1
tail -30 $O/vae_GRU_phase_times.txt
