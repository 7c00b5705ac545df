O=$PWD/gpurun_out/r5t; mkdir -p $O
for i in 1 2 3; do
python tools/vae_phase_times.py ATOM_TAIL_SPLIT=True > $O/split$i.txt 2>&1; grep -E "^unmarked|^marked|atom loop done|tail issued|flush starts|optimizer issued" $O/split$i.txt | head -12
python tools/vae_phase_times.py ATOM_TAIL_SPLIT=False > $O/one$i.txt 2>&1; grep -E "^unmarked|^marked|atom loop done|tail issued|flush starts|optimizer issued" $O/one$i.txt | head -12
done
