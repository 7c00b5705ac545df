import sys, time
sys.path.insert(0, '.')
import torch
from ggpm_amd import synth
from ggpm_amd.decoder import DecodeSchedule
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
for seed in (1, 2, 3, 4):
    specs = synth.random_batch(seed, 32, motifs=(7, 11), n_motif_vocab=500, n_attach_vocab=1500)
    tensors = synth.tensorize(specs)
    sch = DecodeSchedule.from_specs(specs, tensors)
    ap = sch.atom_plan(tensors[1][0].shape[0], tensors[1][1].shape[0])
    ap.compact_tables(5, 3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sch.to_device(dev); ap.to_device(dev); ap.compact_device(5, 3, dev)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("upload of one batch's schedule + plan + compact tables: host %.2f ms (+%.2f ms to drain)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
