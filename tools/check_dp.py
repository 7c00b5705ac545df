"""Data-parallel consistency check (dev tool): N ranks on ONE GPU (gloo) train on different batches; parameters must
stay identical across ranks, with and without the bucketed all-reduce, and both must give the same parameters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench
from ggpm_amd.nnutils import make_cuda
from ggpm_amd.parallel import FlatGradSync, broadcast_parameters
from ggpm_amd.property_vae import HierEncoderVAE, rsample

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group(os.environ.get("BACKEND", "gloo"), rank=rank, world_size=world)
rnn = os.environ.get("RNN", "GRU")
pool = bench.make_batches(4, 8, seed0=1000 + rank * 313, motifs=(4, 7), n_motif=50, n_attach=150)
batches = [make_cuda(b) for b in pool]
results = {}
for bucketed in ("1", "0"):
    os.environ["GGPM_BUCKETED_ALLREDUCE"] = bucketed
    torch.manual_seed(0)
    model = HierEncoderVAE(bench.make_args(rnn, 100, 5, 16, 50, 150)).cuda()
    broadcast_parameters(model)
    sync = FlatGradSync(model.parameters(), encoder=model.encoder)
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    for i in range(6):
        tree, graph = batches[i % len(batches)]
        sync.zero_grad()
        hroot, hnode, hinter, hatom = model.encoder.forward_padded(tree, graph)
        _, kl = rsample(hroot, model.R_mean, model.R_var, perturb=False)
        loss = 0.1 * kl + 1e-3 * (hroot.sum() + hnode.sum() + hinter.sum() + hatom.sum())
        loss.backward()
        sync.all_reduce()
        opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    results[bucketed] = flat.cpu()
    if rank == 0:
        print("bucketed=%s early_numel=%d ranks identical: %s loss %.6f" % (bucketed, sync.early_numel, same, float(loss.detach())), flush=True)
    assert same
if rank == 0:
    d = float((results["1"] - results["0"]).abs().max())
    print("max |bucketed - single collective| over parameters: %.3e" % d, flush=True)
    assert d <= 1e-6
dist.barrier()
dist.destroy_process_group()
