"""cProfile of the host-side enqueue of the bench step (dev tool): where does the Python time go?"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ggpm_amd.nnutils import make_cuda
from ggpm_amd.parallel import FlatGradSync
from ggpm_amd.property_vae import HierEncoderVAE, rsample

rnn = os.environ.get("RNN", "GRU")
dev = torch.device("cuda:0")
pool = bench.make_batches(8, 32, seed0=1000, motifs=(8, 12), n_motif=500, n_attach=1500)
dev_batches = [make_cuda(b) for b in pool]
torch.manual_seed(0)
model = HierEncoderVAE(bench.make_args(rnn, 300, 20, 32, 500, 1500)).to(dev)
sync = FlatGradSync(model.parameters())
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)


host_iter = None
if os.environ.get("HOST_INPUT"):
    import itertools
    from ggpm_amd.dataloader import DevicePrefetcher
    host_iter = iter(DevicePrefetcher(itertools.cycle(pool), depth=2))


def step(i):
    tree, graph = next(host_iter) if host_iter is not None else dev_batches[i % len(dev_batches)]
    sync.zero_grad()
    hroot, hnode, hinter, hatom = model.encoder.forward_padded(tree, graph)
    _, kl = rsample(hroot, model.R_mean, model.R_var, perturb=False)
    loss = 0.1 * kl + 1e-3 * (hroot.sum() + hnode.sum() + hinter.sum() + hatom.sum())
    loss.backward()
    sync.all_reduce()
    opt.step()


for i in range(5):
    step(i)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for i in range(N):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.3f ms/step, total %.3f ms/step" % (1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N))
torch.autograd.set_multithreading_enabled(False)      # run the backward functions in this thread so cProfile sees them
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    step(i)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(60)
print(s.getvalue())
