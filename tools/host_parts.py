"""Host enqueue time of each part of the bench step (dev tool; no profiler, no synchronisation inside the step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
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
sync = FlatGradSync(model.parameters(), encoder=model.encoder)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
acc = {}


def lap(name, t0):
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1


def step(i, record):
    tree, graph = dev_batches[i % len(dev_batches)]
    t = time.perf_counter()
    sync.zero_grad(); t = lap("zero_grad", t) if record else time.perf_counter()
    outs = model.encoder.forward_padded(tree, graph); t = lap("encoder forward", t) if record else time.perf_counter()
    _, kl = rsample(outs[0], model.R_mean, model.R_var, perturb=False); t = lap("rsample", t) if record else time.perf_counter()
    loss = 0.1 * kl + 1e-3 * (outs[0].sum() + outs[1].sum() + outs[2].sum() + outs[3].sum())
    t = lap("loss", t) if record else time.perf_counter()
    loss.backward(); t = lap("backward", t) if record else time.perf_counter()
    sync.all_reduce(); t = lap("all_reduce", t) if record else time.perf_counter()
    opt.step(); t = lap("adam", t) if record else time.perf_counter()


for i in range(40):
    step(i, False)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for i in range(N):
    step(i, True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.3f ms/step, total %.3f ms/step" % (1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N))
for k, v in acc.items():
    print("  %-16s %.3f ms" % (k, 1e3 * v / N))
