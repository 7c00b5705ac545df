"""Experiment: one batch of 32 as two concurrent 16-molecule halves on two streams vs as one batch (dev tool)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ggpm_amd import synth
from ggpm_amd.nnutils import make_cuda
from ggpm_amd.property_vae import HierEncoderVAE, rsample
dev = torch.device("cuda:0")
specs = [synth.random_batch(1000 + i, 32, motifs=(8, 12), n_motif_vocab=500, n_attach_vocab=1500) for i in range(6)]
full = [make_cuda(synth.tensorize(s)) for s in specs]
halves = [(make_cuda(synth.tensorize(s[:16])), make_cuda(synth.tensorize(s[16:]))) for s in specs]
torch.manual_seed(0)
model = HierEncoderVAE(bench.make_args("GRU", 300, 20, 32, 500, 1500)).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
twin = torch.cuda.Stream()

def loss_of(outs):
    hroot, hnode, hinter, hatom = outs
    _, kl = rsample(hroot, model.R_mean, model.R_var, perturb=False)
    return 0.1 * kl + 1e-3 * (hroot.sum() + hnode.sum() + hinter.sum() + hatom.sum())

def step_full(i):
    for p in model.parameters(): p.grad = None
    tree, graph = full[i % len(full)]
    loss_of(model.encoder.forward_padded(tree, graph)).backward()
    opt.step()

def step_halves(i):
    for p in model.parameters(): p.grad = None
    (t0, g0), (t1, g1) = halves[i % len(halves)]
    main = torch.cuda.current_stream()
    twin.wait_stream(main)
    with torch.cuda.stream(twin):
        l1 = loss_of(model.encoder.forward_padded(t1, g1))
    l0 = loss_of(model.encoder.forward_padded(t0, g0))
    with torch.cuda.stream(twin):
        l1.backward()
    l0.backward()
    main.wait_stream(twin)
    opt.step()

modes = {"full": step_full, "halves": step_halves}
sel = os.environ.get("ONLY")
for name, fn in ([(sel, modes[sel])] if sel else [("full", step_full), ("halves", step_halves), ("full", step_full), ("halves", step_halves)]):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    N = 30
    for i in range(N): fn(i)
    host = time.perf_counter() - t
    torch.cuda.synchronize()
    print("%-7s %.3f ms/step (host enqueue %.3f)" % (name, 1e3 * (time.perf_counter() - t) / N, 1e3 * host / N), flush=True)
