"""Host time of the forced single-rank RCCL all-reduce inside the bench step (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GGPM_FORCE_ALLREDUCE"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
import torch, torch.distributed as dist
import bench
from ggpm_amd.nnutils import make_cuda
from ggpm_amd.parallel import FlatGradSync
from ggpm_amd.property_vae import HierEncoderVAE, rsample
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
if os.environ.get("SIDE_FIRST"):
    from ggpm_amd import functional as F_
    F_._side_stream(dev); torch.cuda.current_stream()
    torch.zeros(1, device=dev)
if os.environ.get("MODE") != "nopg":
    kw = {} if os.environ.get("NO_DEVICE_ID") else {"device_id": dev}
    dist.init_process_group(os.environ.get("BACKEND", "nccl"), rank=0, world_size=1, **kw)
pool = bench.make_batches(8, 32, seed0=1000, motifs=(8, 12), n_motif=500, n_attach=1500)
batches = [make_cuda(b) for b in pool]
model = HierEncoderVAE(bench.make_args("GRU", 300, 20, 32, 500, 1500)).to(dev)
sync = FlatGradSync(model.parameters(), encoder=model.encoder)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
mode = os.environ.get("MODE", "sync")
acc = {"bwd": 0.0, "ar": 0.0, "opt": 0.0, "fwd": 0.0}
def step(i, rec):
    tree, graph = batches[i % len(batches)]
    t0 = time.perf_counter()
    sync.zero_grad()
    hroot, hnode, hinter, hatom = model.encoder.forward_padded(tree, graph)
    _, kl = rsample(hroot, model.R_mean, model.R_var, perturb=False)
    loss = 0.1 * kl + 1e-3 * (hroot.sum() + hnode.sum() + hinter.sum() + hatom.sum())
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    if mode == "sync":
        sync.all_reduce()
    elif mode == "async":
        sync.finish(sync.all_reduce(async_op=True))
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    if rec:
        acc["fwd"] += t1 - t0; acc["bwd"] += t2 - t1; acc["ar"] += t3 - t2; acc["opt"] += t4 - t3
for i in range(5): step(i, False)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for i in range(N): step(i, True)
torch.cuda.synchronize()
print("mode %s: %.3f ms/step; host per step: fwd %.3f bwd %.3f allreduce %.3f opt %.3f" % (
    mode, 1e3 * (time.perf_counter() - t0) / N, *(1e3 * acc[k] / N for k in ("fwd", "bwd", "ar", "opt"))))
if dist.is_initialized():
    dist.destroy_process_group()
