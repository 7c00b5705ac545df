"""What FlatGradSync.pack does at the end of a VAE step (dev probe): parameters without a gradient, gradients outside the
flat buffer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench


class A:
    steps, pool, host_input = 10, 4, False


wl = bench.VaeWorkload(bench.CONFIGS[1], os.environ.get("RNN", "GRU"), A, torch.device("cuda:0"))
for i in range(3):
    wl.step(i)
_, dev_tensors, sch, _ = wl.items[0]
wl._zero()
loss, metrics = wl.model(None, None, dev_tensors, wl.orders, None, None, beta=0.1, perturb_z=True, schedule=sch)
loss.backward()
torch.cuda.synchronize()
sync = wl.sync
names = {id(p): k for k, p in wl.model.named_parameters()}
none, outside, inside = [], [], []
for p, v in zip(sync.params, sync.views):
    if p.grad is None:
        none.append((names[id(p)], p.numel()))
    elif p.grad.data_ptr() != v.data_ptr():
        outside.append((names[id(p)], p.numel()))
    else:
        inside.append((names[id(p)], p.numel()))
print("in the flat buffer already: %d params, %d elements" % (len(inside), sum(n for _, n in inside)))
print("copied by pack: %d params, %d elements" % (len(outside), sum(n for _, n in outside)))
print("zero-filled (no gradient): %d params: %s" % (len(none), none))
