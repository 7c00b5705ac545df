"""GPU timeline of the VAE step's backward (dev tool, GPU box): events recorded when the atom level's and the encoder's
backward nodes start and end (on the streams they run on), relative to the start of loss.backward().  RNN=GRU|LSTM."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from ggpm_amd import atom_decode, fused


class A:
    steps, pool, host_input = 30, 8, False


marks = {}


def wrap(cls, name):
    orig = cls.backward

    def timed(ctx, *g):
        a = torch.cuda.Event(enable_timing=True); a.record()
        out = orig(ctx, *g)
        b = torch.cuda.Event(enable_timing=True); b.record()
        marks.setdefault(name, []).append((a, b))
        return out

    cls.backward = staticmethod(timed)


wrap(atom_decode._AtomDecodeCompact, "atom level bwd")
wrap(fused._HierEncoder, "encoder bwd")
wl = bench.VaeWorkload(bench.CONFIGS[1], os.environ.get("RNN", "GRU"), A, torch.device("cuda:0"))
for i in range(16):
    wl.step(i)
torch.cuda.synchronize()
bench._settle_gc()
rows = []
for i in range(16):
    _, dev_tensors, sch, _ = wl.items[i % len(wl.items)]
    wl.opt.zero_grad(set_to_none=True)
    f0 = torch.cuda.Event(enable_timing=True); f0.record()
    loss, metrics = wl.model(None, None, dev_tensors, wl.orders, None, None, beta=0.1, perturb_z=True, schedule=sch)
    marks.clear()
    s = torch.cuda.Event(enable_timing=True); s.record()
    loss.backward()
    e = torch.cuda.Event(enable_timing=True); e.record()
    wl.opt.step()
    o = torch.cuda.Event(enable_timing=True); o.record()
    torch.cuda.synchronize()
    r = {"forward": f0.elapsed_time(s), "backward": s.elapsed_time(e), "optimizer": e.elapsed_time(o)}
    for k, v in marks.items():
        a, b = v[0]
        r[k + " start"] = s.elapsed_time(a)
        r[k + " end"] = s.elapsed_time(b)
    rows.append(r)
for k in rows[0]:
    vals = sorted(r[k] for r in rows)
    print("%-22s median %.2f ms" % (k, vals[len(vals) // 2]))
