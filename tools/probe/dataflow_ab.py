#!/usr/bin/env python3
"""The dataflow form of the atom level's forward depth loop (ggpm_level_dataflow) against the default form on the
configs[1] bench workload: bit comparison of every output and gradient, the timeout word, and ms per step / per forward in
alternating blocks (dev tool, GPU box).  Args: [blocks=6] [steps per block=30] [rnn=GRU]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from ggpm_amd import _lib

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rnn = sys.argv[3] if len(sys.argv) > 3 else "GRU"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = _lib.load(build_if_missing=False)
a = bench.parse_args(["--pool", "8", "--steps", str(steps)])
cfg = dict(bench.CONFIGS[1])
wl = bench.Workload(cfg, rnn, a, 0, 1, dev)
m = wl.model


def one(i, df):
    lib.ggpm_level_dataflow(1 if df else 0)
    tree, graph = wl.dev_batches[i % len(wl.dev_batches)]
    for p in m.parameters():
        p.grad = None
    from ggpm_amd.property_vae import rsample
    outs = m.encoder.forward_padded(tree, graph)
    _, kl = rsample(outs[0], m.R_mean, m.R_var, perturb=False)
    (0.1 * kl + 1e-3 * sum(o.sum() for o in outs)).backward()
    torch.cuda.synchronize()
    return [o.detach().clone() for o in outs] + [kl.detach().clone()], {k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None}


print("timeout word before:", lib.ggpm_level_dataflow(0))
worst = 0
for i in range(4):
    o0, g0 = one(i, False)
    o1, g1 = one(i, True)
    tmo = lib.ggpm_level_dataflow(0)
    same_o = all(torch.equal(x, y) for x, y in zip(o0, o1))
    diff = [k for k in g0 if not torch.equal(g0[k], g1[k])]
    print("batch %d: outputs bit-identical %s, gradients bit-identical %s%s, timeout word %d" % (
        i, same_o, not diff, "" if not diff else " (differ: %s)" % diff[:4], tmo))
    if not same_o:
        for n, (x, y) in enumerate(zip(o0, o1)):
            print("   out %d max |diff| %.3e of %.3e" % (n, float((x - y).abs().max()), float(x.abs().max())))
    worst |= (not same_o) or bool(diff) or tmo != 0

# timing: alternating blocks of full steps
res = {False: [], True: []}
fwd = {False: [], True: []}
for w in range(10):
    wl.step(w)
torch.cuda.synchronize()
for b in range(blocks):
    for df in (False, True):
        lib.ggpm_level_dataflow(1 if df else 0)
        for w in range(4):
            wl.step(w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            wl.step(i)
        torch.cuda.synchronize()
        res[df].append(1e3 * (time.perf_counter() - t0) / steps)
        # forward only (events around the encoder forward)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0.0
        for i in range(8):
            tree, graph = wl.dev_batches[i % len(wl.dev_batches)]
            torch.cuda.synchronize()
            e0.record()
            outs = m.encoder.forward_padded(tree, graph)
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
            del outs
        fwd[df].append(tot / 8)
tmo = lib.ggpm_level_dataflow(0)
for df in (False, True):
    print("%-9s ms/step %s  median %.3f | encoder forward ms %s median %.3f" % (
        "dataflow" if df else "default", " ".join("%.3f" % x for x in res[df]), float(np.median(res[df])),
        " ".join("%.3f" % x for x in fwd[df]), float(np.median(fwd[df]))))
print("timeout word after timing:", tmo)

# ---- per-workgroup stamps of ONE dataflow forward (100 MHz wall clock): where the time of a depth step goes
import ctypes
tree, graph = wl.dev_batches[0]
E1 = int(graph[1].shape[0])
tiles, depths = (E1 + 15) // 16, cfg["depth"]
lib.ggpm_level_dataflow(2)
outs = m.encoder.forward_padded(tree, graph)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (depths * tiles * 4))()
rc = lib.ggpm_dataflow_stamps(buf, depths, tiles)
lib.ggpm_level_dataflow(0)
st = np.frombuffer(buf, dtype=np.uint64).reshape(depths, tiles, 4).astype(np.int64)
print("stamps rc %d; atom level: %d messages, %d row tiles, %d depth steps" % (rc, E1 - 1, tiles, depths))
us = lambda x: x * 0.01
pct = lambda v: "min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % tuple(us(np.percentile(v, q)) for q in (0, 10, 50, 90, 100))
t0 = st[1:, :, 0].min()
print("%5s %9s %9s | %s" % ("depth", "first WG", "last pub", "per workgroup, us: wait | gather | GEMMs + epilogue + publish | life"))
for d in range(depths):
    if st[d, :, 3].min() == 0:
        continue
    a0, a1, a2, a3 = (st[d, :, k] for k in range(4))
    print("%5d %9.1f %9.1f | wait %s | gather %s | rest %s | life %s" % (
        d + 1, us(a0.min() - t0), us(a3.max() - t0), pct(a1 - a0), pct(a2 - a1), pct(a3 - a2), pct(a3 - a0)))
per = [us(st[d, :, 3].max() - st[d - 1, :, 3].max()) for d in range(2, depths)]
print("last publish of depth t minus last publish of depth t-1: median %.2f us (min %.2f, max %.2f)" % (
    float(np.median(per)), min(per), max(per)))
# how long before its last predecessor tile was published did a workgroup START (positive: it started early and waited)
early = []
dep = None
for d in range(2, depths):
    start = st[d, :, 0]
    prev_pub = st[d - 1, :, 3]
    # a tile depends on tiles of its own molecule: use +-6 tiles as the neighbourhood (an upper bound of the real set)
    for t in range(tiles):
        lo, hi = max(0, t - 6), min(tiles, t + 7)
        early.append(us(prev_pub[lo:hi].max() - start[t]))
early = np.array(early)
print("workgroup start relative to the last publish among tiles t-6..t+6 of the previous depth (positive = started early, us): "
      "p10 %.1f median %.1f p90 %.1f; %.0f %% started early" % (np.percentile(early, 10), np.median(early), np.percentile(early, 90),
                                                                100.0 * (early > 0).mean()))
sys.exit(1 if worst or tmo else 0)
