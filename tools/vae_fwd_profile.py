"""cProfile of the FORWARD of the full VAE step (host side; dev tool, GPU box). RNN=GRU|LSTM."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


class A:
    steps, pool, host_input = 30, 8, False


wl = bench.VaeWorkload(bench.CONFIGS[1], os.environ.get("RNN", "GRU"), A, torch.device("cuda:0"))
for i in range(16):
    wl.step(i)
torch.cuda.synchronize()
bench._settle_gc()
pr = cProfile.Profile()
t_issue = 0.0
for i in range(8):
    _, dev_tensors, sch = wl.items[i % len(wl.items)]
    wl.opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pr.enable()
    loss, metrics = wl.model(None, None, dev_tensors, wl.orders, None, None, beta=0.1, perturb_z=True, schedule=sch)
    pr.disable()
    t_issue += time.perf_counter() - t0
    loss.backward()
    wl.opt.step()
print("forward incl. the metrics' .item(): %.2f ms" % (1e3 * t_issue / 8))
pstats.Stats(pr).sort_stats("tottime").print_stats(int(os.environ.get("TOP", "25")))
